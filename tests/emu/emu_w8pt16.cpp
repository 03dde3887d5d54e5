// Host build of the row-per-pair kernel bodies -- TEST INFRASTRUCTURE ONLY (see tests/emu/rowgroup.h).
//   g++ -O1 -std=c++17 -shared -fPIC -Itests/emu -Ipytorch-deepfepe_amd/csrc -Iinclude tests/emu/emu_w8pt16.cpp -o tests/emu/_build/libemu_w8pt16.so
// Exposes emu_w8pt16_fwd / emu_w8pt16_bwd with the argument lists of dfepe_w8pt_fwd / dfepe_w8pt_bwd (host pointers, no
// stream).  The product never loads this library.
#include <functional>
#include <memory>
#include <vector>

#include "rowgroup.h"  // the emulation (this directory comes first on the include path)
#include "w8pt16_body.h"
#include "w8pt16_bwd_body.h"

thread_local emu::Row* emu::g_row = nullptr;

namespace {
thread_local const std::function<void()>* g_body = nullptr;

void fiber_main() {
  emu::Row* r = emu::g_row;
  const int lane = r->cur;
  (*g_body)();
  r->done[lane] = true;
  swapcontext(&r->fib[lane], &r->sched);
}

// run `body` on the 16 lanes of one row, in lockstep
void run_row(const std::function<void()>& body) {
  constexpr size_t kStack = 1 << 20;
  static thread_local std::unique_ptr<char[]> stacks(new char[emu::kLanes * kStack]);
  emu::Row row;
  memset(row.done, 0, sizeof(row.done));
  memset(row.op, 0, sizeof(row.op));
  memset(row.buf, 0, sizeof(row.buf));
  emu::g_row = &row;
  g_body = &body;
  for (int l = 0; l < emu::kLanes; ++l) {
    getcontext(&row.fib[l]);
    row.fib[l].uc_stack.ss_sp = stacks.get() + (size_t)l * kStack;
    row.fib[l].uc_stack.ss_size = kStack;
    row.fib[l].uc_link = &row.sched;
    makecontext(&row.fib[l], fiber_main, 0);
  }
  for (bool any = true; any;) {
    any = false;
    for (int l = 0; l < emu::kLanes; ++l) {
      if (row.done[l]) continue;
      any = true;
      row.cur = l;
      swapcontext(&row.sched, &row.fib[l]);
    }
  }
  emu::g_row = nullptr;
}

template <template <int, bool> class Body, class Args>
int dispatch(const Args& A, int B, int N, bool raw) {
  for (int pair = 0; pair < B; ++pair) {
    double xch[48];
    auto go = [&](auto itc, auto rawc) {
      run_row([&]() { Body<decltype(itc)::value, decltype(rawc)::value>::run(A, pair, xch); });
    };
    auto with_it = [&](auto rawc) {
      if (N <= 16) go(std::integral_constant<int, 1>{}, rawc);
      else if (N <= 32) go(std::integral_constant<int, 2>{}, rawc);
      else if (N <= 64) go(std::integral_constant<int, 4>{}, rawc);
      else if (N <= 112) go(std::integral_constant<int, 7>{}, rawc);
      else go(std::integral_constant<int, 8>{}, rawc);
    };
    if (raw) with_it(std::true_type{}); else with_it(std::false_type{});
  }
  return 0;
}

template <int IT, bool RAW>
struct FwdBody {
  static void run(const W8Args& A, int pair, double* xch) { w8pt16_fwd_pair<IT, RAW>(A, pair, xch); }
};
template <int IT, bool RAW>
struct BwdBody {
  static void run(const W8BwdArgs& A, int pair, double* xch) { w8pt16_bwd_pair<IT, RAW>(A, pair, xch); }
};
}  // namespace

extern "C" int emu_w8pt16_fwd(const float* pts1, const float* pts2, const float* weights, int B, int N, int n_weight_sets,
                              unsigned flags, float image_w, float image_h, float clamp_at, float* F_out, float* residual,
                              float* epi_res, float* save, float* weights_out) {
  if (N < 1 || N > 128) return -3;
  const bool raw = (flags & DFEPE_W8PT_RAW_MATCHES) != 0;
  W8Args A;
  A.pts1 = pts1; A.pts2 = pts2; A.wts = weights;
  A.Bm = B; A.B = B * n_weight_sets; A.N = N;
  A.hw_sx = raw ? 2.0f / image_w : 0.f; A.hw_sy = raw ? 2.0f / image_h : 0.f; A.clamp_at = clamp_at;
  A.F_out = F_out; A.residual = residual; A.epi_res = epi_res; A.save = save; A.weights_out = weights_out;
  A.logits_mode = (flags & DFEPE_W8PT_LOGITS) ? 1 : 0;
  A.variant = flags & (DFEPE_W8PT_SQRT2 | DFEPE_W8PT_NO_ROWNORM | DFEPE_W8PT_FORCE_110 | DFEPE_W8PT_NO_HARTLEY);
  return dispatch<FwdBody>(A, A.B, N, raw);
}

extern "C" int emu_w8pt16_bwd(const float* pts1, const float* pts2, const float* weights, int B, int N, int n_weight_sets,
                              unsigned flags, float image_w, float image_h, float clamp_at, const float* save,
                              const float* F_out, const float* g_F, const float* g_residual, const float* g_epi,
                              const float* g_weights_extra, float* g_weights, float* g_pts1, float* g_pts2) {
  if (N < 1 || N > 128) return -3;
  const bool raw = (flags & DFEPE_W8PT_RAW_MATCHES) != 0;
  W8BwdArgs A;
  A.pts1 = pts1; A.pts2 = pts2; A.wts = weights;
  A.Bm = B; A.B = B * n_weight_sets; A.N = N;
  A.hw_sx = raw ? 2.0f / image_w : 0.f; A.hw_sy = raw ? 2.0f / image_h : 0.f; A.clamp_at = clamp_at;
  A.save = save; A.F_out = F_out; A.g_F = g_F; A.g_res = g_residual; A.g_epi = g_epi; A.g_w_extra = g_weights_extra;
  A.g_w = g_weights; A.g_p1 = g_pts1; A.g_p2 = g_pts2;
  A.logits_mode = (flags & DFEPE_W8PT_LOGITS) ? 1 : 0;
  return dispatch<BwdBody>(A, A.B, N, raw);
}
