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
#include "loss_tail_body.h"

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
      if (N > 128) go(std::integral_constant<int, 0>{}, rawc);
      else if (N <= 16) go(std::integral_constant<int, 1>{}, rawc);
      else if (N <= 32) go(std::integral_constant<int, 2>{}, rawc);
      else if (N <= 64) go(std::integral_constant<int, 4>{}, rawc);
      else if (N <= 112) go(std::integral_constant<int, 7>{}, rawc);
      else go(std::integral_constant<int, 8>{}, rawc);
    };
    if (raw) with_it(std::true_type{}); else with_it(std::false_type{});
  }
  return 0;
}

bool g_lean = false;  // emu_set_lean: run the <= 256-register build of the forward fit (w8pt16_body.h: LEAN) where the library would
template <int IT, bool RAW>
struct FwdBody {
  static void run(const W8Args& A, int pair, double* xch) {
    if constexpr (IT == 7 || IT == 8) {
      if (g_lean) {
        if (A.variant == 0) w8pt16_fwd_pair<IT, RAW, true, 1, true>(A, pair, xch); else w8pt16_fwd_pair<IT, RAW, false, 1, true>(A, pair, xch);
        return;
      }
    }
    if (A.variant == 0) w8pt16_fwd_pair<IT, RAW, true>(A, pair, xch); else w8pt16_fwd_pair<IT, RAW, false>(A, pair, xch);
  }
};
template <int IT, bool RAW>
struct BwdBody {
  static void run(const W8BwdArgs& A, int pair, double* xch) { w8pt16_bwd_pair<IT, RAW>(A, pair, xch); }
};
}  // namespace

extern "C" void emu_set_lean(int on) { g_lean = on != 0; }

extern "C" int emu_w8pt16_fwd(const float* pts1, const float* pts2, const float* weights, int B, int N, int n_weight_sets,
                              unsigned flags, float image_w, float image_h, float clamp_at, float* F_out, float* residual,
                              float* epi_res, float* save, float* weights_out) {
  if (N < 1) return -3;
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
                              const float* g_weights_extra, const float* g_scale, float* g_weights, float* g_pts1, float* g_pts2) {
  if (N < 1) return -3;
  const bool raw = (flags & DFEPE_W8PT_RAW_MATCHES) != 0;
  W8BwdArgs A;
  A.pts1 = pts1; A.pts2 = pts2; A.wts = weights;
  A.Bm = B; A.B = B * n_weight_sets; A.N = N;
  A.hw_sx = raw ? 2.0f / image_w : 0.f; A.hw_sy = raw ? 2.0f / image_h : 0.f; A.clamp_at = clamp_at;
  A.save = save; A.F_out = F_out; A.g_F = g_F; A.g_res = g_residual; A.g_epi = g_epi; A.g_w_extra = g_weights_extra; A.g_scale = g_scale;
  A.g_w = g_weights; A.g_p1 = g_pts1; A.g_p2 = g_pts2;
  A.logits_mode = (flags & DFEPE_W8PT_LOGITS) ? 1 : 0;
  return dispatch<BwdBody>(A, A.B, N, raw);
}

// the fused loss tail, pair by pair (the batch sums of the loss head are left to the caller: part[B][kTailParts])
extern "C" int emu_loss_tail(const float* F_layers, int L, int B, const float* T1, const float* T2, int t_stride, const float* K,
                             const float* virt1, const float* virt2, int M, float clamp_at, const float* q_gt, const float* t_gt,
                             const float* R_gt, float clamp_q, float clamp_t, float coef_F, float coef_q, float coef_t,
                             float* loss_sum, float* E_layers, float* q_l2, float* t_l2, float* R_deg, float* t_deg, int* sel,
                             float* g_F, double* part) {
  if (M < 1 || M > 128 || L < 1 || L > kTailMaxLayers) return -3;
  TailArgs A;
  A.F_layers = F_layers; A.L = L; A.B = B; A.M = M; A.t_stride = t_stride; A.T1 = T1; A.T2 = T2; A.K = K;
  A.virt1 = virt1; A.virt2 = virt2; A.clamp_at = clamp_at; A.q_gt = q_gt; A.t_gt = t_gt; A.R_gt = R_gt;
  A.clamp_q = clamp_q; A.clamp_t = clamp_t; A.coef_F = coef_F; A.coef_q = coef_q; A.coef_t = coef_t;
  A.loss_sum = loss_sum; A.E_layers = E_layers; A.q_l2 = q_l2; A.t_l2 = t_l2; A.R_deg = R_deg; A.t_deg = t_deg; A.sel = sel; A.g_F = g_F;
  for (int pair = 0; pair < B; ++pair) {
    float lds[kTailLdsFloats];
    double* pp = part + (size_t)pair * kTailParts;
    for (int e = 0; e < kTailParts; ++e) pp[e] = 0.0;
    auto go = [&](auto itc) { run_row([&]() { loss_tail_pair<decltype(itc)::value>(A, pair, lds, pp); }); };
    if (M <= 16) go(std::integral_constant<int, 1>{});
    else if (M <= 32) go(std::integral_constant<int, 2>{});
    else if (M <= 64) go(std::integral_constant<int, 4>{});
    else if (M <= 112) go(std::integral_constant<int, 7>{});
    else go(std::integral_constant<int, 8>{});
  }
  return 0;
}
