// dfepe_w8pt_bwd -- C-ABI entry point of the analytic adjoint of the fit (argument validation, then w8pt16.hip).
// Replaces torch.autograd's replay of the per-sample torch.svd calls of Fit.weighted_svd (deepFEPE/models/DeepFNet.py:232-256);
// closed forms and kernels: w8pt16_bwd_body.h.
#include "dfepe_common.h"
#include "w8pt16_bwd_body.h"  // W8BwdArgs

int dfepe_w8pt16_bwd_launch(const W8BwdArgs& A, bool raw, hipStream_t st);  // w8pt16.hip

extern "C" int dfepe_w8pt_bwd(const float* pts1, const float* pts2, const float* weights, int B, int N, int n_weight_sets,
                              unsigned flags,
                              float image_w, float image_h, float clamp_at, const float* save, const float* F_out,
                              const float* g_F, const float* g_residual, const float* g_epi,
                              const float* g_weights_extra, const float* g_scale, float* g_weights, float* g_pts1, float* g_pts2,
                              const void* pending_loss_head, void* stream) {
  const bool raw = (flags & DFEPE_W8PT_RAW_MATCHES) != 0;
  const int logits_mode = (flags & DFEPE_W8PT_LOGITS) ? 1 : 0;
  if (B < 0 || N <= 0 || n_weight_sets < 1) return DFEPE_ERR_INVALID_ARG;
  if (flags & (DFEPE_W8PT_SQRT2 | DFEPE_W8PT_FORCE_110 | DFEPE_W8PT_NO_HARTLEY)) return DFEPE_ERR_UNSUPPORTED;
  const unsigned variant = flags & DFEPE_W8PT_NO_ROWNORM;  // the one variant with an adjoint (row kernels, weight gradients)
  if (variant && g_pts1 != nullptr) return DFEPE_ERR_UNSUPPORTED;
  if (B == 0) return DFEPE_OK;
  if (!pts1 || (!raw && !pts2) || !weights || !save || !g_weights) return DFEPE_ERR_INVALID_ARG;
  if (g_epi && !F_out) return DFEPE_ERR_INVALID_ARG;
  if (reinterpret_cast<uintptr_t>(pending_loss_head) & 15u) return DFEPE_ERR_INVALID_ARG;
  const bool pgrad = g_pts1 != nullptr;
  if (pgrad && n_weight_sets != 1) return DFEPE_ERR_UNSUPPORTED;  // point gradients of shared correspondences would need a sum over the sets
  const int Bm = B;
  B *= n_weight_sets;
  if (!raw && ((g_pts1 == nullptr) != (g_pts2 == nullptr))) return DFEPE_ERR_INVALID_ARG;
  if (raw && pgrad && (reinterpret_cast<uintptr_t>(g_pts1) & 15u)) return DFEPE_ERR_INVALID_ARG;
  if (raw && !(image_w > 0.f && image_h > 0.f)) return DFEPE_ERR_INVALID_ARG;
  if (raw && (reinterpret_cast<uintptr_t>(pts1) & 15u)) return DFEPE_ERR_INVALID_ARG;
  if (flags & ~DFEPE_W8PT_ALL_FLAGS) return DFEPE_ERR_INVALID_ARG;
  {
    W8BwdArgs A;
    A.pts1 = pts1; A.pts2 = pts2; A.wts = weights;
    A.Bm = Bm; A.B = B; A.N = N;
    A.hw_sx = raw ? 2.0f / image_w : 0.f; A.hw_sy = raw ? 2.0f / image_h : 0.f; A.clamp_at = clamp_at;
    A.save = save; A.F_out = F_out; A.g_F = g_F; A.g_res = g_residual; A.g_epi = g_epi; A.g_w_extra = g_weights_extra; A.g_scale = g_scale;
    A.g_w = g_weights; A.g_p1 = g_pts1; A.g_p2 = g_pts2; A.logits_mode = logits_mode; A.variant = variant; A.pending_head = pending_loss_head;
    A.row_per_pair = (flags & DFEPE_W8PT_ROW_PER_PAIR) != 0;
    return dfepe_w8pt16_bwd_launch(A, raw, static_cast<hipStream_t>(stream));
  }
}
