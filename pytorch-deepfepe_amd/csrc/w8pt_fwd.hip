// dfepe_w8pt_fwd -- C-ABI entry point of the weighted normalised 8-point fit (argument validation, then w8pt16.hip).
//
// Arithmetic contract: Fit.normalize / Fit.weighted_svd / NormalizeAndExpand_HW / compute_epi_residual
// (deepFEPE/models/DeepFNet.py:93-120,148-257, deepFEPE/dsac_tools/utils_F.py:400-413); bodies in w8pt16_body.h.
// One kernel family serves every shape: one 16-lane row per pair (N <= 128, or large batches of any N) or the 16 rows of a
// workgroup per pair (N > 128 at small batch).  The round-1 family (one wavefront per pair, fp32 Jacobi in LDS, its own `save`
// format) was retired in round 3: the cooperative rows are faster at every size it still served (N = 1000, 512 pairs: 20 vs 32 us).
#include "dfepe_common.h"
#include "w8pt16_body.h"

int dfepe_w8pt16_fwd_launch(const W8Args& A, bool raw, hipStream_t st);  // w8pt16.hip

extern "C" int dfepe_w8pt_fwd(const float* pts1, const float* pts2, const float* weights, int B, int N,
                              int n_weight_sets, unsigned flags, float image_w, float image_h, float clamp_at, float* F_out,
                              float* residual, float* epi_res, float* save, float* weights_out, void* stream) {
  const bool raw = (flags & DFEPE_W8PT_RAW_MATCHES) != 0;
  const int logits_mode = (flags & DFEPE_W8PT_LOGITS) ? 1 : 0;
  const unsigned variant = flags & (DFEPE_W8PT_SQRT2 | DFEPE_W8PT_NO_ROWNORM | DFEPE_W8PT_FORCE_110 | DFEPE_W8PT_NO_HARTLEY);
  if (B < 0 || N <= 0 || n_weight_sets < 1) return DFEPE_ERR_INVALID_ARG;
  // the textbook variants are forward-only; un-normalised rows alone (Fit(normalize_SVD=False)) have a backward
  if (variant && save && variant != DFEPE_W8PT_NO_ROWNORM) return DFEPE_ERR_UNSUPPORTED;
  if (B == 0) return DFEPE_OK;
  if (!pts1 || (!raw && !pts2) || !weights || !F_out || !residual) return DFEPE_ERR_INVALID_ARG;
  if (raw && !(image_w > 0.f && image_h > 0.f)) return DFEPE_ERR_INVALID_ARG;
  if (raw && (reinterpret_cast<uintptr_t>(pts1) & 15u)) return DFEPE_ERR_INVALID_ARG;  // float4 loads
  if (reinterpret_cast<uintptr_t>(save) & 15u) return DFEPE_ERR_INVALID_ARG;          // wide stores of the record

  if (flags & ~DFEPE_W8PT_ALL_FLAGS) return DFEPE_ERR_INVALID_ARG;  // unknown flag bits are rejected, not ignored
  {
    W8Args A;
    A.pts1 = pts1; A.pts2 = pts2; A.wts = weights;
    A.Bm = B; A.B = B * n_weight_sets; A.N = N;
    A.hw_sx = raw ? 2.0f / image_w : 0.f; A.hw_sy = raw ? 2.0f / image_h : 0.f; A.clamp_at = clamp_at;
    A.F_out = F_out; A.residual = residual; A.epi_res = epi_res; A.save = save; A.weights_out = weights_out;
    A.logits_mode = logits_mode; A.variant = variant; A.row_per_pair = (flags & DFEPE_W8PT_ROW_PER_PAIR) != 0;
    return dfepe_w8pt16_fwd_launch(A, raw, static_cast<hipStream_t>(stream));
  }
}
