/*
 * dfepe.h — C ABI of libdfepe_hip.so: the MI355X (gfx950) weighted-8-point hot path of deepFEPE.
 *
 * The reference (eric-yyjau/pytorch-deepFEPE) is pure Python and has no FFI layer; the drop-in
 * boundary is therefore its Python call surface.  Each entry point below replaces
 * the arithmetic of the cited reference function; the Python mirror of that surface
 * (pytorch-deepfepe_amd/compat) binds these symbols with ctypes.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.  Every pointer is a DEVICE pointer to
 *     contiguous fp32 (or int32 where stated) memory owned by the caller; nothing is allocated,
 *     freed or retained by the library; there is no global mutable state (re-entrant).
 *   - `stream` is a hipStream_t (NULL = default stream); work is enqueued asynchronously on it.
 *     The caller has made the owning device current (PyTorch does).
 *   - Return value: DFEPE_OK (0) or a negative DFEPE_ERR_* code; never throws.
 *   - Layouts follow the reference tensors: points [B,N,3] row-major, matrices [.,3,3] row-major
 *     flattened to 9, per-layer stacks [L,B,...].
 *   - Convention of F: x2^T F x1 = 0 (deepFEPE/models/DeepFNet.py:203-205).
 */
#ifndef DFEPE_H
#define DFEPE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFEPE_VERSION 154 /* 0.6.0 */

#define DFEPE_OK 0
#define DFEPE_ERR_INVALID_ARG (-1) /* null pointer, non-positive size, bad flag combination   */
#define DFEPE_ERR_HIP (-2)         /* a HIP runtime call failed (launch, attribute)            */
#define DFEPE_ERR_UNSUPPORTED (-3) /* valid request this build cannot serve (e.g. N too large) */

/* floats per pair in the `save` buffer handed from dfepe_w8pt_fwd to dfepe_w8pt_bwd */
#define DFEPE_SAVE_FLOATS 128

/* flags for dfepe_w8pt_fwd / dfepe_w8pt_bwd */
#define DFEPE_W8PT_RAW_MATCHES 1u /* `pts1` is matches_xy_ori [B,N,4] in pixels, `pts2` unused (may be NULL);
                                     the image-size normalisation of NormalizeAndExpand_HW is fused in   */
#define DFEPE_W8PT_LOGITS 2u      /* `weights` holds logits [B,N]; softmax over N (F.softmax(logits, dim=2),
                                     DeepFNet.py:443,512) is fused in and the weights are written to `weights_out`;
                                     the backward then returns the gradient w.r.t. the logits                */
/* variants of the textbook normalised 8-point solvers of dsac_tools (forward only):                         */
#define DFEPE_W8PT_SQRT2 4u       /* Hartley scale sqrt(2) (utils_F._normalize_XY, utils_F.py:23) instead of the
                                     literal 1.4142 of Fit.normalize                                         */
#define DFEPE_W8PT_NO_ROWNORM 8u  /* rows of the design matrix are not normalised to unit length
                                     (utils_F._E_from_XY / _F_from_XY, utils_F.py:122-130,239-247)           */
#define DFEPE_W8PT_FORCE_110 16u  /* singular values of the 3x3 forced to (1,1,0) instead of (s1,s2,0)
                                     (utils_F._E_from_XY, utils_F.py:148-149)                                */

#define DFEPE_W8PT_NO_HARTLEY 32u /* no Hartley normalisation, T1 = T2 = I (`normalize=False` of _E_from_XY /
                                     _F_from_XY, utils_F.py:116-119,233-236)                                 */

#define DFEPE_W8PT_ROW_PER_PAIR 64u /* scheduling only, same function, same `save` record: never spread a pair over the 16 rows of
                                       a cooperative workgroup (what the library does for 128 < N <= 2048 below 3072 pairs);
                                       one 16-lane row per pair, correspondences re-read per phase.  A/B timing; pass the same
                                       bit to dfepe_w8pt_bwd or not, the record format does not depend on it.             */
#define DFEPE_W8PT_ALL_FLAGS 127u   /* any other bit in `flags` is DFEPE_ERR_INVALID_ARG                              */
#define DFEPE_W8PT16_MAX_N 128      /* largest N whose correspondences ONE row keeps in registers for the whole kernel; above it
                                       a cooperative workgroup per pair does (N <= 2048), or they are re-read per phase       */

int dfepe_version(void);
const char *dfepe_strerror(int code);
int dfepe_save_floats(void);

/* Self-test of the 16-lane row primitives the small-N solver kernels are built on (DPP row_newbcast / row_mirror /
 * row_half_mirror / quad_perm; pytorch-deepfepe_amd/csrc/rowgroup.h).  One wavefront: x, y [64] doubles in,
 * out [14][64] doubles: bcast<0>, bcast<5>, bcast<15>, row sum, sum of lanes 2..8, exchange partners 15-l, l^7, l^2, l^1,
 * fp32 row max, int32 row sum (of (int)(16 x)), y + bcast<3>(x) y, fp32 row sum, fp32 bcast<9> + int bcast<12> + lane id.
 * No counterpart in the reference; tests/test_rowgroup_gpu.py checks the hardware against these definitions. */
int dfepe_selftest_rowgroup(const double *x, const double *y, double *out, void *stream);

/*
 * Weighted normalised 8-point fit, forward.
 * Replaces: Fit.forward / Fit.weighted_svd / Fit.normalize (deepFEPE/models/DeepFNet.py:148-179,181-257,278-295),
 *           with DFEPE_W8PT_RAW_MATCHES also NormalizeAndExpand_HW (DeepFNet.py:93-120),
 *           and, when epi_res != NULL, utils_F.compute_epi_residual(pts1, pts2, out, clamp_at)
 *           (deepFEPE/dsac_tools/utils_F.py:400-413) as called in the recurrent loop (DeepFNet.py:479).
 *
 *   pts1, pts2 [B,N,3]  homogeneous points (pts*[:,:,2] is used exactly like the reference does), or
 *   pts1       [B,N,4]  pixel matches when DFEPE_W8PT_RAW_MATCHES (image_w/image_h = W,H of image_size)
 *   weights    [S,B,N]  S = n_weight_sets >= 1 weightings of the SAME B pairs (the reference's [B,1,N] is S = 1; S > 1
 *                       serves solver-only workloads that score several weightings per pair, e.g. all IRLS layers of a
 *                       fixed-logits step in one launch); every per-pair output then has a leading S dimension
 *   F_out      [B,9]    T2^T F' T1                          (reference `out`)
 *   residual   [B,N]    X f/|f|                             (reference `residual`)
 *   epi_res    [B,N]    or NULL
 *   save       [B,DFEPE_SAVE_FLOATS] or NULL (needed for backward); 16-byte aligned
 *   weights_out[B,N]    softmax(logits) when DFEPE_W8PT_LOGITS (may be NULL), ignored otherwise
 * Sign gauge: the reference inherits LAPACK's arbitrary sign of f; here f is oriented so that its
 * largest-magnitude component is positive (F_out and residual flip together, everything downstream is
 * sign-invariant).
 */
int dfepe_w8pt_fwd(const float *pts1, const float *pts2, const float *weights, int B, int N, int n_weight_sets,
                   unsigned flags, float image_w, float image_h, float clamp_at,
                   float *F_out, float *residual, float *epi_res, float *save, float *weights_out, void *stream);

/*
 * Backward of dfepe_w8pt_fwd w.r.t. the weights (analytic eigenvector / rank-2 / epipolar-residual
 * adjoints; replaces torch.autograd through the per-sample torch.svd calls, DeepFNet.py:232-256).
 *   g_F [B,9], g_residual [B,N], g_epi [B,N]: upstream gradients, each may be NULL (= zero)
 *   F_out: the forward output (needed when g_epi != NULL)
 *   g_weights_extra [B,N] or NULL: with DFEPE_W8PT_LOGITS, an upstream gradient on the `weights_out` tensor itself
 *     (the next estimator layer reads the weights, DeepFNet.py:487); added before the softmax adjoint
 *   with DFEPE_W8PT_LOGITS `weights` must be the forward's `weights_out` and g_weights receives d/d(logits)
 *   g_scale: device pointer to ONE float that multiplies the three upstream gradients (NULL = 1): the upstream gradient of a
 *     scalar loss whose d loss / d F was formed with unit upstream (dfepe_loss_tail), applied here because everything
 *     downstream of g_F, g_residual, g_epi is linear in them
 *   g_weights [B,N]: written (not accumulated)
 *   g_pts1, g_pts2 [B,N,3] or NULL: gradient w.r.t. the point coordinates (through the rows, the Hartley transforms and,
 *     when g_epi is given, the residual's direct dependence); with DFEPE_W8PT_RAW_MATCHES g_pts1 is [B,N,4] (gradient
 *     w.r.t. the pixel matches, 16-byte aligned) and g_pts2 is ignored.  NULL skips that part of the kernel.
 *   pending_loss_head: NULL, or the `workspace` of a dfepe_loss_tail(..., defer_head = 1) call enqueued earlier on the same
 *     stream: this launch then also finishes that call's loss head (packed / scalars), in spare wavefronts of its first
 *     workgroup where the kernel family allows it, as a kernel of its own behind it otherwise.  Pass it to exactly ONE
 *     backward launch of the step.
 */
int dfepe_w8pt_bwd(const float *pts1, const float *pts2, const float *weights, int B, int N, int n_weight_sets,
                   unsigned flags, float image_w, float image_h, float clamp_at,
                   const float *save, const float *F_out,
                   const float *g_F, const float *g_residual, const float *g_epi, const float *g_weights_extra,
                   const float *g_scale, float *g_weights, float *g_pts1, float *g_pts2, const void *pending_loss_head,
                   void *stream);

/*
 * Closing steps of the 8-point solvers for an EXPLICIT design matrix (dense-W form of the textbook solvers).
 * Replaces: the part of utils_F._E_from_XY / _F_from_XY after `XX = torch.mm(W, XX)` (deepFEPE/dsac_tools/utils_F.py:129-155,
 *           245-275) when W [N,N] is not diagonal: V[:, -1] of torch.svd(XX), the 3x3 step (S3 -> 0, or (1,1,0) with
 *           DFEPE_W8PT_FORCE_110), T2^T F T1.  A diagonal W is a per-correspondence weight: dfepe_w8pt_fwd serves it from the points.
 *   rows [B,N,9] design rows as the SVD sees them (already multiplied by W); T1, T2 [B,9] or both NULL (no de-normalisation);
 *   F_out [B,9].  flags: 0 or DFEPE_W8PT_FORCE_110.  Sign gauge as dfepe_w8pt_fwd.  Forward only.
 */
int dfepe_w8pt_rows_fwd(const float *rows, int B, int N, unsigned flags, const float *T1, const float *T2, float *F_out,
                        void *stream);

/*
 * F-loss and E-from-F over all layers.
 * Replaces: the per-layer body of get_all_loss_DeepF (deepFEPE/train_good_utils.py:325-358):
 *   pts*_eval = T* virt*^T ; losses = compute_epi_residual(pts1_eval, pts2_eval, F_layer, clamp_at) ;
 *   E_layer = K^T T2^T F_layer T1 K.
 *   F_layers [L,B,9]; T1,T2 [B,9] (t_stride = 9) or a single [9] shared by the batch (t_stride = 0);
 *   K [B,9]; virt1, virt2 [B,M,3]
 *   loss_sum [L,B]  = sum over the M virtual points of the clamped residual (caller divides by M / B)
 *   E_layers [L,B,9] or NULL
 */
int dfepe_floss_fwd(const float *F_layers, int L, int B, const float *T1, const float *T2, int t_stride,
                    const float *K, const float *virt1, const float *virt2, int M, float clamp_at,
                    float *loss_sum, float *E_layers, void *stream);

/*  g_loss_sum [L,B] or NULL, g_E [L,B,9] or NULL  ->  g_F_layers [L,B,9] (written).
 *  When g_loss_sum is NULL and g_loss_coef != 0 every entry of g_loss_sum is taken to be g_loss_coef * (*g_scale)
 *  (g_scale: device pointer to one float, NULL = 1): the adjoint of a plain mean over layers, pairs and points. */
int dfepe_floss_bwd(const float *F_layers, int L, int B, const float *T1, const float *T2, int t_stride,
                    const float *K, const float *virt1, const float *virt2, int M, float clamp_at,
                    const float *g_loss_sum, float g_loss_coef, const float *g_scale, const float *g_E,
                    float *g_F_layers, void *stream);

/*
 * Pose loss: decompose E^T into the two rotations / two translations, compare with the ground truth.
 * Replaces: the per-layer, per-sample loop of get_Rt_loss (deepFEPE/train_good_utils.py:96-239):
 *   utils_F._get_M2s (utils_F.py:478-498), utils_geo._R_to_q (utils_geo.py:58-86), _l2_error (:165-167),
 *   strict-'<' candidate selection (:160-168), rot12_to_angle_error (:150-155), vector_angle (:175-179).
 *   E_layers [L,B,9]; q_gt [B,4] (qs_cam); t_gt [B,3] (ts_cam, normalised inside); R_gt [B,9] = inv(delta)[:3,:3]
 *   q_l2, t_l2 [L,B]; R_deg, t_deg [L,B] (may be NULL); sel [L,B] int32 (bit0: R candidate, bit1: t candidate; may be NULL)
 */
int dfepe_pose_fwd(const float *E_layers, int L, int B, const float *q_gt, const float *t_gt, const float *R_gt,
                   float *q_l2, float *t_l2, float *R_deg, float *t_deg, int *sel, void *stream);

/*  g_q_l2, g_t_l2 [L,B] (either may be NULL) -> g_E [L,B,9] (written).
 *  When a gradient pointer is NULL and its coefficient is non-zero, the upstream gradient of that error is
 *  coef * (*g_scale) where the error is <= its clamp and 0 above it: the adjoint of
 *  clamp(stack(err), 0, clamp).mean() * balance  (deepFEPE/Train_model_pipeline.py:580-586) with coef = balance/(L*B). */
int dfepe_pose_bwd(const float *E_layers, int L, int B, const float *q_gt, const float *t_gt,
                   const float *g_q_l2, const float *g_t_l2, float coef_q, float clamp_q, float coef_t, float clamp_t,
                   const float *g_scale, float *g_E, void *stream);

/*
 * Loss head: reduce the per-pair sums to the scalars the training step uses.
 * Replaces: losses.mean() per layer / loss_F (train_good_utils.py:340-364) and the clamped qt mix
 * (Train_model_pipeline.py:580-586), plus the packed sums a data-parallel all-reduce needs.
 *   loss_sum [L,B] (sum over the M virtual points), q_l2, t_l2 [L,B] (may be NULL: no pose loss)
 *   packed   [L+4] doubles: sum_b loss_sum[l,b] (L), sum clamp(q), sum clamp(t), B, M
 *   scalars  [4+L] floats: loss = loss_F + loss_qt, loss_F, loss_qt, 0, then the L per-layer means   (local batch)
 */
int dfepe_loss_head(const float *loss_sum, const float *q_l2, const float *t_l2, int L, int B, int M,
                    float clamp_q, float clamp_t, float balance_q, float balance_t,
                    double *packed, float *scalars, void *stream);

/*
 * The whole loss tail of the hot-path step in ONE launch (one 16-lane row per pair): dfepe_floss_fwd + E-from-F +
 * dfepe_pose_fwd + dfepe_loss_head, and -- when g_F_layers != NULL -- dfepe_pose_bwd + dfepe_floss_bwd for
 *     loss = balance_F * mean_{l,b,m}(F-loss) + balance_q * mean_{l,b}(clamp(q_l2, 0, clamp_q)) + balance_t * mean_{l,b}(clamp(t_l2, 0, clamp_t))
 * with unit upstream gradient (every coefficient of such a loss is a constant: the adjoint of a term is formed where its
 * forward value is).  Replaces: get_all_loss_DeepF's per-layer body (deepFEPE/train_good_utils.py:325-358), get_Rt_loss's
 * loop (:96-239) and the loss mixing of Train_model_pipeline.py:580-587 (which uses balance_F = 0 when if_qt_loss).
 *   arguments as in dfepe_floss_fwd / dfepe_pose_fwd / dfepe_loss_head; M <= 128 (else DFEPE_ERR_UNSUPPORTED: use those)
 *   q_gt == NULL: no pose part (then t_gt, q_l2, t_l2, R_deg, t_deg, sel are ignored)
 *   grad_pairs: number of pairs the means run over in the gradient coefficients (B, or the global batch under data parallelism)
 *   g_F_layers [L,B,9] or NULL; feed it to dfepe_w8pt_bwd with g_scale = the upstream gradient of the loss
 *   packed [L+4] doubles, scalars [4+L] floats: as dfepe_loss_head, scalars[0] = balance_F * loss_F + loss_qt
 *   workspace: dfepe_loss_tail_workspace_bytes(B) bytes, 16-byte aligned, contents irrelevant (per-workgroup partial sums;
 *     a one-workgroup kernel enqueued right behind adds them in a fixed order: no floating-point atomics, deterministic)
 *   defer_head: 0 = packed / scalars are complete when this call's work is (two launches).  1 = only the per-pair outputs
 *     and g_F_layers are; the batch sums are left pending in `workspace` and finished by the dfepe_w8pt_bwd launch that is
 *     handed `workspace` as its pending_loss_head (same stream, before packed / scalars are read).  For a training step
 *     whose backward follows at once (a captured graph): the backward does not need the scalars, and the head's launch
 *     (~7 us) leaves the step's critical path.
 */
size_t dfepe_loss_tail_workspace_bytes(int B);
int dfepe_loss_tail(const float *F_layers, int L, int B, const float *T1, const float *T2, int t_stride, const float *K,
                    const float *virt1, const float *virt2, int M, float clamp_at,
                    const float *q_gt, const float *t_gt, const float *R_gt, float clamp_q, float clamp_t,
                    float balance_F, float balance_q, float balance_t, double grad_pairs,
                    float *loss_sum, float *E_layers, float *q_l2, float *t_l2, float *R_deg, float *t_deg, int *sel,
                    float *g_F_layers, double *packed, float *scalars, void *workspace, int defer_head, void *stream);
/* The pending loss head of a dfepe_loss_tail(..., defer_head = 1) call as a launch of its own, on any stream ordered after that
 * call (instead of riding in a dfepe_w8pt_bwd launch): lets the head and the data-parallel all-reduce of `packed` that follows it
 * run on a side stream while the backward fits run on the main one.  `workspace`: the one given to dfepe_loss_tail. */
int dfepe_loss_head_pending(const void *workspace, void *stream);

/*
 * The loss tail behind the reference's OWN call sequence (get_all_loss_DeepF, then get_Rt_loss, then the caller's clamp /
 * balance mixing in torch, Train_model_pipeline.py:508-586): the coefficients of the loss are not known when the forward
 * runs, so this variant of dfepe_loss_tail leaves, next to the same per-pair outputs, the three JACOBIANS of every
 * (layer, pair):  J [L,B,27] = d loss_sum / dF (9) | d q_l2 / dF (9) | d t_l2 / dF (9)   (unclamped, unit upstream),
 * and dfepe_loss_tail_bwd turns whatever upstream gradients autograd delivers into d loss / dF in one launch:
 *   g_F[l,b] = g_loss_sum[l,b] J_F + g_q_l2[l,b] J_q + g_t_l2[l,b] J_t        (each upstream pointer may be NULL = zero;
 *   gradients that arrive on the batch statistics of dfepe_loss_stats enter through g_mean_* / g_all_*, see there)
 * Replaces: torch.autograd through get_all_loss_DeepF's per-layer body (deepFEPE/train_good_utils.py:325-358) and
 *           get_Rt_loss's per-sample loop (:96-239).  No batch sums here (the caller's torch means take them).
 *   arguments as in dfepe_loss_tail, M <= 112 (else DFEPE_ERR_UNSUPPORTED); q_gt == NULL: no pose part (J_q = J_t = 0); want_floss_jac == 0: J_F = 0 and the F-loss
 *   adjoint work is skipped (an objective without the F-loss, the reference's if_qt_loss)
 */
int dfepe_loss_tail_jac(const float *F_layers, int L, int B, const float *T1, const float *T2, int t_stride, const float *K,
                        const float *virt1, const float *virt2, int M, float clamp_at,
                        const float *q_gt, const float *t_gt, const float *R_gt, int want_floss_jac,
                        float *loss_sum, float *E_layers, float *q_l2, float *t_l2, float *R_deg, float *t_deg, int *sel,
                        float *J, void *stream);
int dfepe_loss_tail_bwd(const float *J, int L, int B, const float *g_loss_sum, const float *g_q_l2, const float *g_t_l2,
                        const float *g_mean_loss, const float *g_all_loss, const float *g_mean_q, const float *g_all_q,
                        const float *g_mean_t, const float *g_all_t, float stat_loss_scale, float *g_F_layers, void *stream);

/*
 * Batch statistics of the per-pair loss terms in one launch.
 * Replaces: the per-layer `.mean()` calls and python sums of get_all_loss_DeepF (deepFEPE/train_good_utils.py:343-364: losses.mean()
 *           per layer, loss_F = sum / len, loss_min_layers / loss_min_batch :375-376, loss_epi_res :429-438) and of get_Rt_loss
 *           (:272-283: x.mean() per layer, mean_list) -- a dozen small reductions in the reference's style, one launch here.
 *   up to four sets k of rows x_k [rows_k, C] (rows_k = 0: absent; rows_k <= 64), C = pairs
 *   out: for every present set in order, rows_k floats = scale_k * mean over C of each row, then 1 float = the mean of those
 *   row_min0 [rows0] or NULL: scale0 * min over C of each row of set 0; col_min0 [C] or NULL: scale0 * min over the rows of set 0
 * dfepe_loss_tail_bwd's g_mean_* [L] / g_all_* [1] take the gradients of such row means / of their mean for the sets loss_sum
 * (scale = stat_loss_scale), q_l2, t_l2 directly: no expansion of a mean's gradient to [L,B] in between.
 */
int dfepe_loss_stats(const float *x0, int rows0, float scale0, const float *x1, int rows1, float scale1,
                     const float *x2, int rows2, float scale2, const float *x3, int rows3, float scale3, int C,
                     float *out, float *row_min0, float *col_min0, void *stream);

/*
 * Cheirality-checked pose from E.
 * Replaces: utils_F._E_to_M_train (deepFEPE/dsac_tools/utils_F.py:679-763): the four candidates of _get_M2s in
 * the order (R1,t),(R1,-t),(R2,t),(R2,-t), linear triangulation of every correspondence (the reference calls
 * cv2.triangulatePoints), count of points with 0 < Z < depth_thres in both cameras, first arg-max wins.
 *   E [B,9]; K [B,9]; matches [B,N,4] pixels
 *   pre [B,9] or NULL: when given, the matrix decomposed is pre^T E pre -- pass E = F (the fit's output) and pre = T K to fuse
 *     E-from-F (E = K^T T^T F T K, train_good_utils.py:356-358) into this launch
 *   Rt_cam [B,12]  inverse (camera motion) of the winner, zeros when no candidate has a valid point
 *   winner [B] int32 (-1 when none); counts [B,4] int32
 * Arithmetic: the DLT null vector of every (correspondence, rotation) in packed fp32; a correspondence whose depth tests lie
 * within that vector's a-posteriori error bound of a threshold is decided by the fp64 route instead (fp64 normal matrix + one
 * Rayleigh-quotient iteration), so the counts are those of an fp64 DLT.
 */
int dfepe_cheirality(const float *E, const float *pre, const float *K, const float *matches, int B, int N, float depth_thres,
                     float *Rt_cam, int *winner, int *counts, void *stream);
/* The same with flags and an optional workspace.
 * DFEPE_CHEIR_FP64_ONLY: every correspondence takes the fp64 route (no packed-fp32 decisions): the build the adaptive default is
 *   tested against for EXACT equality of the counts (tests/test_fullsize_gpu.py), and its upper bound in time.
 * workspace: NULL, or dfepe_cheirality_workspace_bytes(B) bytes of device memory (8-byte aligned, overwritten).  With it the
 *   per-pair constants of the correspondence loop (the decomposition of E, the two candidate projection matrices K [R|t]) are
 *   formed by a preparation launch with one LANE per pair and fetched by the main kernel through scalar loads, instead of being
 *   re-derived by every wavefront (a closed-form fp64 SVD issued for 64 lanes: ~12 % of the launch at one wavefront per pair).
 *   Same outputs bit for bit. */
#define DFEPE_CHEIR_FP64_ONLY 1u
size_t dfepe_cheirality_workspace_bytes(int B);
int dfepe_cheirality_ex(const float *E, const float *pre, const float *K, const float *matches, int B, int N, float depth_thres,
                        unsigned flags, void *workspace, float *Rt_cam, int *winner, int *counts, void *stream);

/*
 * Fit + E-from-F + cheirality-checked pose in one call (BASELINE config 5: one weighted 8-point fit, then the pose of its F).
 * Replaces: Fit.forward -> E = K^T T^T F T K (train_good_utils.py:356-358) -> utils_F._E_to_M_train (utils_F.py:679-763).
 * Same outputs as dfepe_w8pt_fwd (DFEPE_W8PT_RAW_MATCHES required; DFEPE_W8PT_LOGITS optional; no `save`: forward only) followed
 * by dfepe_cheirality(F_out, pre, ...), bit for bit; for 128 < N <= 2048 below 3072 pairs it is ONE launch (the cooperative
 * workgroup of the fit goes on to decompose pre^T F pre and to triangulate its pair), otherwise the two launches.
 * workspace: NULL or dfepe_cheirality_workspace_bytes(B) bytes, handed to dfepe_cheirality_ex when the two launches run.
 */
int dfepe_w8pt_pose_fwd(const float *matches, const float *weights, int B, int N, unsigned flags, float image_w, float image_h,
                        float clamp_at, const float *K, const float *pre, float depth_thres, float *F_out, float *residual,
                        float *epi_res, float *weights_out, float *Rt_cam, int *winner, int *counts, void *workspace, void *stream);

/*
 * Reductions of the validation summary on the device.
 * Replaces: the numpy post-processing of write_metrics_summary (deepFEPE/train_good_utils.py:758-856) over the per-pair
 * results of val_rt (:553-646): epipolar-distance inlier ratios at 0.1 / 1.0 px, F1 of "est < th" against "gt < th",
 * medians and maxima of the pose errors, np.histogram counts of the errors over the thresholds
 * 0, 0.01, 0.03, 0.05, 0.1, 0.3, 0.5, 1, 2, 5, 10, 90, 180 degrees.
 *   epi_est, epi_gt [n_epi] epipolar distances of every correspondence under the estimated / ground-truth F (epi_gt may be NULL)
 *   err_q, err_t [B] rotation / translation errors in degrees
 *   out: dfepe_metrics_summary_bytes() bytes, 8-byte aligned, overwritten:
 *     uint64 counts[8]  = #est<0.1, #est<1, (TP, FP, FN) at 0.1, (TP, FP, FN) at 1.0
 *     uint64 hist[2][12] bin counts for err_q, err_t (bins [th_k, th_k+1), the last closed; values outside [0,180] dropped)
 *     float  max[2], float mids[2][2] = the two middle order statistics of err_q, err_t (median = their mean)
 * Raw counts, not ratios: the host derives the ratios after ONE device-to-host copy per validation epoch.
 */
size_t dfepe_metrics_summary_bytes(void);
int dfepe_metrics_summary(const float *epi_est, const float *epi_gt, size_t n_epi, const float *err_q, const float *err_t,
                          int B, void *out, void *stream);

/*
 * Epipolar metrics of dsac_tools.utils_F (utils_F.py:291-361) on homogeneous-or-not 2-D points.
 *   kind: 0 = _sym_epi_dist (squared; `eps` is added to the two squared line norms: 1e-10 in the reference's batched branch,
 *         0 in its 2-D branch), 1 = _sampson_dist, 2 = _epi_distance (writes 3 planes: mean, d1, d2);
 *         | DFEPE_EPI_HOMOGENEOUS: X, Y are [B,N,3] homogeneous points used as they are (if_homo=True), else [B,N,2]
 *   F [B,9]; out [B,N] (kind 0,1) or [3,B,N] (kind 2); clamp_at < 0 means no clamp (clamp_at=None; kind 0 only)
 */
#define DFEPE_EPI_HOMOGENEOUS 8
int dfepe_epi_metrics(int kind, const float *F, const float *X, const float *Y, int B, int N, float clamp_at,
                      float eps, float *out, void *stream);

/*
 * Stand-alone symmetric epipolar residual (the same arithmetic dfepe_w8pt_fwd fuses in its epilogue).
 * Replaces: utils_F.compute_epi_residual(pts1, pts2, F, clamp_at) (deepFEPE/dsac_tools/utils_F.py:400-413).
 *   pts1, pts2 [B,N,3]; F [B,9]; out [B,N].   bwd: g_out [B,N] -> g_F [B,9] (gradient w.r.t. F only).
 */
int dfepe_epi_residual_fwd(const float *pts1, const float *pts2, const float *F, int B, int N, float clamp_at,
                           float *out, void *stream);
int dfepe_epi_residual_bwd(const float *pts1, const float *pts2, const float *F, int B, int N, float clamp_at,
                           const float *g_out, float *g_F, void *stream);

/*
 * Small pose-geometry helpers, batched over n items (one lane each).
 *   kind 0  rotation -> quaternion          in0 = R [n,9]              out [n,4]   (utils_geo._R_to_q, utils_geo.py:58-86)
 *   kind 1  rotation angle in degrees       in0 = R0 [n,9], in1 = R1   out [n]     (utils_geo.rot12_to_angle_error, :150-155)
 *   kind 2  vector angle in degrees         in0 = v1 [n,3], in1 = v2   out [n]     (utils_geo.vector_angle, :175-179)
 *   kind 3  singular values forced to 1,1,0 in0 = E [n,9]              out [n,9]   (utils_F._F_to_E projection, utils_F.py:457-461;
 *                                                                                   Train_model_pipeline.py:954-964)
 *   kind 4  four-fold decomposition         in0 = E [n,9]              out [n,21] = R1[9] R2[9] t[3]
 *                                                                                  (utils_F._get_M2s, utils_F.py:478-498)
 *   kind 5  congruence A^T F A              in0 = F [n,9], in1 = A [n,9] out [n,9]  (E = K^T T2^T F T1 K with A = T K when
 *                                                                                   T1 = T2: train_good_utils.py:356-358,366-369;
 *                                                                                   utils_F._F_to_E's K^T F K, utils_F.py:456)
 *   kind 6  camera rotation of a scene motion  in0 = delta [n,16]       out [n,9] = inv(delta)[:3,:3]
 *                                                                                  (get_Rt_loss, train_good_utils.py:134,170)
 */
int dfepe_geo_misc(int kind, const float *in0, const float *in1, int n, float *out, void *stream);

/*
 * Inputs of the recurrent model from the pixel matches.
 * Replaces: DeepFNet.get_input (deepFEPE/models/DeepFNet.py:362-391) with NormalizeAndExpand_HW (:93-120): the image-size
 *           normalisation T_HW (x, y, 1) of both point sets and the estimator's input channels ((x+1)/2, (y+1)/2 of both images,
 *           then the Q quality channels) -- about fifteen elementwise / bmm launches in the reference, one here.
 *   matches [B,N,4] pixels (16-byte aligned); quality [B,N,Q] or NULL (Q = 0)
 *   weight_in or NULL: element (copy k, channel c < 4+Q, pair b, point i) goes to
 *     weight_in[k * copy_stride + c * channel_stride + b * batch_stride + i], k < n_copies.  The reference's [B,4+Q,N] tensor is
 *     channel_stride = N, batch_stride = (4+Q) N; channel-major storage [C,B,N] (channel_stride = B N, batch_stride = N) lets the
 *     fit kernels write the recurrent channels (weights, epipolar residual, residual) of the next estimator input as plain
 *     [B,N] blocks, and n_copies fills the point channels of every layer's input buffer in this one launch (no torch.cat per
 *     layer, DeepFNet.py:484-489)
 *   pts1, pts2 [B,N,3] or NULL: homogeneous normalised points (the arithmetic of dfepe_w8pt_fwd's DFEPE_W8PT_RAW_MATCHES prologue)
 */
int dfepe_deepf_input(const float *matches, const float *quality, int B, int N, int Q, float image_w, float image_h,
                      float *weight_in, size_t channel_stride, size_t batch_stride, int n_copies, size_t copy_stride,
                      float *pts1, float *pts2, void *stream);

/*
 * Per-pair dot products of two per-layer stacks: out[l,b] = sum_n a[l,b,n] * b[l,b,n].
 * Replaces: the `epi_res * weights` products under loss_epi_res (deepFEPE/train_good_utils.py:429-438), whose means
 *           dfepe_loss_stats then takes as one more row set.
 *   a, b: n_layers blocks of [B,N] floats, block l at a + l * a_layer_stride (resp. b + l * b_layer_stride): the layers may
 *   live in separate per-layer buffers a fixed distance apart (the channel-major estimator inputs of dfepe_deepf_input);
 *   out [n_layers,B]
 */
int dfepe_row_dot(const float *a, size_t a_layer_stride, const float *b, size_t b_layer_stride, int n_layers, int B, int N,
                  float *out, void *stream);

/*
 * InstanceNorm1d(affine) + LeakyReLU on rows of N contiguous floats ("next" row f-1: the part of the weight estimator
 * between its 1x1 convolutions, deepFEPE/models/ErrorEstimators.py:47-64; the convolutions themselves are GEMMs).
 *   Y, A, gA, gY: [(c*R + r)*N + n]  (channel-major: c < C channels, r < R rows per channel, N points), 16-byte aligned;
 *   gamma, beta [C]; stats [C*R*2] = (mean, 1/sqrt(var+eps)) per row, written by fwd and read by bwd;
 *   row_ggamma, row_gbeta [C*R]: per-row contributions to d/d(gamma[c]), d/d(beta[c]) (sum over r is left to the caller).
 *   N must be a multiple of 4 and <= 512 (DFEPE_ERR_UNSUPPORTED otherwise).
 */
int dfepe_inorm_lrelu_fwd(const float *Y, const float *gamma, const float *beta, int C, int R, int N, float eps, float slope,
                          float *A, float *stats, void *stream);
int dfepe_inorm_lrelu_bwd(const float *Y, const float *gA, const float *gamma, const float *beta, const float *stats,
                          int C, int R, int N, float slope, float *gY, float *row_ggamma, float *row_gbeta, void *stream);

/*
 * The weight estimator on the matrix cores, N = dfepe_est_points() = 100 points per pair.
 * Replaces: ErrorEstimator.forward, the Conv1d(k=1) -> InstanceNorm1d(affine) -> LeakyReLU stack and its autograd backward
 *           (deepFEPE/models/ErrorEstimators.py:47-64, called at deepFEPE/models/DeepFNet.py:441,510).
 * Every fp32 operand travels as 16-bit PLANES, exact remainders of each other.  Forward products (round 5): TWO fp16 planes per
 * operand (22 mantissa bits), three MFMAs (a0 b0 + a0 b1 + a1 b0, fp32 accumulate): fp32-class accuracy at half the matrix work of
 * the six bf16 products of rounds 3-4; a layer's weights are split scaled by the power of two that brings max |W| into [8, 16)
 * (their low plane would otherwise sit in fp16's subnormal range), the accumulators are scaled back before the statistics.
 * Backward products: two bf16 planes each (three MFMAs, ~2^-16; gradients keep bf16's range without any loss scaling), so a
 * forward layer that will be differentiated also leaves its activation as two bf16 planes.  Domain: |activation| <= 65504 (fp16;
 * InstanceNorm bounds it by |gamma| sqrt(N - 1) + |beta|) -- beyond it the logits are NaN, not silently wrong.
 * Plane buffers are 16-bit, POINT-major and K-blocked: element (row, ch) of a plane with `rows` rows
 * lives at ((ch / 32) * rows + row) * 32 + ch % 32; `*_plane` arguments are the element strides between planes; ncols = pairs * 100.
 *   dfepe_est_split      fp32 [rows][C_src] (ld = src_ld) -> n_planes bf16 planes with C (% 32 == 0) channels, the tail zero
 *   dfepe_est_absmax     *word = max(*word, bit pattern of max |src[i]|) -- the caller zeroes the word first; the scale of
 *                        dfepe_est_split_f16 / dfepe_est_layer_fwd / dfepe_est_gemm_nt_f16 is derived from it on the device
 *   dfepe_est_wprep      (version 152) the weights of n_layers <= 8 layers (host arrays of device pointers W[l] [Co[l]][Ci[l]]) in two
 *                        launches: words[l] = bit pattern of max |W[l]| (partial maxima in `workspace`, dfepe_est_wprep_workspace_bytes
 *                        bytes, contents irrelevant: no atomics, no zeroing), then
 *                        planes_f16[l] = two fp16 planes [Co][K = Ci rounded up to 32] of W[l] scaled as in dfepe_est_split_f16 and,
 *                        where planes_wt (and planes_wt[l]) is not null, two bf16 planes of W[l]^T [K][Co] (Co % 32 == 0) -- what
 *                        dfepe_est_gemm_nt / dfepe_est_dgrad_in_bwd multiply dY by.  At the reference's batch sizes (4-32 pairs) a
 *                        call's bookkeeping launches (per layer: maximum, split, transposed split, reductions, fills) cost more
 *                        than its GEMMs; this and dfepe_est_colsum take one estimator call from ~75 launches to ~30
 *   dfepe_est_colsum     dst[s][c] = sum_r src[s][r * cols[s] + c], r < rows[s], for n_seg <= 40 segments in ONE launch, additions in a
 *                        fixed order (rows[s] = 0: zeros, src[s] may be null): all reductions of one backward -- per-pair d gamma /
 *                        d beta partials, split-K partials of the weight gradients -- and the zero gradients of the cancelled biases
 *   dfepe_est_split_f16  the same into two fp16 planes, the values multiplied by the scale of `absmax` (null: unscaled)
 *   dfepe_est_layer_fwd  planes_out[2][...M] (fp16) = split(leaky_relu(instance_norm(W X) * gamma + beta)), planes_bwd[2][...M]
 *                        (bf16; null when no gradient is wanted) the same activation for the backward; rstd [pairs][M];
 *                        W planes [2] of [M][K] split with `absmax` (or null), X planes [2] of [ncols][K], both fp16; the
 *                        convolution bias cancels in the normalisation
 *   dfepe_est_gemm_nt    out[col][m] (fp32, ld = ldc) = sum_k A[m][k] B[col][k] on n_planes (2 or 3) bf16 planes: the data
 *                        gradient dA = W^T dY
 *   dfepe_est_gemm_nt_f16  the forward's plain product on two fp16 planes each, A split with `absmax` (or null)
 *   dfepe_est_gemm_tn    part[slices][Cout][Cin] = split-K partial sums of dW = dY^T X (two planes each; the caller adds the slices)
 *   dfepe_est_in_bwd     dY planes [2] = adjoint of InstanceNorm + LeakyReLU given dA [ncols][C] fp32 (or its rank-one head form
 *                        dlogit[col] * w_head[c]), the layer's output planes [2] (bf16), rstd, gamma, beta; per-pair d gamma / d beta
 *   dfepe_est_dgrad_in_bwd  (round 5) the data gradient dA = dY_next W_next (WT = W_next^T planes [2] of [M][K], dY_next planes [2] of
 *                        [ncols][K], bf16) and dfepe_est_in_bwd of the layer below in ONE launch: dA stays in the accumulators (two
 *                        whole pairs x 128 channels per workgroup), the layer's output planes are read twice (statistics, then dY),
 *                        dY planes [2], per-pair d gamma / d beta out.  dA never reaches memory: 8 of the 20 bytes per element the
 *                        two launches moved.  N = dfepe_est_points() only
 *   dfepe_est_norm_fwd   any N points per pair (the reference's SIFT configurations: up to 2000): planes_out [2] (fp16), planes_bwd
 *                        [2] (bf16, or null) and rstd from the plain product Y fp32 [n_pairs * N][ldy] of dfepe_est_gemm_nt_f16 --
 *                        InstanceNorm (biased variance, two-pass),
 *                        affine, LeakyReLU, split; dfepe_est_layer_fwd is this fused into the product for N = dfepe_est_points().
 *                        splits = 1: one launch, a workgroup per (pair, 64 channels); splits in 2..64 (a dozen pairs do not fill the
 *                        chip): each pair's rows over `splits` workgroups in two launches (partial mean / squared deviations, merged
 *                        pairwise; then the normalisation), part = workspace of n_pairs * splits * 2 * C floats
 *   dfepe_est_in_bwd_n   dfepe_est_in_bwd for any N (ncols = n_pairs * N)
 *   dfepe_est_dgamma_zero  the two adjoints above recover x^ as (z - beta) / gamma and take it as 0 where gamma is exactly 0 (their
 *                        d beta and dY are right there, d gamma is not): this launch, run after them, overwrites dgamma_part[pair][ch]
 *                        of every channel with |gamma| < 1e-30 from a recomputation of the layer's product (input planes [2], bf16, of
 *                        [ncols][K-blocked], fp32 weights W [C][ldw], Ci input channels); channels with gamma != 0 cost an idle
 *                        workgroup each.  N <= 4096.  The upstream gradient comes from dA, from the head's rank-one form, or -- behind
 *                        dfepe_est_dgrad_in_bwd, which never writes dA -- is recomputed for the channel from dY_next planes [2] of
 *                        [ncols][C_next] and the next layer's fp32 weights W_next [C_next][ldw_next].  dfepe_est_dgamma_zero_multi: the
 *                        same for n_layers <= 8 layers in ONE launch (host arrays indexed by layer; ldw = Ci, ldw_next = C)
 *   dfepe_est_forward / dfepe_est_backward  (version 153) ONE call per estimator pass: the whole stack of n_hidden <= 8 layers and its
 *                        one-channel head through the entry points above, in the order the host code used to issue them (bit-identical
 *                        results).  x [B][C0][N], W[l] [Co[l]][Ci[l]] (Ci[l] = Co[l-1], Ci[0] = C0; Co % 32 == 0), gamma / beta [l]
 *                        [Co[l]], w_head [Co of the last layer], b_head [1] or null: fp32, contiguous, host arrays of device pointers.
 *                        The caller brings the memory: `saved` (dfepe_est_saved_bytes; null for a forward no backward will follow; what
 *                        the backward reads: every layer's bf16 planes, reciprocal deviations, transposed weight planes; need_gx does not
 *                        change its layout -- a backward may ask for gx or not), a transient workspace per pass (dfepe_est_*_workspace_bytes;
 *                        16-byte aligned, contents irrelevant) and the outputs: logits [B * N]; g_W[l] [Co][Ci], g_bias[l] [Co] (exact
 *                        zeros: the convolution bias cancels in the normalisation), g_gamma[l], g_beta[l] [Co], g_w_head, g_b_head
 *                        (or null), gx [B][C0][N] (or null).  N = dfepe_est_points() takes the fused epilogues, any other N >= 2 the
 *                        plain products.  Why: at the reference's batch sizes the HOST was the limiter of the eager training step
 *                        (~30 launches and ~40 allocations per pass from Python)
 *   dfepe_est_head_fwd   logits[col] = sum_c w[c] a[col][c] + bias[0]   (the last Conv1d(C -> 1)); a: the forward's planes [2] (fp16)
 *   dfepe_est_head_dw    part[blocks][C] = partial sums of d w = sum_col dlogit[col] a[col][c]; a: the backward's planes [2] (bf16);
 *                        bias_part [blocks] (or null; version 154) = partial sums of dlogit: the head bias gradient's, in the same launch
 * Version 154 (round 6) -- the reference's own shapes (N = 1000-2000 points, 4-12 pairs per batch, deepFEPE/configs/kitti_corr_baseline.yaml:12-13)
 * and an estimator that one model forward calls several times (DeepFNet.update_weights, deepFEPE/models/DeepFNet.py:510):
 *   dfepe_est_gemm_nt_f16_splitk / dfepe_est_gemm_nt_splitk   the plain products with their K steps shared by `splits` workgroups per tile:
 *                        partial products out[z][col][m], z < splits <= K / 32, split_stride (>= ncols * ldc) floats apart.  With a few
 *                        pairs per batch a K-heavy layer is a dozen 128 x 208 tiles: a chain of 16-32 K steps on 5 % of the chip
 *   dfepe_est_norm_fwd_r / dfepe_est_in_bwd_r   dfepe_est_norm_fwd / dfepe_est_in_bwd_n for N <= 2048 with the pair's block RESIDENT IN
 *                        REGISTERS: one launch, one read of the product, given as `splits` partials (added in slice order); a workgroup =
 *                        one pair x 32 channels
 *   dfepe_est_gemm_nt_gx the data gradient of the FIRST layer stored as the estimator's input gradient: element (pair, ch, n) at
 *                        pair * gx_stride_pair + ch * gx_stride_ch + n (was: a transposing launch behind the plain product).  x and gx of
 *                        dfepe_est_forward / dfepe_est_backward take the same two strides: dense [B][C0][N] (C0 N, N) or a view of the
 *                        channel-major [C0][B][N] buffers compat.DeepFNet keeps its estimator inputs in (N, B N) -- no copies
 *   dfepe_est_gemm_tn_multi   dfepe_est_gemm_tn for n_layers <= 8 layers over the same columns in ONE launch (host arrays indexed by layer)
 *   dfepe_est_prep_bytes / dfepe_est_prepare   the weights' planes of one estimator (power-of-two scales, scaled fp16 planes, transposed
 *                        bf16 planes) into a caller-owned buffer `prep`, two launches; dfepe_est_forward / dfepe_est_backward given
 *                        `prep` read them instead of making their own per call.  The weights must not change between dfepe_est_prepare
 *                        and the last backward handed `prep` (one optimizer step apart at the earliest)
 *   dfepe_est_forward / dfepe_est_backward   take `prep` (or null); per layer they run the fused epilogue (N = dfepe_est_points(), enough
 *                        tiles) or plain product -> dfepe_est_norm_fwd_r / dfepe_est_in_bwd_r (any other N <= 2048, and K-heavy layers
 *                        on few tiles, split over K) -- or the strided kernels beyond N = 2048; over few columns the weight gradients of
 *                        all layers are one dfepe_est_gemm_tn_multi launch at the end of the backward
 */
int dfepe_est_points(void);
int dfepe_est_split(const float *src, long rows, int C_src, int src_ld, int C, int n_planes, void *planes, size_t plane_stride,
                    void *stream);
int dfepe_est_absmax(const float *src, long n, unsigned *word, void *stream);
size_t dfepe_est_wprep_workspace_bytes(int n_layers);
int dfepe_est_wprep(int n_layers, const float *const *W, const int *Co, const int *Ci, void *const *planes_f16,
                    void *const *planes_wt, unsigned *words, void *workspace, void *stream);
int dfepe_est_colsum(int n_seg, const float *const *src, const int *rows, const int *cols, float *const *dst, void *stream);
int dfepe_est_split_f16(const float *src, long rows, int C_src, int src_ld, int C, const unsigned *absmax, void *planes,
                        size_t plane_stride, void *stream);
int dfepe_est_layer_fwd(const void *W, size_t w_plane, const void *X, size_t x_plane, int M, int ncols, int K,
                        const unsigned *absmax, const float *gamma, const float *beta, float eps, float slope, void *planes_out,
                        size_t out_plane, void *planes_bwd, size_t bwd_plane, float *rstd, void *stream);
int dfepe_est_gemm_nt_f16(const void *A, size_t a_plane, const void *B, size_t b_plane, int M, int ncols, int K,
                          const unsigned *absmax, float *out, int ldc, void *stream);
int dfepe_est_gemm_nt(const void *A, size_t a_plane, const void *B, size_t b_plane, int M, int ncols, int K, int n_planes,
                      float *out, int ldc, void *stream);
int dfepe_est_gemm_tn(const void *dY, size_t dy_plane, int Cout, const void *X, size_t x_plane, int Cin, int ncols, int slices,
                      float *part, void *stream);
int dfepe_est_gemm_nt_f16_splitk(const void *A, size_t a_plane, const void *B, size_t b_plane, int M, int ncols, int K,
                                 const unsigned *absmax, float *out, int ldc, int splits, size_t split_stride, void *stream);
int dfepe_est_gemm_nt_splitk(const void *A, size_t a_plane, const void *B, size_t b_plane, int M, int ncols, int K, float *out, int ldc,
                             int splits, size_t split_stride, void *stream);
int dfepe_est_gemm_nt_gx(const void *A, size_t a_plane, const void *B, size_t b_plane, int M, int ncols, int K, float *gx, int C0, int N,
                         long gx_stride_pair, long gx_stride_ch, void *stream);
int dfepe_est_gemm_tn_multi(int n_layers, const void *const *dY, const size_t *dy_plane, const int *Cout, const void *const *X,
                            const size_t *x_plane, const int *Cin, int ncols, const int *slices, float *const *part, void *stream);
int dfepe_est_norm_fwd_r(const float *Y, int ldy, int splits, size_t split_stride, int C, long n_pairs, int N, const float *gamma,
                         const float *beta, float eps, float slope, void *planes_out, size_t out_plane, void *planes_bwd,
                         size_t bwd_plane, float *rstd, void *stream);
int dfepe_est_in_bwd_r(const float *dA, int ldd, int splits, size_t split_stride, const float *dlogit, const float *w_head,
                       const void *planes, size_t plane_stride, const float *rstd, const float *gamma, const float *beta, float slope,
                       int C, long n_pairs, int N, void *dY, size_t dy_plane, float *dgamma_part, float *dbeta_part, void *stream);
size_t dfepe_est_prep_bytes(int n_hidden, const int *Co, const int *Ci);
int dfepe_est_prepare(int n_hidden, const float *const *W, const int *Co, const int *Ci, void *prep, void *stream);
int dfepe_est_dgamma_zero(const float *dA, const float *dlogit, const float *w_head, const void *out_planes, size_t out_plane,
                          const void *in_planes, size_t in_plane, const float *W, int ldw, int Ci, const float *rstd, const float *gamma,
                          float slope, int C, int N, long n_pairs, float *dgamma_part, const void *dY_next, size_t dyn_plane,
                          const float *W_next, int ldw_next, int C_next, void *stream);
int dfepe_est_dgamma_zero_multi(int n_layers, const float *const *dA, const float *const *dlogit, const float *const *w_head,
                                const void *const *out_planes, const size_t *out_plane, const void *const *in_planes,
                                const size_t *in_plane, const float *const *W, const int *Ci, const float *const *rstd,
                                const float *const *gamma, const int *C, float *const *dgamma_part, const void *const *dY_next,
                                const size_t *dyn_plane, const float *const *W_next, const int *C_next, float slope, int N, long n_pairs,
                                void *stream);
int dfepe_est_dgrad_in_bwd(const void *WT, size_t wt_plane, const void *dY_next, size_t dyn_plane, int M, int ncols, int K,
                           const void *aout, size_t aout_plane, const float *rstd, const float *gamma, const float *beta, float slope,
                           void *dY, size_t dy_plane, float *dgamma_part, float *dbeta_part, void *stream);
int dfepe_est_in_bwd(const float *dA, const float *dlogit, const float *w_head, const void *planes, size_t plane_stride,
                     const float *rstd, const float *gamma, const float *beta, float slope, int C, int ncols, void *dY,
                     size_t dy_plane, float *dgamma_part, float *dbeta_part, void *stream);
int dfepe_est_norm_fwd(const float *Y, int ldy, int C, long n_pairs, int N, const float *gamma, const float *beta, float eps,
                       float slope, void *planes_out, size_t out_plane, void *planes_bwd, size_t bwd_plane, float *rstd, int splits,
                       float *part, void *stream);
int dfepe_est_in_bwd_n(const float *dA, const float *dlogit, const float *w_head, const void *planes, size_t plane_stride,
                       const float *rstd, const float *gamma, const float *beta, float slope, int C, long n_pairs, int N, void *dY,
                       size_t dy_plane, float *dgamma_part, float *dbeta_part, int splits, float *part, void *stream);
size_t dfepe_est_saved_bytes(int n_hidden, const int *Co, const int *Ci, long B, int C0, int N, int need_gx);
size_t dfepe_est_forward_workspace_bytes(int n_hidden, const int *Co, const int *Ci, long B, int C0, int N, int keep);
size_t dfepe_est_backward_workspace_bytes(int n_hidden, const int *Co, const int *Ci, long B, int C0, int N, int need_gx);
int dfepe_est_forward(const float *x, long x_stride_b, long x_stride_c, long B, int C0, int N, int n_hidden, const float *const *W, const float *const *gamma,
                      const float *const *beta, const int *Co, const int *Ci, const float *w_head, const float *b_head, float eps,
                      float slope, void *saved, int need_gx, void *workspace, const void *prep, float *logits, void *stream);
int dfepe_est_backward(const float *g_logits, long B, int C0, int N, int n_hidden, const float *const *W, const float *const *gamma,
                       const float *const *beta, const int *Co, const int *Ci, const float *w_head, float slope, const void *saved,
                       void *workspace, const void *prep, float *const *g_W, float *const *g_bias, float *const *g_gamma,
                       float *const *g_beta, float *g_w_head, float *g_b_head, float *gx, long gx_stride_b, long gx_stride_c, void *stream);
int dfepe_est_head_fwd(const void *planes, size_t plane_stride, int C, int ncols, const float *w, const float *bias, float *logits,
                       void *stream);
int dfepe_est_head_dw(const void *planes, size_t plane_stride, int C, int ncols, int blocks, const float *dlogit, float *part,
                      float *bias_part, void *stream);

/*
 * Match construction: what produces the [B,N,4] correspondences of dfepe_w8pt_fwd.
 * Replaces the per-pair host loop of get_matches_from_SP (deepFEPE/train_good_utils.py:683-716):
 *   SP_tracker.nn_match_two_way(desc1.T, desc2.T, nn_thresh)  (:687-691; PointTracker of the un-vendored `superpoint`
 *   package, eric-yyjau/pytorch-superpoint models/model_wrap.py) -> dfepe_nn_match_two_way, batched over the pairs;
 *   xs / offsets / quality gathered through crop_or_pad_choice (:693-716) -> dfepe_gather_matches.
 *
 *   desc1 [B,N1,D], desc2 [B,N2,D]  unit-norm descriptors, fp32, row-major (the reference's pts_desc layout,
 *                                   train_good_utils.py:735), 16-byte aligned; D a multiple of 32 (else UNSUPPORTED)
 *   nn_thresh                       keep a match when its L2 distance sqrt(2 - 2 clip(d1.d2,-1,1)) is < nn_thresh
 *                                   (negative: INVALID_ARG, the reference raises ValueError)
 *   workspace                       dfepe_nn_match_workspace_bytes(B,N1,N2) bytes, 8-byte aligned, contents irrelevant
 *   m_idx1, m_idx2 [B,N1] int32, score [B,N1] fp32: the first count[b] entries of row b are the matches of pair b in
 *                                   increasing m_idx1 order (rows 0,1,2 of the reference's [3,n] array); the rest is
 *                                   not written.   count [B] int32.
 * dmat is never materialised: fp32 MFMA tiles (exact fp32 products, arg-min ties resolved to the first index like
 * numpy.argmin) folded into per-row / per-column minima with 64-bit atomicMin.
 */
size_t dfepe_nn_match_workspace_bytes(int B, int N1, int N2);
int dfepe_nn_match_two_way(const float *desc1, const float *desc2, int B, int N1, int N2, int D, float nn_thresh,
                           void *workspace, int *m_idx1, int *m_idx2, float *score, int *count, void *stream);
/*
 *   pts1 [B,N1,2], pts2 [B,N2,2] keypoints; off1, off2 same shapes (sub-pixel offsets; both NULL with offsets NULL)
 *   choice [B,n_out] int32         positions into the match list of each pair (crop_or_pad_choice, utils_misc.py:139-161;
 *                                   drawn on the host from numpy's RNG like the reference); every value < count[b]
 *   xs [B,n_out,4] = (pts1[m_idx1[c]], pts2[m_idx2[c]]); offsets [B,n_out,4] likewise (may be NULL);
 *   quality [B,n_out] = score[c] (may be NULL)                                   (train_good_utils.py:698-716)
 */
int dfepe_gather_matches(const float *pts1, const float *pts2, const float *off1, const float *off2, int B, int N1, int N2,
                         const int *m_idx1, const int *m_idx2, const float *score, const int *choice, int n_out,
                         float *xs, float *offsets, float *quality, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DFEPE_H */
