"""Mirror of the hot-path functions of deepFEPE/dsac_tools/utils_F.py, evaluated by libdfepe_hip.so.
Signatures follow the reference; tensors must live on the GPU."""
import numpy as np
import torch

from .. import _lib, ops
from . import _autograd_paths as _ap
from . import utils_misc


def _gpu(t, dtype=torch.float32):
    t = torch.as_tensor(t)
    if not t.is_cuda:
        raise _lib.DfepeError("dsac_tools.utils_F mirror: tensors must live on the GPU (no CPU implementation here)")
    return t.to(dtype)


def _normalize_XY(X, Y):
    """Hartley normalisation of two point sets [N,2] with scale sqrt(2) (np.sqrt(2), not Fit.normalize's 1.4142):
    returns (X_normalized [N,2], Y_normalized [N,2], T1, T2) like utils_F.py:15-41.  Elementwise plumbing; the solvers
    (_F_from_XY, _E_from_XY) fuse this step into the fit kernel and do not call it."""
    if X.shape[0] != Y.shape[0]:
        raise ValueError("Number of points don't match.")

    def one(P):
        Ph = utils_misc._homo(P)
        c = Ph[:, :2].mean(0, keepdim=True)
        s = (2.0 ** 0.5) / torch.norm(Ph[:, :2] - c, 2, dim=1).mean()
        T = torch.zeros(3, 3, dtype=P.dtype, device=P.device)
        T[0, 0] = s; T[1, 1] = s; T[2, 2] = 1.0; T[0, 2] = -s * c[0, 0]; T[1, 2] = -s * c[0, 1]
        return utils_misc._de_homo((T @ Ph.t()).t()), T

    Xn, T1 = one(X)
    Yn, T2 = one(Y)
    return Xn, Yn, T1, T2


def _normalize_XY_batch(X, Y):
    """Batched form, X, Y [B,N,2] -> (Xn, Yn [B,N,2], T1s, T2s [B,3,3]) (utils_F.py:43-69)."""
    if X.shape[1] != Y.shape[1]:
        raise ValueError("Number of points don't match.")

    def one(P):
        Ph = utils_misc._homo(P)
        c = Ph[:, :, :2].mean(1, keepdim=True)
        s = (2.0 ** 0.5) / torch.norm(Ph[:, :, :2] - c, 2, dim=2).mean(1)
        T = torch.zeros(P.shape[0], 3, 3, dtype=P.dtype, device=P.device)
        T[:, 0, 0] = s; T[:, 1, 1] = s; T[:, 2, 2] = 1.0; T[:, 0, 2] = -s * c[:, 0, 0]; T[:, 1, 2] = -s * c[:, 0, 1]
        return utils_misc._de_homo(torch.bmm(T, Ph.transpose(1, 2)).transpose(1, 2)), T

    Xn, T1s = one(X)
    Yn, T2s = one(Y)
    return Xn, Yn, T1s, T2s


def compute_epi_residual(pts1, pts2, F, clamp_at=0.5):
    """Symmetric epipolar residual, [B,N] (utils_F.py:400-413).  The kernel (with its adjoint w.r.t. F: what the F-loss needs; the
    recurrent model's in-loop residual with point gradients comes out of the fit itself, ops.w8pt); when the POINTS require grad,
    the same formula through torch autograd (_autograd_paths)."""
    if _ap.wants_grad(pts1, pts2):
        return _ap.epi_residual(_gpu(pts1), _gpu(pts2), _gpu(F), clamp_at)
    return ops.epi_residual(_gpu(pts1), _gpu(pts2), _gpu(F), clamp_at)


def _get_M2s(E):
    """E [3,3] -> (R2s, t2s, M2s) like utils_F.py:478-498 (two rotations, +-t, the four [R|t]).
    The candidate *set* equals the reference's; the order within each pair follows this library's SVD gauge (with an E that
    requires grad: torch.linalg.svd's, through the differentiable evaluation of _autograd_paths -- the pose loss on the hot path has
    its own adjoint, ops.pose_errors)."""
    if _ap.wants_grad(E):
        R1, R2, t = _ap.decompose_essential(_gpu(E))
        R2s, t2s = [R1, R2], [t, -t]
        return R2s, t2s, [torch.cat((R, tt), 1) for R in R2s for tt in t2s]
    R1, R2, t = ops.decompose_essential(_gpu(E).reshape(1, 3, 3))
    R2s = [R1[0], R2[0]]
    t2s = [t[0].reshape(3, 1), -t[0].reshape(3, 1)]
    M2s = [torch.cat((R, tt), 1) for R in R2s for tt in t2s]
    return R2s, t2s, M2s


def _F_to_E(F, K):
    """E = K^T F K with singular values forced to (1,1,0) (utils_F.py:455-462)."""
    F, K = _gpu(F), _gpu(K)
    return ops.project_essential((K.t() @ F @ K).reshape(1, 3, 3))[0]


def _E_to_F(E, K):
    """F = K^-T E K^-1 (utils_F.py:464-469); 2-D or batched."""
    E, K = _gpu(E), _gpu(K)
    Ki = torch.linalg.inv(K)
    return Ki.transpose(-1, -2) @ E @ Ki


def E_to_F_np(E, K):
    """F = K^-T E K^-1 in numpy, single [3,3] or batched [B,3,3] (utils_F.py:471-476).  The reference's batched branch drops its
    result (no assignment, :475) and then fails on the return; here it returns the value that line computes, with the
    transposition written as the batched transpose it means."""
    E, K = np.asarray(E), np.asarray(K)
    Kinv = np.linalg.inv(K)
    if E.ndim == 2:
        return Kinv.T @ E @ Kinv
    return np.transpose(Kinv, (0, 2, 1)) @ E @ Kinv


def _E_F_from_Rt(R_th, t_th, K_th, tensor_input=False):
    """(E, F) of a relative pose, E = [t]_x R, F = K^-T E K^-1; numpy inputs are promoted to float64 tensors unless
    ``tensor_input`` (utils_F.py:820-833).  R [3,3] / t [3,1] / K [3,3] or batched [B,...].  Plain small torch ops on whatever
    device the inputs live on, differentiable like the reference's."""
    from . import utils_misc

    if not tensor_input:
        K_th = torch.from_numpy(np.asarray(K_th)).to(torch.float64)
        R_th = torch.from_numpy(np.asarray(R_th)).to(torch.float64)
        t_th = torch.from_numpy(np.asarray(t_th)).to(torch.float64)
    E_gt_th = utils_misc._skew_symmetric(t_th) @ R_th
    Kinv = torch.inverse(K_th)
    if R_th.dim() == 2:
        F_gt_th = torch.matmul(torch.matmul(Kinv.t(), E_gt_th), Kinv)
    else:
        F_gt_th = Kinv.transpose(1, 2) @ E_gt_th @ Kinv
    return E_gt_th, F_gt_th


def E_F_from_Rt_np(R, t, K):
    """numpy twin of _E_F_from_Rt (utils_F.py:835-846): the ground-truth convention of the data sets.  The
    reference's batched branch calls ndarray.transpose(1, 2) on a 3-D array (:845), which raises; the batched transpose it
    means is used here."""
    from . import utils_misc

    R, t, K = np.asarray(R), np.asarray(t), np.asarray(K)
    E_gt = utils_misc.skew_symmetric_np(t) @ R
    Kinv = np.linalg.inv(K)
    if R.ndim == 2:
        F_gt = Kinv.T @ E_gt @ Kinv
    else:
        F_gt = np.transpose(Kinv, (0, 2, 1)) @ E_gt @ Kinv
    return E_gt, F_gt


def _epi_args(what, F, X, Y, if_homo):
    F, X, Y = _gpu(F), _gpu(X), _gpu(Y)
    single = X.dim() == 2
    if single:
        F, X, Y = F.unsqueeze(0), X.unsqueeze(0), Y.unsqueeze(0)
    want = 3 if if_homo else 2  # if_homo=True: [.,N,3] homogeneous points, used as they are (the reference skips _homo then)
    if X.shape[-1] != want or Y.shape[-1] != want:
        raise ValueError(f"{what}: points must be [..., N, {want}] when if_homo={if_homo}")
    return F, X.contiguous(), Y.contiguous(), single


def _sym_epi_dist(F, X, Y, if_homo=False, clamp_at=None):
    """Squared symmetric epipolar distance (utils_F.py:310-339); the 1e-10 guard only in the batched form (:329)."""
    if _ap.wants_grad(F, X, Y):  # the reference's gradients (legacy callers: train_good_utils.py:55-61), see _autograd_paths
        return _ap.sym_epi_dist(_gpu(F), _gpu(X), _gpu(Y), if_homo, clamp_at, eps=0.0 if torch.as_tensor(X).dim() == 2 else 1e-10)
    F, X, Y, single = _epi_args("_sym_epi_dist", F, X, Y, if_homo)
    out = ops.epi_metrics(0, F, X, Y, clamp_at=clamp_at, eps=0.0 if single else 1e-10)
    return out[0] if single else out


def _sampson_dist(F, X, Y, if_homo=False):
    if _ap.wants_grad(F, X, Y):  # dsac_tools/dsac.py:138-176 differentiates the soft inlier count through this
        return _ap.sampson_dist(_gpu(F), _gpu(X), _gpu(Y), if_homo)
    F, X, Y, single = _epi_args("_sampson_dist", F, X, Y, if_homo)
    out = ops.epi_metrics(1, F, X, Y)
    return out[0] if single else out


def _epi_distance(F, X, Y, if_homo=False):
    """Returns ((d1+d2)/2, d1, d2) (utils_F.py:341-361)."""
    if _ap.wants_grad(F, X, Y):
        return _ap.epi_distance(_gpu(F), _gpu(X), _gpu(Y), if_homo)
    F, X, Y, single = _epi_args("_epi_distance", F, X, Y, if_homo)
    out = ops.epi_metrics(2, F, X, Y)
    return (out[0, 0], out[1, 0], out[2, 0]) if single else (out[0], out[1], out[2])


def _E_to_M_train(E_est_th, K, x1, x2, inlier_mask=None, delta_Rt_gt_cam=None, depth_thres=50.0, show_debug=False,
                  show_result=True, method_name="ours"):
    """Cheirality-checked pose (utils_F.py:679-763).  Returns (M2_list, error_Rt, Rt_cam) like the reference:
    Rt_cam [3,4] = camera motion of the winning candidate or None; error_Rt = [R deg, t deg] when a ground truth
    is given; M2_list is the empty list the reference returns too (it is created at :699 and never filled).
    The triangulation is a linear DLT per correspondence (the reference calls cv2.triangulatePoints)."""
    dev = E_est_th.device if torch.is_tensor(E_est_th) and E_est_th.is_cuda else torch.device("cuda")
    E = torch.as_tensor(E_est_th, dtype=torch.float32).to(dev).reshape(1, 3, 3)
    Kt = torch.as_tensor(K, dtype=torch.float32).to(dev).reshape(1, 3, 3)
    x1 = torch.as_tensor(x1, dtype=torch.float32).to(dev)
    x2 = torch.as_tensor(x2, dtype=torch.float32).to(dev)
    if inlier_mask is not None:
        m = torch.as_tensor(inlier_mask).to(dev)
        x1, x2 = x1[m], x2[m]
        if x1.shape[0] < 8:
            print("ERROR! Less than 8 points after inlier mask!")
            return None
    Rt, win, cnt = ops.cheirality(E, Kt, torch.cat((x1, x2), 1).unsqueeze(0).contiguous(), depth_thres)
    if int(win[0].item()) < 0:
        print("ERROR! 0 of qualified [R|t] found!")
        return [], [], None
    Rt_cam = Rt[0]
    error_Rt = []
    if delta_Rt_gt_cam is not None:
        gt = torch.as_tensor(delta_Rt_gt_cam, dtype=torch.float32).to(dev)
        eR = ops.rot_angle_deg(Rt_cam[:, :3].reshape(1, 3, 3), gt[:3, :3].reshape(1, 3, 3))[0].item()
        et = ops.vector_angle_deg(Rt_cam[:, 3].reshape(1, 3), gt[:3, 3].reshape(1, 3))[0].item()
        if show_result:
            print("Recovered by %s (camera): The rotation error (degree) %.4f, and translation error (degree) %.4f" % (method_name, eR, et))
        error_Rt = [eR, et]
    return [], error_Rt, Rt_cam


def _E_to_M(E_est_th, K, x1, x2, inlier_mask=None, delta_Rt_gt=None, depth_thres=50.0, show_debug=False, show_result=True,
            method_name="ours"):
    """utils_F._E_to_M (utils_F.py:521-677): same selection as _E_to_M_train, ground truth given as camera motion."""
    return _E_to_M_train(E_est_th, K, x1, x2, inlier_mask=inlier_mask, delta_Rt_gt_cam=delta_Rt_gt, depth_thres=depth_thres,
                         show_debug=show_debug, show_result=show_result, method_name=method_name)


def goodCorr_eval_nondecompose(p1s, p2s, E_hat, delta_Rtij_inv, K, scores, if_my_decomp=False):
    """Evaluation-time pose from E (utils_F.py:909-954): the reference calls cv2.recoverPose (cheirality inside OpenCV,
    50-unit distance threshold), inverts the pose and measures the rotation / translation angle against the ground-truth
    camera motion.  Here the cheirality kernel does the selection (same candidates, DLT triangulation, threshold 50).
    Returns (np.hstack((R, t)) in the scene convention x2 ~ R x1 + t, (err_q_deg, err_t_deg)); the failure fall-backs
    (fewer than 5 points: 180 / 90 degrees and the identity pose) follow the reference (:942-952)."""
    import numpy as np

    p1s, p2s = np.asarray(p1s), np.asarray(p2s)
    if scores is not None:
        scores = np.asarray(scores)
        num_top = max(1, len(scores) // 10)
        th = np.sort(scores)[::-1][num_top]
        mask = scores >= th
        p1s, p2s = p1s[mask], p2s[mask]
    if p1s.shape[0] < 5:
        return np.hstack((np.eye(3, dtype=np.float32), np.zeros((3, 1), np.float32))), (180.0, 90.0)
    dev = torch.device("cuda")
    gt = torch.as_tensor(np.asarray(delta_Rtij_inv), dtype=torch.float32, device=dev)
    m = torch.as_tensor(np.hstack((p1s, p2s)), dtype=torch.float32, device=dev).unsqueeze(0).contiguous()
    Rt, win, _ = ops.cheirality(torch.as_tensor(np.asarray(E_hat), dtype=torch.float32, device=dev).reshape(1, 3, 3),
                                torch.as_tensor(np.asarray(K), dtype=torch.float32, device=dev).reshape(1, 3, 3), m, 50.0)
    if int(win[0].item()) < 0:
        print("Failed in evaluation")
        return np.hstack((np.eye(3, dtype=np.float32), np.zeros((3, 1), np.float32))), (180.0, 90.0)
    R_cam, t_cam = Rt[0, :, :3], Rt[0, :, 3]
    err_q = ops.rot_angle_deg(R_cam.reshape(1, 3, 3), gt[:3, :3].reshape(1, 3, 3))[0].item()
    err_t = ops.vector_angle_deg(t_cam.reshape(1, 3), gt[:3, 3].reshape(1, 3))[0].item()
    R = R_cam.t()
    t = -(R @ t_cam.reshape(3, 1))
    return torch.cat((R, t), 1).cpu().numpy(), (err_q, err_t)


def _get_M2s_batch(Es_batch):
    """Batched four-fold decomposition (utils_F.py:500-519; the reference needs the external batch_svd CUDA extension).
    Returns ([R1 [B,3,3], R2 [B,3,3]], [t [B,3,1], -t])."""
    R1, R2, t = ops.decompose_essential(_gpu(Es_batch))
    t = t.unsqueeze(-1)
    return [R1, R2], [t, -t]


def epi_distance_np(F, X, Y, if_homo=False):
    """numpy-in / numpy-out evaluation metric (utils_F.py:363-385): returns (d1 + d2, d1, d2), not squared."""
    import numpy as np

    dev = torch.device("cuda")
    Fm = torch.as_tensor(np.asarray(F), dtype=torch.float32, device=dev)
    Xt = torch.as_tensor(np.asarray(X), dtype=torch.float32, device=dev)
    Yt = torch.as_tensor(np.asarray(Y), dtype=torch.float32, device=dev)
    _, d1, d2 = _epi_distance(Fm, Xt, Yt, if_homo=if_homo)
    d1, d2 = d1.cpu().numpy(), d2.cpu().numpy()
    return d1 + d2, d1, d2


def _diag_weights(W, N):
    """The reference left-multiplies the design matrix by W [N,N] (utils_F.py:129-130); its callers pass torch.diag(w)
    (train_good_utils.get_E_ests): a per-correspondence weight, which the fit kernel takes from the points.  Returns the weight
    vector for a vector or a diagonal matrix, None for a dense matrix (served by _dense_w_solve)."""
    W = _gpu(W)
    if W.dim() == 1:
        return W
    off = W - torch.diag(torch.diagonal(W))
    return torch.diagonal(W) if float(off.abs().max()) == 0.0 else None


def _dense_w_solve(X, Y, W, essential, normalize):
    """Dense W [N,N] (utils_F.py:129-130,245-246): rows of W @ XX are mixtures of correspondences, so the design matrix is formed
    explicitly (elementwise plumbing + one [N,N]x[N,9] product) and handed to the explicit-rows closing kernel
    (dfepe_w8pt_rows_fwd: null vector, 3x3 step, T2^T F T1)."""
    T1 = T2 = None
    if normalize:
        X, Y, T1, T2 = _normalize_XY(X, Y)
    x1, y1, x2, y2 = X[:, 0], X[:, 1], Y[:, 0], Y[:, 1]
    XX = torch.stack((x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, torch.ones_like(x1)), dim=1)  # [N,9] (:122-127)
    rows = torch.mm(_gpu(W), XX).unsqueeze(0).contiguous()
    return ops.eight_point_rows(rows, None if T1 is None else T1.unsqueeze(0), None if T2 is None else T2.unsqueeze(0), essential)[0]


def _F_from_XY(X, Y, W=None, normalize=True, show_debug=False):
    """Normalised 8-point fundamental matrix from X, Y [N,2] (utils_F.py:223-275); sign follows this library's gauge (with inputs
    that require grad: torch.linalg.svd's, through the differentiable evaluation of _autograd_paths)."""
    X, Y = _gpu(X), _gpu(Y)
    if _ap.wants_grad(X, Y, W):
        return _ap.eight_point(X, Y, None if W is None else _gpu(W), essential=False, normalize=normalize)
    w = None if W is None else _diag_weights(W, X.shape[0])
    if W is not None and w is None:
        return _dense_w_solve(X, Y, W, False, normalize)
    return ops.eight_point(X.unsqueeze(0), Y.unsqueeze(0), None if w is None else w.unsqueeze(0), essential=False, normalize=normalize)[0]


def _E_from_XY_batch(X, Y, K, W=None, if_normzliedK=False, normalize=True, show_debug=False):
    """Batched _E_from_XY (utils_F.py:157-221): X, Y [B,N,2], K [B,3,3]; like the reference it returns the NEGATED matrix
    (:221).  W: per-correspondence weights [B,N] (the reference takes dense [B,N,N] matrices; only diagonals are built)."""
    X, Y = _gpu(X), _gpu(Y)
    if not if_normzliedK:
        Ki = torch.linalg.inv(_gpu(K))
        ones = torch.ones(X.shape[0], X.shape[1], 1, device=X.device)
        Xh, Yh = torch.cat((X, ones), 2) @ Ki.transpose(1, 2), torch.cat((Y, ones), 2) @ Ki.transpose(1, 2)
        X, Y = Xh[..., :2] / (Xh[..., 2:3] + 1e-10), Yh[..., :2] / (Yh[..., 2:3] + 1e-10)
    w = None
    if W is not None:
        W = _gpu(W)
        w = torch.diagonal(W, dim1=1, dim2=2) if W.dim() == 3 else W
    return -ops.eight_point(X.contiguous(), Y.contiguous(), w, essential=True, normalize=normalize)


def _E_from_XY(X, Y, K, W=None, if_normzliedK=False, normalize=True, show_debug=False):
    """Normalised 8-point essential matrix (utils_F.py:104-155): K^-1 points, sqrt(2) Hartley, singular values (1,1,0)."""
    X, Y = _gpu(X), _gpu(Y)
    if not if_normzliedK:
        Ki = torch.linalg.inv(_gpu(K))
        ones = torch.ones(X.shape[0], 1, device=X.device)
        Xh, Yh = torch.cat((X, ones), 1) @ Ki.t(), torch.cat((Y, ones), 1) @ Ki.t()
        X, Y = Xh[:, :2] / (Xh[:, 2:3] + 1e-10), Yh[:, :2] / (Yh[:, 2:3] + 1e-10)  # _de_homo (utils_misc.py:69-78)
    if _ap.wants_grad(X, Y, W):  # the reference's own differentiable route (legacy callers, dsac.py:138-176)
        return _ap.eight_point(X, Y, None if W is None else _gpu(W), essential=True, normalize=normalize)
    w = None if W is None else _diag_weights(W, X.shape[0])
    if W is not None and w is None:
        return _dense_w_solve(X, Y, W, True, normalize)
    return ops.eight_point(X.unsqueeze(0), Y.unsqueeze(0), None if w is None else w.unsqueeze(0), essential=True, normalize=normalize)[0]
