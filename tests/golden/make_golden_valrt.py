#!/usr/bin/env python3
"""Golden vectors that pin the reference's OWN code around the evaluation-time pose recovery:

* utils_F.goodCorr_eval_nondecompose (deepFEPE/dsac_tools/utils_F.py:909-954): the top-10 % score mask (:912-919), the
  "< 5 correspondences" fall-back (180 / 90 degrees, identity pose, :949-952), utils_geo.invert_Rt of the recovered pose and the
  rotation / translation angles against the ground-truth camera motion (:942-944);
* train_good_utils.val_rt (deepFEPE/train_good_utils.py:553-646): the estimated-E and ground-truth-E legs
  and the epi_distance_np statistics of both F (:605-614).

The reference itself runs (imported unmodified, see make_golden.py).  The ONE call that lives in OpenCV, cv2.recoverPose, goes
through a stand-in written from OpenCV's published algorithm (decomposeEssentialMat; four candidates [R1|t], [R2|t], [R1|-t],
[R2|-t]; linear triangulation of the focal/pp-normalised points; a point counts when its depth is in (0, 50) in both cameras; the
first candidate with the largest count wins), so every array here is "stubbed-cv2" for the pose itself and pinned for everything
the reference's own source decides from it.

    python tests/golden/make_golden_valrt.py      # rewrites tests/golden/valrt.npz (build container only)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (stubs, synth, quiet)


GOODS = []  # per call of the stand-in: the four in-front counts (a tie at the top is decided by the SVD gauge: not pinnable)


def _recover_pose_stub(E, points1, points2, focal=1.0, pp=(0.0, 0.0), mask=None):
    """Stand-in for cv2.recoverPose(E, p1, p2, focal=, pp=) of OpenCV 3.4 (distance threshold 50)."""
    E = np.asarray(E, dtype=np.float64)
    p1 = (np.asarray(points1, dtype=np.float64) - np.asarray(pp)) / focal
    p2 = (np.asarray(points2, dtype=np.float64) - np.asarray(pp)) / focal
    U, _, Vt = np.linalg.svd(E)
    if np.linalg.det(U) < 0:
        U = -U
    if np.linalg.det(Vt) < 0:
        Vt = -Vt
    Wm = np.array([[0.0, 1.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    R1, R2, t = U @ Wm @ Vt, U @ Wm.T @ Vt, U[:, 2:3]
    P0 = np.hstack((np.eye(3), np.zeros((3, 1))))
    dist = 50.0
    goods, masks = [], []
    for R, tt in ((R1, t), (R2, t), (R1, -t), (R2, -t)):
        P = np.hstack((R, tt))
        Q = mg._triangulate_stub(P0, P, p1.T, p2.T)
        m = (Q[2] * Q[3]) > 0
        Q = Q / Q[3]
        m &= Q[2] < dist
        Q2 = P @ Q
        m &= (Q2[2] > 0) & (Q2[2] < dist)
        goods.append(int(m.sum()))
        masks.append(m)
    cands = ((R1, t), (R2, t), (R1, -t), (R2, -t))
    GOODS.append(list(goods))
    k = int(np.argmax(goods))  # the first of the largest counts, like the >= cascade of recoverPose
    return goods[k], cands[k][0].copy(), cands[k][1].copy(), (masks[k].astype(np.uint8) * 255).reshape(-1, 1)


def main():
    mg.install_stubs()
    cv2 = sys.modules["cv2"]
    cv2.recoverPose = _recover_pose_stub
    # val_rt cannot run with if_opencv=False (M_opencv is unbound at its return, train_good_utils.py:636-646), so its OpenCV
    # baseline leg (recover_camera_opencv: findEssentialMat + RANSAC; out of this build's scope) is fed a placeholder E and its
    # three outputs are not recorded
    cv2.RANSAC = 8
    cv2.findEssentialMat = lambda x1, x2, **kw: (np.array([[0.0, -1.0, 0.1], [1.0, 0.0, -0.3], [-0.1, 0.3, 0.0]]), np.ones((x1.shape[0], 1), np.uint8))
    with mg.quiet():
        import deepFEPE.dsac_tools.utils_F as utils_F
        import train_good_utils as tgu
    out = {}
    # ---- val_rt on whole pairs: estimated E / F = perturbed ground truth (float32 like the pipeline hands them over) ----
    B, N = 10, 200
    sc = mg.synth.make_scene(B, N, seed=61, outlier_ratio=0.25, noise_px=0.5)
    g = torch.Generator().manual_seed(62)
    E_gt = sc["E_gt"] / sc["E_gt"].flatten(1).norm(dim=1)[:, None, None]
    E_est = (E_gt + torch.tensor([0.0, 1e-3, 3e-3, 1e-2, 3e-2, 1e-1, 0.3, 1.0, 0.0, 2e-3])[:, None, None] * torch.randn(B, 3, 3, generator=g)).float()
    K = sc["Ks"].float()
    Ki = torch.linalg.inv(K.double())
    F_est = (Ki.transpose(1, 2) @ E_est.double() @ Ki).float()
    F_gt = sc["F_gt"].float()
    m = sc["matches_xy_ori"].float()
    delta = sc["delta_Rtijs_4_4"].float()
    keys = ("err_est", "epi_est", "err_gt", "epi_gt", "M_est")
    res = {k: [] for k in keys}
    goods_est, goods_gt = [], []
    for b in range(B):
        del GOODS[:]
        with mg.quiet():
            r = tgu.val_rt(b, K[b].numpy(), m[b, :, :2].numpy(), m[b, :, 2:].numpy(), E_est[b].numpy(), E_gt[b].float().numpy(), F_est[b].numpy(),
                           F_gt[b].numpy(), delta[b].numpy(), five_point=True, if_opencv=True)
        assert r[6] == b and len(GOODS) == 3  # estimated E, ground-truth E, the OpenCV-baseline leg
        goods_est.append(GOODS[0]); goods_gt.append(GOODS[1])
        res["err_est"].append(np.array(r[0], dtype=np.float64))
        res["epi_est"].append(np.asarray(r[1], dtype=np.float64))
        res["err_gt"].append(np.array(r[4], dtype=np.float64))
        res["epi_gt"].append(np.asarray(r[5], dtype=np.float64))
        res["M_est"].append(np.asarray(r[7], dtype=np.float64))
    out.update({"valrt_K": mg.npy(K), "valrt_matches": mg.npy(m), "valrt_E_est": mg.npy(E_est), "valrt_E_gt": mg.npy(E_gt.float()),
                "valrt_F_est": mg.npy(F_est), "valrt_F_gt": mg.npy(F_gt), "valrt_delta": mg.npy(delta)})
    for k in keys:
        out["valrt_" + k] = np.stack(res[k])
    out["valrt_counts_est"], out["valrt_counts_gt"] = np.array(goods_est), np.array(goods_gt)
    print("val_rt counts", goods_est, goods_gt)
    print("val_rt err_est", np.round(out["valrt_err_est"], 3).tolist())

    # ---- goodCorr_eval_nondecompose: score mask, too few correspondences --------------------------------------------
    Bs, Ns = 6, 120
    sc = mg.synth.make_scene(Bs, Ns, seed=63, outlier_ratio=0.3, noise_px=0.5)
    g = torch.Generator().manual_seed(64)
    E = (sc["E_gt"] / sc["E_gt"].flatten(1).norm(dim=1)[:, None, None] + 2e-3 * torch.randn(Bs, 3, 3, generator=g)).float()
    m = sc["matches_xy_ori"].float()
    K = sc["Ks"].float()
    scores = torch.rand(Bs, Ns, generator=g)
    scores[:, int(0.3 * Ns):] += 1.0  # the synthetic outliers are the first 30 %: the top decile is inliers only
    scores = scores.float().numpy()
    scores[1, 80:] = scores[1].max()  # ties at the threshold: `>=` keeps all 40 of them (:914-915)
    Ms, errs, kept, goods = [], [], [], []
    for b in range(Bs):
        del GOODS[:]
        dinv = np.linalg.inv(sc["delta_Rtijs_4_4"][b].float().numpy())[:3]
        with mg.quiet():
            Mb, eb = utils_F.goodCorr_eval_nondecompose(m[b, :, :2].numpy(), m[b, :, 2:].numpy(), E[b].numpy().astype(np.float64), dinv, K[b].numpy(), scores[b])
        Ms.append(np.asarray(Mb, dtype=np.float64)); errs.append(np.array(eb, dtype=np.float64))
        num_top = max(1, Ns // 10)
        kept.append(int((scores[b] >= np.sort(scores[b])[::-1][num_top]).sum()))
        goods.append(GOODS[0])
    out.update({"scores_K": mg.npy(K), "scores_matches": mg.npy(m), "scores_E": mg.npy(E), "scores_scores": scores,
                "scores_delta": mg.npy(sc["delta_Rtijs_4_4"].float()), "scores_M": np.stack(Ms), "scores_err": np.stack(errs), "scores_kept": np.array(kept),
                "scores_counts": np.array(goods)})
    print("scores kept", kept, "err", np.round(out["scores_err"], 3).tolist())
    few_M, few_err, few_goods = [], [], []
    bb = int(sc["delta_Rtijs_4_4"][:, :3, 3].norm(dim=1).argmax())  # the longest baseline: depths well inside recoverPose's 50-baseline bound
    for n in (0, 3, 4, 5, 9):
        dinv = np.linalg.inv(sc["delta_Rtijs_4_4"][bb].float().numpy())[:3]
        del GOODS[:]
        with mg.quiet():
            Mb, eb = utils_F.goodCorr_eval_nondecompose(m[bb, 40:40 + n, :2].numpy(), m[bb, 40:40 + n, 2:].numpy(), E[bb].numpy().astype(np.float64), dinv, K[bb].numpy(), None)
        few_M.append(np.asarray(Mb, dtype=np.float64)); few_err.append(np.array(eb, dtype=np.float64))
        few_goods.append(GOODS[0] if GOODS else [0, 0, 0, 0])
    out.update({"few_n": np.array([0, 3, 4, 5, 9]), "few_pair": np.array(bb), "few_M": np.stack(few_M), "few_err": np.stack(few_err), "few_counts": np.array(few_goods)})
    print("few err", np.stack(few_err).tolist(), few_goods, "scores counts", goods)
    np.savez_compressed(os.path.join(HERE, "valrt.npz"), **out)
    with open(os.path.join(HERE, "MANIFEST.txt"), "a") as f:
        f.write(f"valrt.npz: {len(out)} arrays, {os.path.getsize(os.path.join(HERE, 'valrt.npz'))} bytes (make_golden_valrt.py: the reference's "
                "val_rt / goodCorr_eval_nondecompose with a stand-in for cv2.recoverPose: own-source logic pinned, recoverPose stubbed-cv2)\n")


if __name__ == "__main__":
    main()
