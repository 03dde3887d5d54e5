#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the *reference itself*.

Runs only in the build container (needs /root/reference, which never travels to the GPU box).
The reference is imported unmodified with stub modules for packages the image lacks
(cv2, pebble, superpoint); nothing from /root/reference is copied:
only input/output arrays are written.  Outputs that pass through a cv2 stub are stored under
keys prefixed ``stubcv2_`` and are NOT used to pin parity (OpenCV arithmetic is unpinned).

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz
"""
import contextlib
import importlib.util
import io
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"


def _load_synth():
    spec = importlib.util.spec_from_file_location("dfepe_synth", os.path.join(REPO, "pytorch-deepfepe_amd", "synth.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


synth = _load_synth()


# ---------------------------------------------------------------- stubs
def _rodrigues_stub(R):
    R = np.asarray(R, dtype=np.float64)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    n = np.linalg.norm(v)
    ang = math.atan2(n, np.trace(R) - 1.0)
    axis = v / n if n > 0 else np.zeros(3)
    return (axis * ang).reshape(3, 1), None


def _triangulate_stub(P1, P2, x1, x2):
    N = x1.shape[1]
    X = np.zeros((4, N))
    for i in range(N):
        A = np.stack((x1[0, i] * P1[2] - P1[0], x1[1, i] * P1[2] - P1[1], x2[0, i] * P2[2] - P2[0], x2[1, i] * P2[2] - P2[1]))
        X[:, i] = np.linalg.svd(A)[2][-1]
    return X


def install_stubs():
    cv2 = types.ModuleType("cv2")
    cv2.Rodrigues = _rodrigues_stub
    cv2.triangulatePoints = _triangulate_stub
    sys.modules["cv2"] = cv2
    pebble = types.ModuleType("pebble")
    pebble.ProcessPool = object
    sys.modules["pebble"] = pebble
    sp = types.ModuleType("superpoint")
    spu = types.ModuleType("superpoint.utils")
    spl = types.ModuleType("superpoint.utils.logging")
    spuu = types.ModuleType("superpoint.utils.utils")
    for n in ("tensor2array", "save_checkpoint", "load_checkpoint", "save_path_formatter", "flattenDetection"):
        setattr(spuu, n, lambda *a, **k: None)
    sys.modules.update({"superpoint": sp, "superpoint.utils": spu, "superpoint.utils.logging": spl, "superpoint.utils.utils": spuu})
    sys.path[:0] = [REF, os.path.join(REF, "deepFEPE")]
    torch.Tensor.cuda = lambda self, *a, **k: self  # DeepFNet.__init__ calls .cuda() unconditionally


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def npy(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    if isinstance(x, (list, tuple)):
        return np.stack([npy(v) for v in x])
    return np.asarray(x)


IMAGE_SIZE = [synth.IMAGE_H, synth.IMAGE_W, 3]


def main():
    install_stubs()
    with quiet():
        from deepFEPE.models.DeepFNet import DeepFNet, Fit, NormalizeAndExpand_HW
        import deepFEPE.dsac_tools.utils_F as utils_F
        import deepFEPE.dsac_tools.utils_geo as utils_geo
        import train_good_utils as tgu
    torch.set_num_threads(4)

    # ---------------- G1-G4: normaliser, Hartley, Fit, epi residual on four scene kinds, fp32 and fp64
    kinds = {
        "general": dict(seed=11, outlier_ratio=0.0, noise_px=0.5),
        "clean": dict(seed=12, outlier_ratio=0.0, noise_px=0.0),
        "outlier40": dict(seed=13, outlier_ratio=0.4, noise_px=0.5),
        "planar": dict(seed=14, outlier_ratio=0.0, noise_px=0.5, planar=True),
        "dense1000": dict(seed=15, outlier_ratio=0.2, noise_px=0.5),
    }
    fit_out = {}
    for name, kw in kinds.items():
        B, N = (8, 100) if name != "dense1000" else (2, 1000)
        sc = synth.make_scene(B, N, dtype=torch.float64, **kw)
        w64 = torch.softmax(sc["logits_layers"][0], dim=1).unsqueeze(1)  # [B,1,N]
        for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
            matches = sc["matches_xy_ori"].to(dt)
            w = w64.to(dt)
            with quiet():
                norm = NormalizeAndExpand_HW(IMAGE_SIZE, is_cuda=False)
                norm.ones_b = norm.ones_b.to(dt)
                p1, p2, T1, T2 = norm(matches)
                pts1, pts2 = p1.permute(0, 2, 1), p2.permute(0, 2, 1)
                fit = Fit(is_cuda=False)
                for a in ("ones_b", "zero_b", "T_b", "mask"):
                    setattr(fit, a, getattr(fit, a).to(dt))
                hart1, Th1 = fit.normalize(pts1, torch.ones(B, N, 1, dtype=dt))
                out, residual = fit(pts1, pts2, w)
                epi05 = utils_F.compute_epi_residual(pts1, pts2, out)
                epi002 = utils_F.compute_epi_residual(pts1, pts2, out, 0.02)
            pre = f"{name}_{tag}_"
            if tag == "f32":
                fit_out[pre + "matches"] = npy(matches)
                fit_out[pre + "weights"] = npy(w)
            fit_out[pre + "pts1"] = npy(pts1)
            fit_out[pre + "pts2"] = npy(pts2)
            fit_out[pre + "T_hw"] = npy(T1)
            fit_out[pre + "hartley1_pts"] = npy(hart1)
            fit_out[pre + "hartley1_T"] = npy(Th1)
            fit_out[pre + "out"] = npy(out)
            fit_out[pre + "residual"] = npy(residual)
            fit_out[pre + "epi_0p5"] = npy(epi05)
            fit_out[pre + "epi_0p02"] = npy(epi002)
    np.savez_compressed(os.path.join(HERE, "fit.npz"), **fit_out)

    # ---------------- G5-G9: full recurrent forward (fixed per-layer logits AND seeded estimators),
    #                  F-loss, pose loss, gradients w.r.t. the logits / estimator parameters
    pipe = {}
    for name, kw, B, N, depth in (
        ("solver", dict(seed=21, outlier_ratio=0.2, noise_px=0.5), 6, 100, 5),
        ("solver_d1", dict(seed=22, outlier_ratio=0.0, noise_px=0.5), 4, 100, 1),
    ):
        sc = synth.make_scene(B, N, dtype=torch.float32, depth_layers=depth, **kw)
        logits = sc["logits_layers"].clone().requires_grad_(True)
        with quiet():
            net = DeepFNet(depth=depth, image_size=IMAGE_SIZE, if_quality=False, is_cuda=False, if_cpu_svd=False)
            del net._modules["input_weights"], net._modules["update_weights"]
            counter = {"i": 0}

            def fixed(_x):
                i = counter["i"]
                counter["i"] += 1
                return logits[i].unsqueeze(1)

            net.input_weights = fixed
            net.update_weights = fixed
            batch = {"matches_xy_ori": sc["matches_xy_ori"], "matches_good_unique_nums": None, "t_scene_scale": None}
            outs = net(batch)
            loss_params = {"depth": depth, "clamp_at": 0.02, "if_tri_depth": False, "if_sample_loss": False,
                           "topK": 8, "matches_good_unique_nums": None}
            losses, E_ests, F_ests, _, _, _, E_layers = tgu.get_all_loss_DeepF(
                outs, sc["pts1_virt_ori"], sc["pts2_virt_ori"], sc["Ks"], loss_params, get_residual_summaries=False)
            rt = tgu.get_Rt_loss(E_layers, sc["Ks"], sc["matches_xy_ori"][:, :, :2], sc["matches_xy_ori"][:, :, 2:],
                                 sc["delta_Rtijs_4_4"], sc["qs_cam"], sc["ts_cam"], device="cpu")
            gF, = torch.autograd.grad(losses["loss_F"], logits, retain_graph=True)
            loss_q = torch.clamp(torch.stack(rt["q_l2_error_layers_list"]), 0.0, 0.1).mean()
            loss_t = torch.clamp(torch.stack(rt["t_l2_error_layers_list"]), 0.0, 0.5).mean()
            loss_qt = loss_q * 1.0 + loss_t * 0.1
            gQT, = torch.autograd.grad(loss_qt, logits)
        pre = name + "_"
        for k in ("matches_xy_ori", "Ks", "delta_Rtijs_4_4", "qs_cam", "ts_cam", "pts1_virt_ori", "pts2_virt_ori"):
            pipe[pre + k] = npy(sc[k])
        pipe[pre + "logits_layers"] = npy(logits)
        pipe[pre + "out_layers"] = npy(outs["out_layers"])
        pipe[pre + "residual_layers"] = npy(outs["residual_layers"])
        if depth > 1:
            pipe[pre + "epi_res_layers"] = npy(outs["epi_res_layers"])
        pipe[pre + "weights_layers"] = npy(outs["weights_layers"])
        pipe[pre + "F_est"] = npy(outs["F_est"])
        pipe[pre + "loss_layers"] = npy(losses["loss_layers"])
        pipe[pre + "loss_F"] = npy(losses["loss_F"])
        pipe[pre + "loss_min_layers"] = npy(losses["loss_min_layers"])
        pipe[pre + "loss_min_batch"] = npy(losses["loss_min_batch"])
        if depth > 1:
            pipe[pre + "loss_epi_res"] = npy(losses["loss_epi_res"])
        pipe[pre + "E_ests"] = npy(E_ests)
        pipe[pre + "F_ests"] = npy(F_ests)
        pipe[pre + "E_layers"] = npy(E_layers)
        pipe[pre + "q_l2_layers"] = npy(rt["q_l2_error_layers_list"])
        pipe[pre + "t_l2_layers"] = npy(rt["t_l2_error_layers_list"])
        pipe[pre + "t_l2_error_mean"] = npy(rt["t_l2_error_mean"])
        pipe[pre + "q_l2_error_mean"] = npy(rt["q_l2_error_mean"])
        pipe[pre + "t_l2_error_list"] = npy(rt["t_l2_error_list"])
        pipe[pre + "q_l2_error_list"] = npy(rt["q_l2_error_list"])  # reference stacks the *t* list here (:276)
        pipe[pre + "t_angle_layers"] = np.stack(rt["t_angle_error_layers_list"])  # math.acos path, no cv2
        pipe[pre + "stubcv2_R_angle_layers"] = np.stack(rt["R_angle_error_layers_list"])
        pipe[pre + "loss_qt"] = npy(loss_qt)
        pipe[pre + "grad_logits_lossF"] = npy(gF)
        pipe[pre + "grad_logits_lossQT"] = npy(gQT)

    # seeded estimators (the real DeepFNet module, parameters filled by synth.fill_params_deterministic)
    B, N, depth = 4, 100, 3
    sc = synth.make_scene(B, N, seed=31, outlier_ratio=0.2, dtype=torch.float32)
    with quiet():
        net = DeepFNet(depth=depth, image_size=IMAGE_SIZE, if_quality=False, is_cuda=False, if_cpu_svd=False)
    synth.fill_params_deterministic(net, seed=5)
    with quiet():
        outs = net({"matches_xy_ori": sc["matches_xy_ori"], "matches_good_unique_nums": None, "t_scene_scale": None})
        loss_params = {"depth": depth, "clamp_at": 0.02, "if_tri_depth": False, "if_sample_loss": False,
                       "topK": 8, "matches_good_unique_nums": None}
        losses, E_ests, F_ests, _, _, _, E_layers = tgu.get_all_loss_DeepF(
            outs, sc["pts1_virt_ori"], sc["pts2_virt_ori"], sc["Ks"], loss_params, get_residual_summaries=False)
        losses["loss_F"].backward()
    pre = "net_"
    for k in ("matches_xy_ori", "Ks", "pts1_virt_ori", "pts2_virt_ori"):
        pipe[pre + k] = npy(sc[k])
    pipe[pre + "state_keys"] = np.array(sorted(net.state_dict().keys()))
    pipe[pre + "param_checksum"] = np.array([float(p.double().abs().sum()) for _, p in sorted(net.named_parameters())])
    pipe[pre + "logits"] = npy(outs["logits"])
    pipe[pre + "logits_layers"] = npy(outs["logits_layers"])
    pipe[pre + "out_layers"] = npy(outs["out_layers"])
    pipe[pre + "residual_layers"] = npy(outs["residual_layers"])
    pipe[pre + "epi_res_layers"] = npy(outs["epi_res_layers"])
    pipe[pre + "weights_layers"] = npy(outs["weights_layers"])
    pipe[pre + "loss_F"] = npy(losses["loss_F"])
    pipe[pre + "E_layers"] = npy(E_layers)
    gnorms = {n: float(p.grad.double().norm()) for n, p in net.named_parameters()}
    pipe[pre + "grad_norms"] = np.array([gnorms[n] for n in sorted(gnorms)])
    pipe[pre + "grad_first_conv"] = npy(net.input_weights.fw[0].weight.grad)
    pipe[pre + "grad_last_conv_update"] = npy(net.update_weights.fw[15].weight.grad)
    np.savez_compressed(os.path.join(HERE, "pipeline.npz"), **pipe)

    # ---------------- G7, G10, G11: small geometry functions
    geo = {}
    sc = synth.make_scene(8, 64, seed=41, noise_px=0.3, dtype=torch.float64)
    Es = sc["E_gt"] + 1e-3 * torch.randn(8, 3, 3, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    R1, R2, tt, qs = [], [], [], []
    with quiet():
        for E in Es:
            Rs, ts, _ = utils_F._get_M2s(E)
            R1.append(Rs[0]); R2.append(Rs[1]); tt.append(ts[0])
            qs.append(torch.stack((utils_geo._R_to_q(Rs[0]), utils_geo._R_to_q(Rs[1]))))
    geo["E_in"] = npy(Es)
    geo["M2s_R1"], geo["M2s_R2"], geo["M2s_t"], geo["M2s_q"] = npy(R1), npy(R2), npy(tt), npy(qs)
    # quaternion branches: rotations by ~pi about x, y, z and a generic one
    Rq = []
    for ax, ang in (((1, 0, 0), 3.0), ((0, 1, 0), 3.0), ((0, 0, 1), 3.0), ((1, 1, 1), 0.4), ((1, -2, 0.5), 2.5), ((0.2, 1, -3), 1.7)):
        a = torch.tensor(ax, dtype=torch.float64)
        Rq.append(synth._expm_so3((a / a.norm() * ang).unsqueeze(0))[0])
    with quiet():
        geo["Rq_in"] = npy(Rq)
        geo["Rq_q"] = npy([utils_geo._R_to_q(R) for R in Rq])
        geo["vec_angle"] = np.array([utils_geo.vector_angle(npy(tt[i]), npy(sc["ts_cam"][i])) for i in range(8)])
        x1 = sc["matches_xy_ori"][:, :, :2]
        x2 = sc["matches_xy_ori"][:, :, 2:]
        F = sc["F_gt"]
        geo["x1"], geo["x2"], geo["F_in"], geo["K"] = npy(x1), npy(x2), npy(F), npy(sc["Ks"][0])
        geo["sym_epi_b"] = npy(utils_F._sym_epi_dist(F, x1, x2))
        geo["sym_epi_2d"] = npy(utils_F._sym_epi_dist(F[0], x1[0], x2[0]))
        geo["sampson_b"] = npy(utils_F._sampson_dist(F, x1, x2))
        geo["epi_dist_b"] = npy(torch.stack(utils_F._epi_distance(F, x1, x2)))
        geo["F_to_E"] = npy(utils_F._F_to_E(F[0], sc["Ks"][0]))
        geo["E_to_F"] = npy(utils_F._E_to_F(sc["E_gt"], sc["Ks"]))
        geo["F_from_XY"] = npy(utils_F._F_from_XY(x1[0], x2[0]))
        geo["E_from_XY"] = npy(utils_F._E_from_XY(x1[0], x2[0], sc["Ks"][0]))
        wdiag = torch.diag(torch.softmax(sc["logits_layers"][0, 0], 0))
        geo["E_from_XY_W"] = npy(utils_F._E_from_XY(x1[0], x2[0], sc["Ks"][0], W=wdiag))
        geo["W_diag"] = npy(torch.diagonal(wdiag))
        # cheirality with the DLT stub
        wins, Rts = [], []
        for b in range(8):
            _, _, Rt_cam = utils_F._E_to_M_train(sc["E_gt"][b], npy(sc["Ks"][b]), npy(x1[b]), npy(x2[b]), show_debug=False, show_result=False)
            Rts.append(npy(Rt_cam))
        geo["stubcv2_cheirality_Rt_cam"] = np.stack(Rts)
        geo["delta_Rtijs_4_4"] = npy(sc["delta_Rtijs_4_4"])
    np.savez_compressed(os.path.join(HERE, "geometry.npz"), **geo)

    with open(os.path.join(HERE, "MANIFEST.txt"), "w") as f:
        f.write("generated by tests/golden/make_golden.py from the reference at /root/reference (unmodified, cv2/pebble/superpoint stubbed)\n")
        f.write(f"torch {torch.__version__}  numpy {np.__version__}\n")
        for fn in ("fit.npz", "pipeline.npz", "geometry.npz"):
            z = np.load(os.path.join(HERE, fn))
            f.write(f"{fn}: {len(z.files)} arrays, {os.path.getsize(os.path.join(HERE, fn))} bytes\n")
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
