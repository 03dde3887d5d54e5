"""CPU-side checks: the C-ABI library loads and exports every symbol include/dfepe.h declares, host-side argument
validation, no compute launches (no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "dfepe.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dfepe_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol(dfepe):
    import __graft_entry__

    __graft_entry__.build()
    lib = ctypes.CDLL(dfepe.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 11
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/dfepe.h but not exported"
    assert sorted(dfepe.EXPORTED_SYMBOLS) == declared  # the ctypes table covers exactly the header


def test_version_strerror_and_save_layout(dfepe):
    L = dfepe._lib.lib()
    assert L.dfepe_version() == 154
    assert L.dfepe_save_floats() == 128
    assert L.dfepe_strerror(0) == b"ok"
    assert b"invalid" in L.dfepe_strerror(-1)
    assert b"unknown" in L.dfepe_strerror(-99)


def test_argument_validation_without_launching(dfepe):
    """Bad arguments are rejected on the host before any HIP call (safe without a GPU)."""
    L = dfepe._lib.lib()
    assert L.dfepe_w8pt_fwd(None, None, None, 4, 100, 1, 0, 0.0, 0.0, 0.5, None, None, None, None, None, None) == -1
    assert L.dfepe_w8pt_fwd(None, None, None, -1, 100, 1, 0, 0.0, 0.0, 0.5, None, None, None, None, None, None) == -1
    assert L.dfepe_w8pt_fwd(None, None, None, 0, 100, 1, 0, 0.0, 0.0, 0.5, None, None, None, None, None, None) == 0  # empty batch
    assert L.dfepe_w8pt_bwd(None, None, None, 2, 0, 1, 0, 0.0, 0.0, 0.5, None, None, None, None, None, None, None, None, None, None, None, None) == -1
    assert L.dfepe_w8pt_fwd(None, None, None, 4, 100, 1, 1 << 9, 0.0, 0.0, 0.5, None, None, None, None, None, None) == -1  # unknown flag bit
    assert L.dfepe_floss_fwd(None, 0, 4, None, None, 0, None, None, None, 100, 0.02, None, None, None) == -1
    assert L.dfepe_floss_fwd(None, 5, 0, None, None, 0, None, None, None, 100, 0.02, None, None, None) == 0
    assert L.dfepe_floss_fwd(None, 5, 4, None, None, 3, None, None, None, 100, 0.02, None, None, None) == -1  # bad stride
    assert L.dfepe_pose_fwd(None, 5, 4, None, None, None, None, None, None, None, None, None) == -1
    assert L.dfepe_pose_bwd(None, 5, 0, None, None, None, None, 0.0, 0.0, 0.0, 0.0, None, None, None) == 0
    assert L.dfepe_loss_head(None, None, None, 5, 4, 100, 0.1, 0.5, 1.0, 0.1, None, None, None) == -1
    tail = lambda L_, B_, M_: L.dfepe_loss_tail(None, L_, B_, None, None, 0, None, None, None, M_, 0.02, None, None, None, 0.1, 0.5, 1.0, 1.0, 0.1, 4.0,
                                               None, None, None, None, None, None, None, None, None, None, None, 0, None)
    assert tail(5, 4, 100) == -1 and tail(0, 4, 100) == -1 and tail(5, 4, 200) == -3  # null pointers; no layers; grid too large for the fused kernel
    assert L.dfepe_loss_tail_workspace_bytes(4096) >= 256 * 48 * 8 + 256  # partials + the descriptor slot of a deferred head
    # the round-4 entry points behind the reference's call sequence
    jac = lambda L_, B_, M_: L.dfepe_loss_tail_jac(None, L_, B_, None, None, 0, None, None, None, M_, 0.02, None, None, None, 1, None, None, None, None,
                                                  None, None, None, None, None)
    assert jac(5, 4, 100) == -1 and jac(17, 4, 100) == -1 and jac(5, 4, 113) == -3  # null pointers; too many layers; more points than a row keeps
    assert L.dfepe_loss_tail_bwd(None, 5, 4, None, None, None, None, None, None, None, None, None, 0.01, None, None) == -1
    assert L.dfepe_loss_tail_bwd(None, 5, 0, None, None, None, None, None, None, None, None, None, 0.01, None, None) == 0
    assert L.dfepe_loss_head_pending(None, None) == -1
    stats = lambda r0, C: L.dfepe_loss_stats(None, r0, 1.0, None, 0, 1.0, None, 0, 1.0, None, 0, 1.0, C, None, None, None, None)
    assert stats(5, 4096) == -1 and stats(65, 4096) == -1 and stats(5, 0) == -1
    assert L.dfepe_row_dot(None, 0, None, 0, 4, 8, 100, None, None) == -1 and L.dfepe_row_dot(None, 0, None, 0, 0, 8, 100, None, None) == 0
    assert L.dfepe_row_dot(None, 0, None, 0, 4, 8, 0, None, None) == -1
    inp = lambda B_, N_, Q_, W_: L.dfepe_deepf_input(None, None, B_, N_, Q_, W_, 376.0, None, 0, 0, 1, 0, None, None, None)
    assert inp(4, 100, 0, 1241.0) == -1 and inp(0, 100, 0, 1241.0) == 0 and inp(4, 100, 0, 0.0) == -1 and inp(4, 100, -1, 1241.0) == -1
    assert L.dfepe_geo_misc(7, None, None, 4, None, None) == -1 and L.dfepe_geo_misc(6, None, None, 0, None, None) == 0
    # the estimator's normalisation for any number of points per pair: null pointers, channel counts off the 32 grid, N < 1
    assert L.dfepe_est_norm_fwd(None, 64, 64, 2, 1000, None, None, 1e-5, 0.01, None, 0, None, 0, None, 1, None, None) == -1
    assert L.dfepe_est_absmax(None, 4, None, None) == -1 and L.dfepe_est_split_f16(None, 4, 32, 32, 32, None, None, 0, None) == -1
    assert L.dfepe_est_layer_fwd(None, 0, None, 0, 64, 200, 32, None, None, None, 1e-5, 0.01, None, 0, None, 0, None, None) == -1
    assert L.dfepe_est_gemm_nt_f16(None, 0, None, 0, 64, 200, 32, None, None, 64, None) == -1
    assert L.dfepe_est_wprep(0, None, None, None, None, None, None, None, None) == -1 and L.dfepe_est_wprep(9, None, None, None, None, None, None, None, None) == -1
    assert L.dfepe_est_wprep_workspace_bytes(5) >= 5 * 4 and L.dfepe_est_wprep_workspace_bytes(0) == 0
    assert L.dfepe_est_dgamma_zero_multi(0, *([None] * 17), 0.01, 100, 4, None) == -1 and L.dfepe_est_dgamma_zero_multi(2, *([None] * 17), 0.01, 100, 4, None) == -1
    assert L.dfepe_est_saved_bytes(0, None, None, 4, 7, 100, 0) == 0 and L.dfepe_est_forward_workspace_bytes(9, None, None, 4, 7, 100, 1) == 0
    assert L.dfepe_est_forward(None, 700, 100, 4, 7, 100, 5, None, None, None, None, None, None, None, 1e-5, 0.01, None, 0, None, None, None, None) == -1
    assert L.dfepe_est_backward(None, 4, 7, 100, 5, None, None, None, None, None, None, 0.01, None, None, None, None, None, None, None, None, None, None, 700, 100, None) == -1
    assert L.dfepe_est_colsum(0, None, None, None, None, None) == -1 and L.dfepe_est_colsum(41, None, None, None, None, None) == -1
    # version 154: split-K products, the register-resident normalisations, the table-driven weight gradient, prepared parameters
    assert L.dfepe_est_gemm_nt_f16_splitk(None, 0, None, 0, 64, 200, 256, None, None, 64, 2, 0, None) == -1
    assert L.dfepe_est_gemm_nt_splitk(None, 0, None, 0, 64, 200, 256, None, 64, 2, 0, None) == -1
    assert L.dfepe_est_gemm_nt_gx(None, 0, None, 0, 32, 200, 64, None, 7, 100, 700, 100, None) == -1
    assert L.dfepe_est_gemm_tn_multi(0, None, None, None, None, None, None, 800, None, None, None) == -1
    assert L.dfepe_est_gemm_tn_multi(9, None, None, None, None, None, None, 800, None, None, None) == -1
    assert L.dfepe_est_norm_fwd_r(None, 64, 1, 0, 64, 2, 1000, None, None, 1e-5, 0.01, None, 0, None, 0, None, None) == -1
    assert L.dfepe_est_in_bwd_r(None, 64, 1, 0, None, None, None, 0, None, None, None, 0.01, 64, 2, 1000, None, 0, None, None, None) == -1
    assert L.dfepe_est_prepare(5, None, None, None, None, None) == -1 and L.dfepe_est_prep_bytes(0, None, None) == 0
    assert L.dfepe_est_head_dw(None, 0, 256, 800, 512, None, None, None, None) == -1
    assert L.dfepe_est_dgrad_in_bwd(None, 0, None, 0, 64, 200, 32, None, 0, None, None, None, 0.01, None, 0, None, None, None) == -1
    assert L.dfepe_est_in_bwd_n(None, None, None, None, 0, None, None, None, 0.01, 64, 2, 1000, None, 0, None, None, 1, None, None) == -1


def test_estimator_pass_buffer_sizes(dfepe):
    """The caller-owned buffers of dfepe_est_forward / dfepe_est_backward: sizes for the reference's estimator (7 -> 64 -> 128 -> 1024 ->
    512 -> 256) grow with the batch, do not depend on need_gx, a forward that keeps nothing needs no
    `saved` at all; widths off the 32 grid or a broken chain of widths are refused (size 0)."""
    import ctypes
    L = dfepe._lib.lib()
    arr = lambda v: (ctypes.c_int * len(v))(*v)
    Co, Ci = arr([64, 128, 1024, 512, 256]), arr([7, 64, 128, 1024, 512])
    s8, s16 = L.dfepe_est_saved_bytes(5, Co, Ci, 8, 7, 100, 0), L.dfepe_est_saved_bytes(5, Co, Ci, 16, 7, 100, 0)
    assert 0 < s8 < s16 and L.dfepe_est_saved_bytes(5, Co, Ci, 8, 7, 100, 1) == s8  # a backward may ask for gx or not: same layout
    # per column: two bf16 planes of 32 + 64 + 128 + 1024 + 512 + 256 channels
    assert s8 >= 8 * 100 * (32 + 64 + 128 + 1024 + 512 + 256) * 4
    assert L.dfepe_est_forward_workspace_bytes(5, Co, Ci, 8, 7, 100, 1) > 0 and L.dfepe_est_backward_workspace_bytes(5, Co, Ci, 8, 7, 100, 1) > 0
    assert L.dfepe_est_forward_workspace_bytes(5, Co, Ci, 8, 7, 37, 0) > L.dfepe_est_forward_workspace_bytes(5, Co, Ci, 8, 7, 37, 1) * 0  # generic N: sized too
    assert L.dfepe_est_saved_bytes(5, arr([64, 100, 1024, 512, 256]), Ci, 8, 7, 100, 0) == 0   # 100 % 32 != 0
    assert L.dfepe_est_saved_bytes(5, Co, arr([7, 64, 128, 1000, 512]), 8, 7, 100, 0) == 0     # Ci[3] != Co[2]
    assert L.dfepe_est_saved_bytes(5, Co, Ci, 8, 7, 1, 0) == 0                                   # one point per pair: the stock stack's error
    # the reference's own shape (8 pairs x 1000 points): plain products, a K-heavy layer's partials included; prepared parameters
    assert L.dfepe_est_forward_workspace_bytes(5, Co, Ci, 8, 7, 1000, 1) >= 8 * 1000 * 1024 * 4
    assert L.dfepe_est_backward_workspace_bytes(5, Co, Ci, 8, 7, 1000, 1) >= 8 * 1000 * 1024 * 4
    pb = L.dfepe_est_prep_bytes(5, Co, Ci)
    assert pb >= 2 * 2 * 2 * (64 * 32 + 128 * 64 + 1024 * 128 + 512 * 1024 + 256 * 512)       # two formats x two planes x two bytes
    assert L.dfepe_est_prep_bytes(5, arr([64, 100, 1024, 512, 256]), Ci) == 0


def test_no_cpu_fallback(dfepe):
    """The product refuses CPU tensors instead of silently computing somewhere else."""
    with pytest.raises(dfepe.DfepeError):
        dfepe.ops.w8pt(torch.zeros(2, 10, 3), torch.zeros(2, 10, 3), torch.zeros(2, 10))


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(REPO, "pytorch-deepfepe_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                hits = re.findall(r"^\s*(?:from|import)\s+[^\n]*oracle|import_module\([^)]*oracle", src, flags=re.M)
                assert not hits, f"{f} imports the oracle: the product path must not use it ({hits})"


def test_synth_scene_conventions(dfepe, oracle):
    sc = dfepe.synth.make_scene(4, 50, seed=0, noise_px=0.0, dtype=torch.float64)
    # x2^T F x1 = 0 for the generated F and virtual points; E^T decomposes to the camera motion
    x1 = torch.cat((sc["matches_xy_ori"][:, :, :2], torch.ones(4, 50, 1, dtype=torch.float64)), 2)
    x2 = torch.cat((sc["matches_xy_ori"][:, :, 2:], torch.ones(4, 50, 1, dtype=torch.float64)), 2)
    r = ((x2 @ sc["F_gt"]) * x1).sum(2)
    assert r.abs().max() < 1e-9
    rv = ((sc["pts2_virt_ori"] @ sc["F_gt"]) * sc["pts1_virt_ori"]).sum(2)
    assert rv.abs().max() < 1e-9
    pose = oracle.rt_loss([sc["E_gt"]], sc["delta_Rtijs_4_4"], sc["qs_cam"], sc["ts_cam"])
    assert pose["q_l2"].max() < 1e-7 and pose["t_l2"].max() < 1e-7
    assert pose["R_deg"].max() < 1e-4 and pose["t_deg"].max() < 1e-2


def test_compat_surface_without_a_gpu(dfepe):
    """Host-side contract of the compat layer that needs no device: constructor signatures, state_dict keys, the branches
    that are deliberately not built, argument checks that the reference also makes on the host."""
    C = dfepe.compat
    assert C.DeepFNet.Fit(normalize_SVD=False).normalize_SVD is False  # built in round 3 (un-normalised rows, DeepFNet.py:211)
    for flag in ("if_goodCorresArch", "if_tri_depth", "if_des"):
        with pytest.raises(NotImplementedError):
            C.DeepFNet.DeepFNet(depth=2, image_size=[376, 1241, 3], if_quality=False, **{flag: True})
    # extra keyword arguments of train_good.py (img_zoom_xy, if_img_des_to_pointnet, ...) are swallowed like in the reference
    net = C.DeepFNet.DeepFNet(depth=3, image_size=[376, 1241, 3], if_quality=False, img_zoom_xy=(1.0, 1.0), if_img_feat=False,
                              if_cpu_svd=True)
    keys = list(net.state_dict().keys())
    assert any(k.startswith("input_weights.fw.") for k in keys) and any(k.startswith("update_weights.fw.") for k in keys)
    with pytest.raises(NotImplementedError):
        C.train_good_utils.get_all_loss_DeepF({}, None, None, None, {"if_tri_depth": True})
    with pytest.raises(dfepe.DfepeError):  # CPU tensors: refused, not silently computed elsewhere
        net({"matches_xy_ori": torch.zeros(2, 16, 4), "matches_good_unique_nums": None, "t_scene_scale": None})
    tr = C.model_wrap.PointTracker(max_length=2, nn_thresh=0.7)
    assert tr.nn_thresh == 0.7
    with pytest.raises(ValueError):
        C.model_wrap.PointTracker(max_length=1)
    assert tr.nn_match_two_way(np.zeros((8, 0)), np.zeros((8, 5)), 0.7).shape == (3, 0)   # empty side: [3,0] like the reference
    with pytest.raises(ValueError):
        tr.nn_match_two_way(np.ones((8, 3)), np.ones((8, 5)), -0.1)
    with pytest.raises(AssertionError):
        C.utils_misc.crop_or_pad_choice(5, 0)


def test_boundary_helpers_of_the_reference_surface(dfepe, oracle, golden):
    """The small helpers of the mirrored call surface that are plain torch / numpy (no kernel): Fit.normalize against the
    reference's own output (golden), _normalize_XY(_batch) against the pinned oracle, utils_misc / utils_geo helpers against
    their definitions."""
    um, ug, uF = dfepe.compat.utils_misc, dfepe.compat.utils_geo, dfepe.compat.utils_F
    g = golden("fit")
    pts1 = torch.from_numpy(g["general_f32_pts1"])
    fit = dfepe.compat.DeepFNet.Fit(is_cuda=False)
    out, T = fit.normalize(pts1, torch.ones(pts1.shape[0], pts1.shape[1], 1))  # weighted_svd passes unit weights (DeepFNet.py:198-199)
    np.testing.assert_allclose(T.numpy(), g["general_f32_hartley1_T"], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(out.numpy(), g["general_f32_hartley1_pts"], rtol=2e-5, atol=2e-6)
    gen = torch.Generator().manual_seed(0)
    X, Y = torch.randn(40, 2, generator=gen, dtype=torch.float64) * 3 + 1, torch.randn(40, 2, generator=gen, dtype=torch.float64) - 2
    Xn, Yn, T1, T2 = uF._normalize_XY(X, Y)
    Xo, To = oracle._normalize_xy(X)
    np.testing.assert_allclose(Xn.numpy(), Xo.numpy(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(T1.numpy(), To.numpy(), rtol=1e-12, atol=1e-12)
    assert abs(float(Yn.norm(dim=1).mean()) - 2 ** 0.5) < 1e-9 and float(Yn.mean(0).abs().max()) < 1e-9
    Xb, Yb, T1b, T2b = uF._normalize_XY_batch(torch.stack((X, Y)), torch.stack((Y, X)))
    np.testing.assert_allclose(Xb[0].numpy(), Xn.numpy(), atol=1e-12)
    np.testing.assert_allclose(T2b[1].numpy(), T1.numpy(), atol=1e-12)
    with pytest.raises(ValueError):
        uF._normalize_XY(X, Y[:5])
    # homogeneous helpers, cross-product matrix, rigid-transform inversion
    assert um._homo(X).shape == (40, 3) and torch.equal(um._homo(X)[:, 2], torch.ones(40, dtype=torch.float64))
    np.testing.assert_allclose(um._de_homo(um._homo(X) * 2.0).numpy(), X.numpy(), rtol=1e-9)
    v, w = torch.randn(3, 1, generator=gen), torch.randn(3, 1, generator=gen)
    np.testing.assert_allclose((um._skew_symmetric(v) @ w).flatten().numpy(), torch.linalg.cross(v.flatten(), w.flatten()).numpy(), atol=1e-6)
    assert um._skew_symmetric(torch.stack((v, w))).shape == (2, 3, 3)
    np.testing.assert_allclose(um.skew_symmetric_np(v.numpy()), um._skew_symmetric(v).numpy())
    sc = dfepe.synth.make_scene(2, 10, seed=0, dtype=torch.float64)
    Rt = sc["delta_Rtijs_4_4"][0, :3, :]
    inv = um._inv_Rt(Rt)
    np.testing.assert_allclose(um.Rt_pad(inv.numpy()) @ um.Rt_pad(Rt.numpy()), np.eye(4), atol=1e-12)
    np.testing.assert_allclose(um.inv_Rt_np(Rt.numpy()), inv.numpy(), atol=1e-15)
    R12, t12 = ug.invert_Rt(Rt[:, :3].numpy(), Rt[:, 3:4].numpy())
    np.testing.assert_allclose(np.hstack((R12, t12)), um.Rt_depad(np.linalg.inv(um.Rt_pad(Rt.numpy()))), atol=1e-12)
    assert um.identity_Rt().shape == (3, 4) and um.homo_np(X.numpy()).shape == (40, 3)
    np.testing.assert_allclose(um.de_homo_np(um.homo_np(X.numpy())), X.numpy(), rtol=1e-9)
