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
    assert L.dfepe_version() == 120
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
    assert L.dfepe_w8pt_bwd(None, None, None, 2, 0, 1, 0, 0.0, 0.0, 0.5, None, None, None, None, None, None, None, None, None, None, None) == -1
    assert L.dfepe_w8pt_fwd(None, None, None, 4, 100, 1, 1 << 9, 0.0, 0.0, 0.5, None, None, None, None, None, None) == -1  # unknown flag bit
    assert L.dfepe_floss_fwd(None, 0, 4, None, None, 0, None, None, None, 100, 0.02, None, None, None) == -1
    assert L.dfepe_floss_fwd(None, 5, 0, None, None, 0, None, None, None, 100, 0.02, None, None, None) == 0
    assert L.dfepe_floss_fwd(None, 5, 4, None, None, 3, None, None, None, 100, 0.02, None, None, None) == -1  # bad stride
    assert L.dfepe_pose_fwd(None, 5, 4, None, None, None, None, None, None, None, None, None) == -1
    assert L.dfepe_pose_bwd(None, 5, 0, None, None, None, None, 0.0, 0.0, 0.0, 0.0, None, None, None) == 0
    assert L.dfepe_loss_head(None, None, None, 5, 4, 100, 0.1, 0.5, 1.0, 0.1, None, None, None) == -1
    tail = lambda L_, B_, M_: L.dfepe_loss_tail(None, L_, B_, None, None, 0, None, None, None, M_, 0.02, None, None, None, 0.1, 0.5, 1.0, 1.0, 0.1, 4.0,
                                               None, None, None, None, None, None, None, None, None, None, None, None)
    assert tail(5, 4, 100) == -1 and tail(0, 4, 100) == -1 and tail(5, 4, 200) == -3  # null pointers; no layers; grid too large for the fused kernel
    assert L.dfepe_loss_tail_workspace_bytes(4096) >= 64 + 256 * 48 * 8


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
    with pytest.raises(NotImplementedError):
        C.DeepFNet.Fit(normalize_SVD=False)
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
