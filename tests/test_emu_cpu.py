"""The row-per-pair HIP kernel bodies (csrc/w8pt16_body.h, csrc/w8pt16_bwd_body.h), compiled for the HOST against the
row-group emulation of tests/emu/ and compared with the oracle.  Runs without a GPU: it checks the arithmetic of the
kernels' own source (every phase, the save record, the analytic adjoint), not the DPP encodings -- those are covered on
the GPU box by tests/test_rowgroup_gpu.py.  The emulation library is test infrastructure; the product never loads it."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
EMU_DIR = os.path.join(REPO, "tests", "emu")
IMAGE_SIZE = [376, 1241, 3]
RAW, LOGITS = 1, 2


@pytest.fixture(scope="session")
def emu():
    out = os.path.join(EMU_DIR, "_build")
    os.makedirs(out, exist_ok=True)
    lib = os.path.join(out, "libemu_w8pt16.so")
    srcs = [os.path.join(EMU_DIR, "emu_w8pt16.cpp"), os.path.join(EMU_DIR, "rowgroup.h")] + [
        os.path.join(REPO, "pytorch-deepfepe_amd", "csrc", f) for f in ("w8pt16_body.h", "w8pt16_bwd_body.h", "dfepe_math.h")]
    if not os.path.exists(lib) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in srcs):
        subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", f"-I{EMU_DIR}", f"-I{REPO}/pytorch-deepfepe_amd/csrc",
                        f"-I{REPO}/include", srcs[0], "-o", lib], check=True)
    L = ctypes.CDLL(lib)
    P, I, U, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_float
    L.emu_w8pt16_fwd.restype = I
    L.emu_w8pt16_fwd.argtypes = [P, P, P, I, I, I, U, F, F, F, P, P, P, P, P]
    L.emu_set_lean.restype = None
    L.emu_set_lean.argtypes = [I]
    L.emu_w8pt16_bwd.restype = I
    L.emu_w8pt16_bwd.argtypes = [P, P, P, I, I, I, U, F, F, F, P, P, P, P, P, P, P, P, P, P]
    return L


def _p(t):
    return None if t is None else t.data_ptr()


def emu_fwd(L, pts1, pts2, w, flags, want_epi=True, want_save=True, clamp_at=0.5):
    B, N = w.shape
    F = torch.empty(B, 3, 3)
    res = torch.empty(B, N)
    epi = torch.empty(B, N) if want_epi else None
    save = torch.zeros(B, 128) if want_save else None
    wout = torch.empty(B, N) if flags & LOGITS else None
    rc = L.emu_w8pt16_fwd(_p(pts1), _p(pts2), _p(w), B, N, 1, flags, float(IMAGE_SIZE[1]), float(IMAGE_SIZE[0]), clamp_at,
                          _p(F), _p(res), _p(epi), _p(save), _p(wout))
    assert rc == 0
    return F, res, epi, save, wout


def emu_bwd(L, pts1, pts2, w, flags, save, F, gF, gRes, gEpi, want_pts=False, clamp_at=0.5, g_scale=None):
    B, N = w.shape
    gW = torch.empty(B, N)
    gP1 = torch.empty_like(pts1) if want_pts else None
    gP2 = torch.empty_like(pts2) if (want_pts and pts2 is not None) else None
    rc = L.emu_w8pt16_bwd(_p(pts1), _p(pts2), _p(w), B, N, 1, flags, float(IMAGE_SIZE[1]), float(IMAGE_SIZE[0]), clamp_at,
                          _p(save), _p(F), _p(gF), _p(gRes), _p(gEpi), None, _p(g_scale), _p(gW), _p(gP1), _p(gP2))
    assert rc == 0
    return gW, gP1, gP2


def unit_align(a, ref):
    a = a.reshape(a.shape[0], -1).double()
    r = ref.reshape(ref.shape[0], -1).double()
    a = a / a.norm(dim=1, keepdim=True)
    r = r / r.norm(dim=1, keepdim=True)
    s = torch.sign((a * r).sum(1, keepdim=True))
    s[s == 0] = 1
    return a * s, r, s[:, 0]


@pytest.mark.parametrize("B,N,outl,noise", [(24, 100, 0.2, 0.5), (8, 100, 0.4, 0.5), (8, 100, 0.0, 0.0), (6, 128, 0.2, 0.5),
                                            (6, 113, 0.2, 0.5), (6, 64, 0.2, 0.5), (6, 17, 0.2, 0.5), (6, 16, 0.2, 0.5),
                                            (6, 12, 0.2, 0.5), (6, 9, 0.2, 0.5), (6, 8, 0.2, 0.5), (4, 5, 0.2, 0.5),
                                            (4, 129, 0.2, 0.5), (3, 1000, 0.2, 0.5), (3, 257, 0.4, 0.5),
                                            (5, 33, 0.2, 0.5), (5, 48, 0.2, 0.5), (5, 80, 0.2, 0.5), (5, 81, 0.2, 0.5), (5, 96, 0.2, 0.5)])
def test_forward_body_matches_fp64_oracle(emu, dfepe, oracle, B, N, outl, noise):
    sc = dfepe.synth.make_scene(B, N, seed=7 * N + B, outlier_ratio=outl, noise_px=noise)
    m = sc["matches_xy_ori"].contiguous()
    logits = sc["logits_layers"][0].contiguous()
    w = torch.softmax(logits, dim=1)
    F, res, epi, save, wout = emu_fwd(emu, m, None, logits, RAW | LOGITS)
    np.testing.assert_allclose(wout.numpy(), w.numpy(), rtol=2e-6, atol=1e-9)
    p1, p2, _ = oracle.normalize_hw(m.double(), IMAGE_SIZE)
    o_out, o_res, _ = oracle.fit_forward(p1, p2, wout.double().unsqueeze(1))
    a, r, s = unit_align(F, o_out)
    tol = 1e-5 if N < 12 else 1e-6  # near-minimal systems are ill-conditioned
    assert (a - r).norm(dim=1).max().item() < 2 * tol  # the fp32 image-size normalisation x^ = 2x/W - 1 differs by 1 ulp
    np.testing.assert_allclose((res * s[:, None]).numpy(), o_res.numpy(), atol=5e-7, rtol=1e-4)
    o_epi = oracle.compute_epi_residual(p1, p2, o_out, 0.5).numpy()
    np.testing.assert_allclose(epi.numpy(), o_epi, atol=5e-5, rtol=1e-3)
    # same function through the homogeneous-points entry with plain weights
    p1f, p2f, _ = oracle.normalize_hw(m, IMAGE_SIZE)
    F2, res2, _, _, _ = emu_fwd(emu, p1f.contiguous(), p2f.contiguous(), wout, 0)
    o2, _, _ = oracle.fit_forward(p1f.double(), p2f.double(), wout.double().unsqueeze(1))  # same fp32-rounded points
    a2, r2, _ = unit_align(F2, o2)
    assert (a2 - r2).norm(dim=1).max().item() < tol


@pytest.mark.parametrize("N", [100, 128, 113, 65, 112])
@pytest.mark.parametrize("flags", [RAW | LOGITS, RAW, 0])
def test_lean_forward_body_is_bit_identical(emu, dfepe, oracle, N, flags):
    """The <= 256-register build of the forward fit (w8pt16_body.h: LEAN -- the lane's correspondences and their 1 / |p| are fetched /
    derived again for the output phase instead of being held across the eigen solve; the library takes it from 8192 pairs on) against
    the resident build: every output and the whole `save` record bit for bit, with logits or weights, pixel matches or homogeneous
    points, dropped (non-finite) correspondences and a ragged last group."""
    B = 6
    sc = dfepe.synth.make_scene(B, N, seed=50 + N, outlier_ratio=0.2, noise_px=0.5)
    m = sc["matches_xy_ori"].clone().contiguous()
    m[1, 3, 2] = float("nan")   # dropped correspondences: they keep their softmax weight in weights_out, zero in X
    m[2, N - 1, 0] = float("inf")
    w = sc["logits_layers"][0].contiguous() if flags & LOGITS else torch.softmax(sc["logits_layers"][0], 1).contiguous()
    if flags & RAW:
        a, b = m, None
    else:
        p1, p2, _ = oracle.normalize_hw(torch.nan_to_num(m, nan=1e30, posinf=1e30), IMAGE_SIZE)
        a, b = p1.float().contiguous(), p2.float().contiguous()
    emu.emu_set_lean(0)
    ref = emu_fwd(emu, a, b, w, flags)
    emu.emu_set_lean(1)
    try:
        lean = emu_fwd(emu, a, b, w, flags)
    finally:
        emu.emu_set_lean(0)
    for x, y in zip(ref, lean):
        if x is not None:
            assert torch.equal(torch.nan_to_num(x), torch.nan_to_num(y)) and torch.equal(torch.isnan(x), torch.isnan(y))


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-300)


@pytest.mark.parametrize("N,outl", [(100, 0.0), (100, 0.4), (128, 0.2), (20, 0.2), (9, 0.0), (300, 0.2), (1000, 0.2), (40, 0.2), (75, 0.2), (96, 0.2)])
@pytest.mark.parametrize("use_res,use_epi", [(False, False), (True, False), (True, True)])
def test_backward_body_vs_oracle_autograd(emu, dfepe, oracle, N, outl, use_res, use_epi):
    """d/d(logits) of <F, GF> + <residual, GR> + <epi, GE> through the emulated w8pt16 forward + backward bodies against
    fp64 autograd of the oracle (same sign gauge).  Tolerance: the save record carries f, z, the reflectors and the 3x3
    SVD in fp32 (relative 1e-7, no gap amplification); T and lambda travel in fp64."""
    B = 5
    sc = dfepe.synth.make_scene(B, N, seed=7 + N, outlier_ratio=outl)
    g = torch.Generator().manual_seed(1)
    m = sc["matches_xy_ori"].contiguous()
    logits = sc["logits_layers"][0].contiguous()
    GF = torch.randn(B, 3, 3, generator=g)
    GR = torch.randn(B, N, generator=g) if use_res else None
    GE = torch.randn(B, N, generator=g) if use_epi else None
    F, res, epi, save, wout = emu_fwd(emu, m, None, logits, RAW | LOGITS)
    assert (save[:, 127] == 17.0).all()  # S16_TAG_VALUE: the round-3 record layout
    gL, _, _ = emu_bwd(emu, m, None, wout, RAW | LOGITS, save, F, GF, GR, GE)
    lo_in = logits.double().requires_grad_(True)
    wo = torch.softmax(lo_in, 1)
    p1, p2, _ = oracle.normalize_hw(m.double(), IMAGE_SIZE)
    o_out, o_res, _ = oracle.fit_forward(p1, p2, wo.unsqueeze(1))
    s = torch.sign((o_out.detach() * F.double()).flatten(1).sum(1))
    lo = (s[:, None, None] * o_out * GF.double()).sum()
    if use_res:
        lo = lo + (s[:, None] * o_res * GR.double()).sum()
    if use_epi:
        lo = lo + (oracle.compute_epi_residual(p1, p2, o_out, 0.5) * GE.double()).sum()
    lo.backward()
    assert relerr(gL.numpy(), lo_in.grad.numpy()) < (2e-4 if N >= 20 else 2e-3)


@pytest.mark.parametrize("N", [60, 200])
@pytest.mark.parametrize("raw", [True, False])
def test_point_gradients_body_vs_oracle_autograd(emu, dfepe, oracle, raw, N):
    B = 4
    sc = dfepe.synth.make_scene(B, N, seed=21, outlier_ratio=0.2)
    g = torch.Generator().manual_seed(3)
    m = sc["matches_xy_ori"].contiguous()
    w = torch.softmax(sc["logits_layers"][0], 1).contiguous()
    GF = torch.randn(B, 3, 3, generator=g)
    GR = torch.randn(B, N, generator=g)
    GE = torch.randn(B, N, generator=g)
    if raw:
        F, res, epi, save, _ = emu_fwd(emu, m, None, w, RAW)
        gW, gP1, _ = emu_bwd(emu, m, None, w, RAW, save, F, GF, GR, GE, want_pts=True)
        mo = m.double().requires_grad_(True)
        p1, p2, _ = oracle.normalize_hw(mo, IMAGE_SIZE)
    else:
        p1f, p2f, _ = oracle.normalize_hw(m, IMAGE_SIZE)
        p1f, p2f = p1f.contiguous(), p2f.contiguous()
        F, res, epi, save, _ = emu_fwd(emu, p1f, p2f, w, 0)
        gW, gP1, gP2 = emu_bwd(emu, p1f, p2f, w, 0, save, F, GF, GR, GE, want_pts=True)
        p1 = p1f.double().requires_grad_(True)
        p2 = p2f.double().requires_grad_(True)
    wo = w.double().requires_grad_(True)
    o_out, o_res, _ = oracle.fit_forward(p1, p2, wo.unsqueeze(1))
    s = torch.sign((o_out.detach() * F.double()).flatten(1).sum(1))
    lo = (s[:, None, None] * o_out * GF.double()).sum() + (s[:, None] * o_res * GR.double()).sum()
    lo = lo + (oracle.compute_epi_residual(p1, p2, o_out, 0.5) * GE.double()).sum()
    lo.backward()
    assert relerr(gW.numpy(), wo.grad.numpy()) < 2e-4
    if raw:
        assert relerr(gP1.numpy(), mo.grad.numpy()) < 5e-4
    else:
        assert relerr(gP1[:, :, :2].numpy(), p1.grad[:, :, :2].numpy()) < 5e-4
        assert relerr(gP2[:, :, :2].numpy(), p2.grad[:, :, :2].numpy()) < 5e-4


@pytest.mark.parametrize("L,M,qt,noise", [(5, 100, True, 0.003), (3, 100, False, 0.003), (2, 17, True, 0.02), (4, 128, True, 0.0005)])
def test_loss_tail_body_vs_oracle(emu, dfepe, oracle, L, M, qt, noise):
    """The fused loss tail (csrc/loss_tail_body.h) under the row emulation: per-pair F-loss sums, E, pose errors and
    d loss / d F of every layer against the oracle's f_loss / rt_loss and fp64 autograd."""
    I, P, F32 = ctypes.c_int, ctypes.c_void_p, ctypes.c_float
    emu.emu_loss_tail.restype = I
    emu.emu_loss_tail.argtypes = [P, I, I, P, P, I, P, P, P, I, F32, P, P, P, F32, F32, F32, F32, F32] + [P] * 9
    B = 6
    sc = dfepe.synth.make_scene(B, 50, seed=2 + L, n_virtual=M) if "n_virtual" in dfepe.synth.make_scene.__code__.co_varnames else dfepe.synth.make_scene(B, 50, seed=2 + L)
    v1, v2 = sc["pts1_virt_ori"][:, :M].contiguous(), sc["pts2_virt_ori"][:, :M].contiguous()
    if v1.shape[1] < M:  # tile the virtual points up to M
        rep = (M + v1.shape[1] - 1) // v1.shape[1]
        v1, v2 = v1.repeat(1, rep, 1)[:, :M].contiguous(), v2.repeat(1, rep, 1)[:, :M].contiguous()
    g = torch.Generator().manual_seed(4)
    T = oracle.hw_matrix(IMAGE_SIZE, torch.float64)
    Tinv = torch.linalg.inv(T)
    Fn = Tinv.T @ sc["F_gt"].double() @ Tinv
    Fn = Fn / Fn.flatten(1).norm(dim=1)[:, None, None]
    Fl = torch.stack([Fn + noise * (l + 1) * torch.randn(B, 3, 3, generator=g, dtype=torch.float64) for l in range(L)])
    Fl = Fl.float().contiguous()  # both sides see the same fp32-representable F
    clamp, cq, ct, bF, bq, bt = 0.02, 0.1, 0.5, 0.7, 1.0, 0.1
    coefF, coefq, coeft = bF / (L * B * M), bq / (L * B), bt / (L * B)
    Ks = sc["Ks"].contiguous()
    q_gt, t_gt = sc["qs_cam"].reshape(B, 4).contiguous(), sc["ts_cam"].reshape(B, 3).contiguous()
    R_gt = sc["delta_Rtijs_4_4"][:, :3, :3].transpose(1, 2).contiguous()
    Tf = T.float().contiguous()
    loss_sum, E = torch.zeros(L, B), torch.zeros(L, B, 3, 3)
    q_l2, t_l2, R_deg, t_deg = (torch.zeros(L, B) for _ in range(4))
    sel = torch.zeros(L, B, dtype=torch.int32)
    gF = torch.zeros(L, B, 3, 3)
    part = torch.zeros(B, 48, dtype=torch.float64)
    rc = emu.emu_loss_tail(_p(Fl), L, B, _p(Tf), _p(Tf), 0, _p(Ks), _p(v1), _p(v2), M, clamp, _p(q_gt) if qt else None, _p(t_gt), _p(R_gt),
                           cq, ct, coefF, coefq, coeft, _p(loss_sum), _p(E), _p(q_l2), _p(t_l2), _p(R_deg), _p(t_deg), _p(sel), _p(gF), _p(part))
    assert rc == 0
    Fo = Fl.double().clone().requires_grad_(True)
    outs = {"T1": T.expand(B, 3, 3), "T2": T.expand(B, 3, 3), "out_layers": [Fo[l] for l in range(L)], "F_est": Fo[-1],
            "epi_res_layers": [], "weights_layers": []}
    losses, _, _, E_layers = oracle.f_loss(outs, v1.double(), v2.double(), Ks.double(), L, clamp)
    per_pair_sum = losses["loss_per_pair"] * M
    assert relerr(loss_sum.numpy(), per_pair_sum.detach().numpy()) < 2e-5
    assert relerr(E.numpy(), torch.stack(E_layers).detach().numpy()) < 2e-6
    np.testing.assert_allclose(part[:, :L].numpy().T, loss_sum.numpy(), rtol=0, atol=0)
    lo = bF * losses["loss_F"]
    if qt:
        pose = oracle.rt_loss(E_layers, sc["delta_Rtijs_4_4"].double(), sc["qs_cam"].double(), sc["ts_cam"].double())
        np.testing.assert_allclose(q_l2.numpy(), pose["q_l2"].detach().numpy(), atol=2e-6, rtol=1e-5)
        np.testing.assert_allclose(t_l2.numpy(), pose["t_l2"].detach().numpy(), atol=2e-6, rtol=1e-5)
        np.testing.assert_allclose(R_deg.numpy(), pose["R_deg"], atol=2e-3, rtol=1e-4)
        np.testing.assert_allclose(t_deg.numpy(), pose["t_deg"], atol=2e-2, rtol=1e-4)
        np.testing.assert_allclose(part[:, 16:16 + L].numpy().T, np.clip(q_l2.numpy(), 0, cq), rtol=0, atol=0)
        np.testing.assert_allclose(part[:, 32:32 + L].numpy().T, np.clip(t_l2.numpy(), 0, ct), rtol=0, atol=0)
        lo = lo + oracle.qt_training_loss(pose["q_l2"], pose["t_l2"], cq, ct, bq, bt)
    lo.backward()
    assert relerr(gF.numpy(), Fo.grad.numpy()) < 3e-4


def test_loss_tail_body_degenerate_matrices_stay_finite(emu, dfepe, oracle):
    """Rank-1, zero, and badly scaled F (whatever a diverged estimator may produce) through the fused loss tail: the
    closed-form 3x3 SVD (svd3_closed, with its rank-1 branch of the null vector) and the pose adjoint return finite numbers, and
    where the oracle's pose is well defined (the scaled essential matrices) they still agree with it."""
    I, P, F32 = ctypes.c_int, ctypes.c_void_p, ctypes.c_float
    emu.emu_loss_tail.restype = I
    emu.emu_loss_tail.argtypes = [P, I, I, P, P, I, P, P, P, I, F32, P, P, P, F32, F32, F32, F32, F32] + [P] * 9
    B, L, M = 6, 2, 32
    sc = dfepe.synth.make_scene(B, 50, seed=11)
    v1, v2 = sc["pts1_virt_ori"][:, :M].contiguous(), sc["pts2_virt_ori"][:, :M].contiguous()
    T = oracle.hw_matrix(IMAGE_SIZE, torch.float64)
    Tinv = torch.linalg.inv(T)
    Fn = Tinv.T @ sc["F_gt"].double() @ Tinv
    Fn = Fn / Fn.flatten(1).norm(dim=1)[:, None, None]
    g = torch.Generator().manual_seed(3)
    a, b = torch.randn(3, generator=g, dtype=torch.float64), torch.randn(3, generator=g, dtype=torch.float64)
    Fl = torch.stack([Fn.clone(), Fn.clone()])
    Fl[:, 0] = torch.outer(a, b)          # rank 1
    Fl[:, 1] = 0.0                        # zero
    Fl[:, 2] = Fn[2] * 1e-6               # tiny
    Fl[:, 3] = Fn[3] * 1e6                # huge
    Fl[:, 4] = torch.outer(a, a) * 1e-3   # rank 1, symmetric
    Fl = Fl.float().contiguous()
    Ks = sc["Ks"].contiguous()
    q_gt, t_gt = sc["qs_cam"].reshape(B, 4).contiguous(), sc["ts_cam"].reshape(B, 3).contiguous()
    R_gt = sc["delta_Rtijs_4_4"][:, :3, :3].transpose(1, 2).contiguous()
    Tf = T.float().contiguous()
    loss_sum, E = torch.zeros(L, B), torch.zeros(L, B, 3, 3)
    q_l2, t_l2, R_deg, t_deg = (torch.zeros(L, B) for _ in range(4))
    sel = torch.zeros(L, B, dtype=torch.int32)
    gF = torch.zeros(L, B, 3, 3)
    part = torch.zeros(B, 48, dtype=torch.float64)
    rc = emu.emu_loss_tail(_p(Fl), L, B, _p(Tf), _p(Tf), 0, _p(Ks), _p(v1), _p(v2), M, 0.02, _p(q_gt), _p(t_gt), _p(R_gt),
                           0.1, 0.5, 1.0 / (L * B * M), 1.0 / (L * B), 0.1 / (L * B), _p(loss_sum), _p(E), _p(q_l2), _p(t_l2), _p(R_deg),
                           _p(t_deg), _p(sel), _p(gF), _p(part))
    assert rc == 0
    for t in (loss_sum, E, q_l2, t_l2, R_deg, t_deg, gF, part):
        assert torch.isfinite(t).all()
    # the pose of a (tiny / huge) essential matrix does not depend on its scale
    E_layers = [Tf.double().T @ Fl[l].double() @ Tf.double() for l in range(L)]
    E_layers = [Ks.double().transpose(1, 2) @ e @ Ks.double() for e in E_layers]
    pose = oracle.rt_loss(E_layers, sc["delta_Rtijs_4_4"].double(), sc["qs_cam"].double(), sc["ts_cam"].double())
    for bidx in (2, 3, 5):
        np.testing.assert_allclose(q_l2[:, bidx].numpy(), pose["q_l2"][:, bidx].numpy(), atol=5e-6)
        np.testing.assert_allclose(t_l2[:, bidx].numpy(), pose["t_l2"][:, bidx].numpy(), atol=5e-6)
