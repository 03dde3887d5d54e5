"""Parity of the analytic adjoints (w8pt_bwd, floss_bwd, pose_bwd) and of the whole hot-path step against
torch.autograd of the CPU oracle run in fp64.  GPU box only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
IMAGE_SIZE = [376, 1241, 3]
DEV = "cuda:0"


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-300)


@pytest.mark.parametrize("N,outl", [(100, 0.0), (100, 0.4), (257, 0.2)])
@pytest.mark.parametrize("use_epi", [False, True])
def test_w8pt_backward_vs_oracle_autograd(dfepe, oracle, N, outl, use_epi):
    B = 5
    sc = dfepe.synth.make_scene(B, N, seed=7 + N, outlier_ratio=outl)
    g = torch.Generator().manual_seed(1)
    m = sc["matches_xy_ori"]
    logits = sc["logits_layers"][0]
    GF = torch.randn(B, 3, 3, generator=g)
    GR = torch.randn(B, N, generator=g)
    GE = torch.randn(B, N, generator=g)
    # ours
    w = torch.softmax(logits, 1).to(DEV).requires_grad_(True)
    outs = dfepe.ops.w8pt_raw(m.to(DEV), w, IMAGE_SIZE[1], IMAGE_SIZE[0], clamp_at=0.5, want_epi=True)
    F, res, epi = outs
    loss = (F * GF.to(DEV)).sum() + (res * GR.to(DEV)).sum()
    if use_epi:
        loss = loss + (epi * GE.to(DEV)).sum()
    loss.backward()
    # oracle (fp64 autograd), with the per-pair sign gauge of our forward applied to the sign-odd outputs
    wo = torch.softmax(logits.double(), 1).requires_grad_(True)
    p1, p2, _ = oracle.normalize_hw(m.double(), IMAGE_SIZE)
    o_out, o_res, _ = oracle.fit_forward(p1, p2, wo.unsqueeze(1))
    s = torch.sign((o_out.detach() * F.detach().cpu().double()).flatten(1).sum(1))
    lo = (s[:, None, None] * o_out * GF.double()).sum() + (s[:, None] * o_res * GR.double()).sum()
    if use_epi:
        lo = lo + (oracle.compute_epi_residual(p1, p2, o_out, 0.5) * GE.double()).sum()
    lo.backward()
    # tolerance: the eigenpairs other than the selected one come from the fp32 Jacobi (error ~ eps32 |M| / gap);
    # the reference's own fp32 autograd is ~50x further from the fp64 truth (see tests/test_oracle_golden.py, 5e-2)
    assert relerr(w.grad.cpu().numpy(), wo.grad.numpy()) < 2e-3


def test_floss_forward_backward(dfepe, oracle):
    L, B, M = 5, 7, 100
    sc = dfepe.synth.make_scene(B, 50, seed=2)
    g = torch.Generator().manual_seed(4)
    # F layers: ground-truth F in HW-normalised coordinates, perturbed differently per layer
    T = oracle.hw_matrix(IMAGE_SIZE, torch.float64)
    Tinv = torch.linalg.inv(T)
    Fn = Tinv.T @ sc["F_gt"].double() @ Tinv
    Fn = Fn / Fn.flatten(1).norm(dim=1)[:, None, None]
    Fl = torch.stack([Fn + 0.003 * (l + 1) * torch.randn(B, 3, 3, generator=g, dtype=torch.float64) for l in range(L)])
    Fl = Fl.float().double()  # both sides see the same fp32-representable F (clamp masks and sign(dd) are discontinuous)
    GL = torch.rand(L, B, generator=g, dtype=torch.float64)
    GEm = torch.randn(L, B, 3, 3, generator=g, dtype=torch.float64)
    clamp = 0.02
    # oracle
    Fo = Fl.clone().requires_grad_(True)
    outs = {"T1": T.expand(B, 3, 3), "T2": T.expand(B, 3, 3), "out_layers": [Fo[l] for l in range(L)], "F_est": Fo[-1],
            "epi_res_layers": [], "weights_layers": []}
    losses, _, _, E_layers = oracle.f_loss(outs, sc["pts1_virt_ori"].double(), sc["pts2_virt_ori"].double(), sc["Ks"].double(), L, clamp)
    per_pair_sum = losses["loss_per_pair"] * M
    lo = (per_pair_sum * GL).sum() + (torch.stack(E_layers) * GEm).sum()
    lo.backward()
    # ours
    Fd = Fl.float().to(DEV).requires_grad_(True)
    Td = T.float().to(DEV)
    loss_sum, E = dfepe.ops.floss(Fd, Td, Td, sc["Ks"].to(DEV), sc["pts1_virt_ori"].to(DEV), sc["pts2_virt_ori"].to(DEV), clamp)
    ((loss_sum * GL.float().to(DEV)).sum() + (E * GEm.float().to(DEV)).sum()).backward()
    assert relerr(loss_sum.detach().cpu().numpy(), per_pair_sum.detach().numpy()) < 2e-5
    assert relerr(E.detach().cpu().numpy(), torch.stack(E_layers).detach().numpy()) < 2e-6
    assert relerr(Fd.grad.cpu().numpy(), Fo.grad.numpy()) < 2e-4
    # per-pair transforms [B,3,3] give the same answer as the shared one
    loss_sum2, E2 = dfepe.ops.floss(Fd.detach(), Td.expand(B, 3, 3).contiguous(), Td.expand(B, 3, 3).contiguous(), sc["Ks"].to(DEV),
                                    sc["pts1_virt_ori"].to(DEV), sc["pts2_virt_ori"].to(DEV), clamp)
    assert torch.equal(loss_sum2, loss_sum.detach()) and torch.equal(E2, E.detach())


@pytest.mark.parametrize("noise", [0.05, 1e-3, 0.0])
def test_pose_forward_backward(dfepe, oracle, noise):
    L, B = 3, 16
    sc = dfepe.synth.make_scene(B, 20, seed=9, dtype=torch.float64)
    g = torch.Generator().manual_seed(2)
    E = torch.stack([sc["E_gt"] / sc["E_gt"].flatten(1).norm(dim=1)[:, None, None] + noise * (l + 1) * torch.randn(B, 3, 3, generator=g, dtype=torch.float64) for l in range(L)])
    E = E.float().double()  # the kernel sees fp32 inputs
    # ground truth of a *different* pair (roll) so that the errors are away from their kink at 0
    q_gt = torch.roll(sc["qs_cam"], 1, 0)
    t_gt = torch.roll(sc["ts_cam"], 1, 0)
    delta = torch.roll(sc["delta_Rtijs_4_4"], 1, 0)
    GQ = torch.rand(L, B, generator=g, dtype=torch.float64)
    GT = torch.rand(L, B, generator=g, dtype=torch.float64)
    Eo = E.clone().requires_grad_(True)
    pose = oracle.rt_loss([Eo[l] for l in range(L)], delta, q_gt, t_gt)
    R_gt = torch.linalg.inv(delta)[:, :3, :3]
    Ed = E.float().to(DEV).requires_grad_(True)
    q_l2, t_l2, R_deg, t_deg, sel = dfepe.ops.pose_errors(Ed, q_gt.float().to(DEV), t_gt.float().to(DEV), R_gt.float().to(DEV))
    np.testing.assert_allclose(q_l2.detach().cpu().numpy(), pose["q_l2"].detach().numpy(), atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(t_l2.detach().cpu().numpy(), pose["t_l2"].detach().numpy(), atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(R_deg.cpu().numpy(), pose["R_deg"], atol=2e-3, rtol=1e-4)
    np.testing.assert_allclose(t_deg.cpu().numpy(), pose["t_deg"], atol=2e-2, rtol=1e-4)
    ((q_l2 * GQ.float().to(DEV)).sum() + (t_l2 * GT.float().to(DEV)).sum()).backward()
    assert torch.isfinite(Ed.grad).all()
    if noise > 0:  # at exact essential matrices torch's svd_backward divides by s1^2 - s2^2 = 0: no autograd yard-stick there
        ((pose["q_l2"] * GQ).sum() + (pose["t_l2"] * GT).sum()).backward()
        assert relerr(Ed.grad.cpu().numpy(), Eo.grad.numpy()) < (2e-4 if noise > 1e-2 else 2e-2)
    else:  # finite differences of the kernel's own forward in the direction of a random perturbation
        D = torch.randn(L, B, 3, 3, generator=g, dtype=torch.float64)
        eps = 1e-3
        def f(Ex):
            q, t, *_ = dfepe.ops.pose_errors(Ex.float().to(DEV), q_gt.float().to(DEV), t_gt.float().to(DEV), R_gt.float().to(DEV))
            return (q.double().cpu() * GQ).sum(0) + (t.double().cpu() * GT).sum(0)
        num = (f(E + eps * D) - f(E - eps * D)) / (2 * eps)
        ana = (Ed.grad.double().cpu() * D).sum(dim=(0, 2, 3))
        assert relerr(ana.numpy(), num.numpy()) < 2e-2


@pytest.mark.parametrize("depth,qt", [(5, True), (5, False), (1, True)])
def test_hot_path_step_matches_oracle(dfepe, oracle, depth, qt):
    B, N = 6, 100
    sc = dfepe.synth.make_scene(B, N, seed=21, outlier_ratio=0.2, depth_layers=depth)
    ours = dfepe.pipeline.hot_path_step(dfepe.pipeline.scene_to_device(sc, DEV), IMAGE_SIZE, depth, 0.02, qt=qt)
    sc64 = {k: v.double() for k, v in sc.items()}
    ref = oracle.hot_path_step(sc64, IMAGE_SIZE, depth, 0.02, qt=qt, mode="batched")
    assert abs(ours["loss"].item() - ref["loss"].item()) < 2e-6 * max(1.0, abs(ref["loss"].item()))
    np.testing.assert_allclose(ours["loss_layers"].detach().cpu().numpy(), torch.stack(ref["losses"]["loss_layers"]).detach().numpy(), rtol=2e-5, atol=1e-8)
    if qt:
        np.testing.assert_allclose(ours["q_l2"].detach().cpu().numpy(), ref["pose"]["q_l2"].detach().numpy(), atol=2e-6, rtol=1e-5)
        np.testing.assert_allclose(ours["t_l2"].detach().cpu().numpy(), ref["pose"]["t_l2"].detach().numpy(), atol=2e-5, rtol=1e-4)
        np.testing.assert_allclose(ours["R_deg"].cpu().numpy(), ref["pose"]["R_deg"], atol=2e-3, rtol=1e-4)
        np.testing.assert_allclose(ours["t_deg"].cpu().numpy(), ref["pose"]["t_deg"], atol=2e-2, rtol=1e-4)
    assert relerr(ours["grad_logits"].cpu().numpy(), ref["grad_logits"].numpy()) < 5e-4


def test_hot_path_step_matches_reference_golden(dfepe, oracle, golden):
    """Same step against the reference's own fp32 run (tests/golden/pipeline.npz).

    The reference's fp32 autograd through B x depth torch.svd nodes is the noisy side of the gradient comparison: the test
    measures, on the same inputs, reference-fp32 vs fp64-oracle (4e-5 of the largest entry) and ours vs fp64-oracle (1e-6),
    prints both, and bounds ours-vs-reference by 5e-4."""
    g = golden("pipeline")
    pre = "solver_"
    sc = {k: torch.from_numpy(g[pre + k]) for k in ("matches_xy_ori", "Ks", "delta_Rtijs_4_4", "qs_cam", "ts_cam", "pts1_virt_ori", "pts2_virt_ori", "logits_layers")}
    dev = dfepe.pipeline.scene_to_device(sc, DEV)
    ours = dfepe.pipeline.hot_path_step(dev, IMAGE_SIZE, 5, 0.02, qt=True, backward=False)
    np.testing.assert_allclose(ours["loss_layers"].cpu().numpy(), g[pre + "loss_layers"], rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(ours["q_l2"].cpu().numpy(), g[pre + "q_l2_layers"], atol=2e-4, rtol=2e-3)
    np.testing.assert_allclose(ours["t_l2"].cpu().numpy(), g[pre + "t_l2_layers"], atol=3e-3, rtol=3e-3)
    np.testing.assert_allclose(ours["t_deg"].cpu().numpy(), g[pre + "t_angle_layers"], atol=0.2, rtol=3e-3)
    for l in range(5):
        np.testing.assert_allclose(ours["epi_res_layers"][l].cpu().numpy()[:, None, :] if l < 4 else 0, g[pre + "epi_res_layers"][l] if l < 4 else 0, atol=5e-4, rtol=5e-3)
    # gradients of the two reference training losses w.r.t. the logits
    for key, qt_only in (("grad_logits_lossF", False), ("grad_logits_lossQT", True)):
        logits = dev["logits_layers"].clone().requires_grad_(True)
        out = dfepe.pipeline.hot_path_forward(dev["matches_xy_ori"], logits, dev["Ks"], dev["pts1_virt_ori"], dev["pts2_virt_ori"],
                                              dev["qs_cam"], dev["ts_cam"], dev["R_gt"], IMAGE_SIZE, 0.02, qt=True)
        (out["loss_qt"] if qt_only else out["loss_F"]).backward()
        ref = g[pre + key]
        ours_g = logits.grad.cpu().numpy()
        cos = (ours_g * ref).sum() / (np.linalg.norm(ours_g) * np.linalg.norm(ref))
        assert cos > 0.99999 and relerr(ours_g, ref) < 5e-4  # measured 4e-5: all of it the reference's own fp32 noise (below)
        # the fp64 truth of the same loss on the same inputs
        s64 = {k: v.double() for k, v in sc.items()}
        lg = s64["logits_layers"].clone().requires_grad_(True)
        o64 = oracle.deepf_forward(s64["matches_xy_ori"], IMAGE_SIZE, 5, logits_layers=lg)
        l64, _, _, E64 = oracle.f_loss(o64, s64["pts1_virt_ori"], s64["pts2_virt_ori"], s64["Ks"], 5, 0.02)
        if qt_only:
            pose = oracle.rt_loss(E64, s64["delta_Rtijs_4_4"], s64["qs_cam"], s64["ts_cam"])
            loss64 = oracle.qt_training_loss(pose["q_l2"], pose["t_l2"], 0.1, 0.5, 1.0, 0.1)
        else:
            loss64 = l64["loss_F"]
        truth, = torch.autograd.grad(loss64, lg)
        truth = truth.numpy()
        e_ours, e_ref = relerr(ours_g, truth), relerr(ref, truth)
        print(f"{key}: |ours - fp64 truth| = {e_ours:.2e}, |reference fp32 - fp64 truth| = {e_ref:.2e} (relative to the largest entry)")
        assert e_ours < 2e-5 and e_ref < 5e-4


def test_fused_step_equals_unfused_ops(dfepe):
    """The single-node fused step (softmax fused into the fit, scalar-coefficient adjoints, loss head) and the
    per-op autograd path are the same computation."""
    B, N, depth = 9, 100, 4
    sc = dfepe.pipeline.scene_to_device(dfepe.synth.make_scene(B, N, seed=33, outlier_ratio=0.3, depth_layers=depth), DEV)
    a = dfepe.pipeline.hot_path_step(sc, IMAGE_SIZE, depth, 0.02, qt=True, fused=True)
    b = dfepe.pipeline.hot_path_step(sc, IMAGE_SIZE, depth, 0.02, qt=True, fused=False)
    assert abs(a["loss"].item() - b["loss"].item()) < 1e-6
    assert (a["F_layers"] - b["F_layers"]).abs().max().item() < 2e-6 * b["F_layers"].abs().max().item()
    np.testing.assert_allclose(a["loss_layers"].cpu().numpy(), b["loss_layers"].detach().cpu().numpy(), rtol=1e-5)
    np.testing.assert_allclose(a["q_l2"].cpu().numpy(), b["q_l2"].detach().cpu().numpy(), atol=1e-6)
    for l in range(depth):
        np.testing.assert_allclose(a["weights_layers"][l].cpu().numpy(), b["weights_layers"][l].detach().cpu().numpy(), rtol=2e-6, atol=1e-9)
    assert relerr(a["grad_logits"].cpu().numpy(), b["grad_logits"].cpu().numpy()) < 2e-5
    # packed sums = what dist.pack_loss_sums builds from the tensors
    ref = dfepe.dist.pack_loss_sums(a["loss_sum"], 100, a["q_l2"], a["t_l2"], 0.1, 0.5)
    np.testing.assert_allclose(a["packed"].cpu().numpy(), ref.cpu().numpy(), rtol=1e-6)


@pytest.mark.parametrize("B,N,batched", [(37, 100, False), (300, 100, True), (5, 200, False)])
def test_deferred_loss_head_gives_the_same_step(dfepe, B, N, batched):
    """defer_loss_head: the batch sums of the loss tail are finished by the first backward launch (three spare wavefronts of
    its first workgroup for the row kernels; a launch behind it for the cooperative kernels, N = 200 at B = 5) instead
    of a launch of their own.  After backward every output and the gradients are those of the undeferred step, bit for bit."""
    depth = 5
    sc = dfepe.pipeline.scene_to_device(dfepe.synth.make_scene(B, N, seed=41, outlier_ratio=0.3, depth_layers=depth), DEV)

    def run(defer):
        logits = sc["logits_layers"][:depth].detach().clone().requires_grad_(True)
        o = dfepe.pipeline.hot_path_fused(sc["matches_xy_ori"], logits, sc["Ks"], sc["pts1_virt_ori"], sc["pts2_virt_ori"], sc["qs_cam"],
                                          sc["ts_cam"], sc["R_gt"], IMAGE_SIZE, 0.02, True, layers_batched=batched, defer_loss_head=defer)
        (o["loss"] * 1.5).backward()
        torch.cuda.synchronize()
        o["grad_logits"] = logits.grad
        return o

    a, b = run(False), run(True)
    for k in ("loss", "loss_F", "loss_qt", "loss_layers", "packed", "loss_sum", "q_l2", "t_l2", "grad_logits", "F_layers", "E_layers"):
        assert torch.equal(a[k], b[k]), k
    # without a backward (no gradient requested) the head is not deferred: the scalars are there after the forward
    with torch.no_grad():
        c = dfepe.pipeline.hot_path_fused(sc["matches_xy_ori"], sc["logits_layers"][:depth], sc["Ks"], sc["pts1_virt_ori"], sc["pts2_virt_ori"],
                                          sc["qs_cam"], sc["ts_cam"], sc["R_gt"], IMAGE_SIZE, 0.02, True, defer_loss_head=True)
    assert torch.equal(c["loss"], a["loss"]) and torch.equal(c["packed"], a["packed"])


@pytest.mark.parametrize("B,depth", [(40, 9), (40, 16), (4096, 9), (4100, 3)])
def test_loss_head_layer_groups_and_workgroup_counts(dfepe, B, depth):
    """The loss head walks the layers of a kind in groups of eight and has a fast path for workgroup counts that are multiples
    of 256 (B = 4096 -> 256 workgroups of the tail): more than eight layers, and workgroup counts on both sides of that
    condition, deferred and not, against the sums formed from the per-pair outputs."""
    N = 20
    sc = dfepe.pipeline.scene_to_device(dfepe.synth.make_scene(B, N, seed=77 + depth, outlier_ratio=0.3, depth_layers=depth), DEV)
    for defer in (False, True):
        logits = sc["logits_layers"][:depth].detach().clone().requires_grad_(True)
        o = dfepe.pipeline.hot_path_fused(sc["matches_xy_ori"], logits, sc["Ks"], sc["pts1_virt_ori"], sc["pts2_virt_ori"], sc["qs_cam"],
                                          sc["ts_cam"], sc["R_gt"], IMAGE_SIZE, 0.02, True, defer_loss_head=defer)
        o["loss"].backward()
        torch.cuda.synchronize()
        M = sc["pts1_virt_ori"].shape[1]
        ref = dfepe.dist.pack_loss_sums(o["loss_sum"], M, o["q_l2"], o["t_l2"], 0.1, 0.5)
        np.testing.assert_allclose(o["packed"].cpu().numpy(), ref.cpu().numpy(), rtol=1e-12)
        np.testing.assert_allclose(o["loss_layers"].cpu().numpy(), (o["loss_sum"].double().sum(1) / (B * M)).cpu().numpy(), rtol=1e-6)
        loss_F = o["loss_sum"].double().sum() / (B * M * depth)
        loss_qt = (o["q_l2"].double().clamp(max=0.1).sum() * 1.0 + o["t_l2"].double().clamp(max=0.5).sum() * 0.1) / (B * depth)
        np.testing.assert_allclose(o["loss"].item(), (loss_F + loss_qt).item(), rtol=1e-6)


@pytest.mark.parametrize("qt,balance_F", [(True, 1.0), (True, 0.0), (False, 1.0), (True, 0.3)])
def test_fused_loss_tail_equals_the_five_kernel_tail(dfepe, qt, balance_F):
    """dfepe_loss_tail (one launch: F-loss + E + pose + loss head + d loss / d F, unit upstream, g_scale applied by w8pt_bwd)
    against the round-1 tail (floss_fwd, pose_fwd, loss_head | pose_bwd, floss_bwd) on the same fits, with a non-unit
    upstream gradient; balance_F = 0 is the reference's qt-only objective (Train_model_pipeline.py:580-587)."""
    B, N, depth = 37, 100, 5
    sc = dfepe.pipeline.scene_to_device(dfepe.synth.make_scene(B, N, seed=35, outlier_ratio=0.3, depth_layers=depth), DEV)

    def run(**kw):
        logits = sc["logits_layers"][:depth].detach().clone().requires_grad_(True)
        o = dfepe.pipeline.hot_path_fused(sc["matches_xy_ori"], logits, sc["Ks"], sc["pts1_virt_ori"], sc["pts2_virt_ori"], sc["qs_cam"],
                                          sc["ts_cam"], sc["R_gt"], IMAGE_SIZE, 0.02, qt, **kw)
        (o["loss"] * 2.5).backward()
        o["grad_logits"] = logits.grad
        return o

    a = run(balance_F=balance_F, fused_tail=True)
    if balance_F == 1.0:
        b = run(fused_tail=False)
        assert abs(a["loss"].item() - b["loss"].item()) < 1e-6
        assert relerr(a["grad_logits"].cpu().numpy(), b["grad_logits"].cpu().numpy()) < 2e-5
    else:  # the unfused tail only knows loss_F + loss_qt: combine its two pure cases linearly
        b = run(fused_tail=False)
        lf, lq = b["loss_F"].item(), b["loss_qt"].item()
        assert abs(a["loss"].item() - (balance_F * lf + lq)) < 1e-6
        c = run(balance_F=1.0, fused_tail=True)           # g(loss_F + loss_qt)
        sc0 = run(balance_F=0.0, fused_tail=True) if balance_F != 0.0 else a  # g(loss_qt)
        gF_only = c["grad_logits"] - sc0["grad_logits"]
        expect = balance_F * gF_only + sc0["grad_logits"]
        assert relerr(a["grad_logits"].cpu().numpy(), expect.cpu().numpy()) < 5e-5
    for k in ("loss_sum", "E_layers", "F_layers") + (("q_l2", "t_l2", "R_deg", "t_deg") if qt else ()):
        np.testing.assert_allclose(a[k].cpu().numpy(), b[k].cpu().numpy(), rtol=2e-6, atol=2e-6, err_msg=k)
    if qt:
        assert torch.equal(a["sel"], b["sel"])
    np.testing.assert_allclose(a["packed"].cpu().numpy(), b["packed"].cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose(a["loss_layers"].cpu().numpy(), b["loss_layers"].cpu().numpy(), rtol=1e-6)
    # the completion ticket is back at zero and a second call gives the same bits (deterministic batch sums)
    a2 = run(balance_F=balance_F, fused_tail=True)
    assert torch.equal(a2["packed"], a["packed"]) and torch.equal(a2["grad_logits"], a["grad_logits"])


def test_layers_batched_launch_is_bit_identical(dfepe):
    """n_weight_sets = L (all layers' weightings of the same pairs in one grid) runs the same per-wave program as L
    separate launches: every output and the logits gradient are bit-identical.  Point gradients are refused there."""
    B, N, depth = 37, 100, 5
    sc = dfepe.pipeline.scene_to_device(dfepe.synth.make_scene(B, N, seed=7, outlier_ratio=0.4, depth_layers=depth), DEV)
    a = dfepe.pipeline.hot_path_step(sc, IMAGE_SIZE, depth, 0.02, qt=True, fused=True)
    b = dfepe.pipeline.hot_path_step(sc, IMAGE_SIZE, depth, 0.02, qt=True, fused=True, layers_batched=True)
    for k in ("F_layers", "loss_layers", "q_l2", "t_l2", "packed", "grad_logits", "R_deg", "t_deg"):
        assert torch.equal(a[k], b[k]), k
    for l in range(depth):
        assert torch.equal(a["weights_layers"][l], b["weights_layers"][l])
    L = dfepe._lib.lib()
    m, w = sc["matches_xy_ori"], torch.softmax(sc["logits_layers"][:depth], dim=2).contiguous()
    buf = torch.empty(depth * B * 128, device=DEV)
    rc = L.dfepe_w8pt_bwd(m.data_ptr(), None, w.data_ptr(), B, N, depth, 1, 1241.0, 376.0, 0.5, buf.data_ptr(), buf.data_ptr(),
                          buf.data_ptr(), None, None, None, None, buf.data_ptr(), buf.data_ptr(), None, None, None)
    assert rc == -3  # DFEPE_ERR_UNSUPPORTED: a point gradient would have to be summed over the sets


@pytest.mark.parametrize("N", [448, 600])
def test_cooperative_workgroup_path_in_the_training_step(dfepe, oracle, N):
    """N so large that the forward switches to one 256-thread workgroup per pair (four wavefronts share the
    per-correspondence phases): fused softmax, save record, backward and the layers-batched launch all go through it."""
    B, depth = 5, 2
    sc = dfepe.synth.make_scene(B, N, seed=N, outlier_ratio=0.3, depth_layers=depth)
    dev = dfepe.pipeline.scene_to_device(sc, DEV)
    ours = dfepe.pipeline.hot_path_step(dev, IMAGE_SIZE, depth, 0.02, qt=True)
    ref = oracle.hot_path_step({k: v.double() for k, v in sc.items()}, IMAGE_SIZE, depth, 0.02, qt=True, mode="batched")
    assert abs(ours["loss"].item() - ref["loss"].item()) < 1e-5
    assert relerr(ours["grad_logits"].cpu().numpy(), ref["grad_logits"].numpy()) < 2e-3
    b = dfepe.pipeline.hot_path_step(dev, IMAGE_SIZE, depth, 0.02, qt=True, layers_batched=True)
    for k in ("F_layers", "packed", "grad_logits"):
        assert torch.equal(ours[k], b[k]), k


def test_logits_fused_fit_matches_softmax_then_fit(dfepe):
    B, N = 6, 100
    sc = dfepe.synth.make_scene(B, N, seed=12, outlier_ratio=0.2)
    g = torch.Generator().manual_seed(5)
    m = sc["matches_xy_ori"].to(DEV)
    GF, GR, GE, GW = (torch.randn(s, generator=g).to(DEV) for s in ((B, 3, 3), (B, N), (B, N), (B, N)))
    la = sc["logits_layers"][0].to(DEV).requires_grad_(True)
    F, res, epi, w = dfepe.ops.w8pt_raw_logits(m, la, 1241, 376)
    ((F * GF).sum() + (res * GR).sum() + (epi * GE).sum() + (w * GW).sum()).backward()
    lb = sc["logits_layers"][0].to(DEV).requires_grad_(True)
    wb = torch.softmax(lb, dim=1)
    F2, res2, epi2 = dfepe.ops.w8pt_raw(m, wb, 1241, 376)
    ((F2 * GF).sum() + (res2 * GR).sum() + (epi2 * GE).sum() + (wb * GW).sum()).backward()
    np.testing.assert_allclose(w.detach().cpu().numpy(), wb.detach().cpu().numpy(), rtol=2e-6, atol=1e-10)
    assert (F - F2).abs().max().item() < 2e-6 * F2.abs().max().item()
    assert relerr(la.grad.cpu().numpy(), lb.grad.cpu().numpy()) < 2e-5


def test_floss_many_virtual_points_generic_path(dfepe, oracle):
    """M > 128 takes the non-cached code path of floss: same answer as the oracle."""
    L, B, M = 2, 3, 300
    sc = dfepe.synth.make_scene(B, 20, seed=8, M_virt=M)
    T = oracle.hw_matrix(IMAGE_SIZE, torch.float64)
    Tinv = torch.linalg.inv(T)
    Fn = Tinv.T @ sc["F_gt"].double() @ Tinv
    Fn = Fn / Fn.flatten(1).norm(dim=1)[:, None, None]
    g = torch.Generator().manual_seed(1)
    Fl = torch.stack([Fn + 0.004 * (l + 1) * torch.randn(B, 3, 3, generator=g, dtype=torch.float64) for l in range(L)]).float().double()
    outs = {"T1": T.expand(B, 3, 3), "T2": T.expand(B, 3, 3), "out_layers": [Fl[l] for l in range(L)], "F_est": Fl[-1]}
    losses, _, _, E_layers = oracle.f_loss(outs, sc["pts1_virt_ori"].double(), sc["pts2_virt_ori"].double(), sc["Ks"].double(), L, 0.02)
    loss_sum, E = dfepe.ops.floss(Fl.float().to(DEV), T.float().to(DEV), T.float().to(DEV), sc["Ks"].to(DEV), sc["pts1_virt_ori"].to(DEV), sc["pts2_virt_ori"].to(DEV), 0.02)
    assert relerr(loss_sum.cpu().numpy(), (losses["loss_per_pair"] * M).numpy()) < 5e-5
    assert relerr(E.cpu().numpy(), torch.stack(E_layers).numpy()) < 2e-6


@pytest.mark.parametrize("use_epi", [False, True])
def test_point_gradients_vs_oracle_autograd(dfepe, oracle, use_epi):
    """d/d(pts1), d/d(pts2) of the fit (rows, Hartley transforms, de-normalisation, residual's direct dependence) and
    d/d(matches) of the raw-matches entry, against fp64 autograd of the oracle."""
    B, N = 5, 100
    sc = dfepe.synth.make_scene(B, N, seed=41, outlier_ratio=0.2)
    g = torch.Generator().manual_seed(3)
    GF, GR, GE = torch.randn(B, 3, 3, generator=g), torch.randn(B, N, generator=g), torch.randn(B, N, generator=g)
    w = torch.softmax(sc["logits_layers"][0], 1)
    p1, p2, _ = oracle.normalize_hw(sc["matches_xy_ori"], IMAGE_SIZE)
    p1 = p1.clone(); p2 = p2.clone()
    p1[:, :, 2] += 0.05 * torch.randn(B, N, generator=g)  # general homogeneous coordinate
    p2[:, :, 2] += 0.05 * torch.randn(B, N, generator=g)
    a1, a2, aw = p1.to(DEV).requires_grad_(True), p2.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    outs = dfepe.ops.w8pt(a1, a2, aw, clamp_at=0.5, want_epi=True)
    loss = (outs[0] * GF.to(DEV)).sum() + (outs[1] * GR.to(DEV)).sum() + ((outs[2] * GE.to(DEV)).sum() if use_epi else 0.0)
    loss.backward()
    o1, o2, ow = p1.double().requires_grad_(True), p2.double().requires_grad_(True), w.double().requires_grad_(True)
    o_out, o_res, _ = oracle.fit_forward(o1, o2, ow.unsqueeze(1))
    s = torch.sign((o_out.detach() * outs[0].detach().cpu().double()).flatten(1).sum(1))
    lo = (s[:, None, None] * o_out * GF.double()).sum() + (s[:, None] * o_res * GR.double()).sum()
    if use_epi:
        lo = lo + (oracle.compute_epi_residual(o1, o2, o_out, 0.5) * GE.double()).sum()
    lo.backward()
    assert relerr(aw.grad.cpu().numpy(), ow.grad.numpy()) < 2e-3
    assert relerr(a1.grad.cpu().numpy(), o1.grad.numpy()) < 2e-3
    assert relerr(a2.grad.cpu().numpy(), o2.grad.numpy()) < 2e-3
    # raw pixel matches: chain through the image-size normalisation
    m = sc["matches_xy_ori"].to(DEV).requires_grad_(True)
    outs = dfepe.ops.w8pt_raw(m, w.to(DEV), 1241, 376, clamp_at=0.5, want_epi=True)
    ((outs[0] * GF.to(DEV)).sum() + (outs[1] * GR.to(DEV)).sum() + ((outs[2] * GE.to(DEV)).sum() if use_epi else 0.0)).backward()
    mo = sc["matches_xy_ori"].double().requires_grad_(True)
    q1, q2, _ = oracle.normalize_hw(mo, IMAGE_SIZE)
    o_out, o_res, _ = oracle.fit_forward(q1, q2, w.double().unsqueeze(1))
    s = torch.sign((o_out.detach() * outs[0].detach().cpu().double()).flatten(1).sum(1))
    lo = (s[:, None, None] * o_out * GF.double()).sum() + (s[:, None] * o_res * GR.double()).sum()
    if use_epi:
        lo = lo + (oracle.compute_epi_residual(q1, q2, o_out, 0.5) * GE.double()).sum()
    lo.backward()
    assert relerr(m.grad.cpu().numpy(), mo.grad.numpy()) < 2e-3


@pytest.mark.parametrize("balance_F", [0.0, 0.3])
def test_unfused_tail_honours_balance_F(dfepe, balance_F):
    """fused_tail=False (and the forced fall-backs: M > 128 virtual points, depth > 16) mixes balance_F * loss_F + loss_qt like the
    one-launch tail; balance_F = 0 is the reference's real qt objective (Train_model_pipeline.py:580-587)."""
    B, N, depth = 21, 100, 3
    sc = dfepe.pipeline.scene_to_device(dfepe.synth.make_scene(B, N, seed=36, outlier_ratio=0.3, depth_layers=depth), DEV)
    a = dfepe.pipeline.hot_path_step(sc, IMAGE_SIZE, depth, 0.02, qt=True, balance_F=balance_F, fused_tail=True)
    b = dfepe.pipeline.hot_path_step(sc, IMAGE_SIZE, depth, 0.02, qt=True, balance_F=balance_F, fused_tail=False)
    assert abs(a["loss"].item() - b["loss"].item()) < 1e-6
    assert relerr(a["grad_logits"].cpu().numpy(), b["grad_logits"].cpu().numpy()) < 5e-5


def test_limits_of_the_loss_launches_are_checked_before_anything_runs(dfepe):
    """More than 16 layers: refused up front with the limit in the message (every loss kernel stacks <= 16 layers per launch).
    More than 128 virtual points: the five-kernel tail takes over silently, with any balance_F."""
    B, N = 6, 100
    sc = dfepe.pipeline.scene_to_device(dfepe.synth.make_scene(B, N, seed=37, outlier_ratio=0.2, depth_layers=18), DEV)
    with pytest.raises(dfepe.DfepeError, match="depth 18 > 16"):
        dfepe.pipeline.hot_path_step(sc, IMAGE_SIZE, 18, 0.02, qt=True)
    sc = dfepe.pipeline.scene_to_device(dfepe.synth.make_scene(B, N, seed=38, outlier_ratio=0.2, depth_layers=3), DEV)
    big = dict(sc)
    big["pts1_virt_ori"] = torch.cat((sc["pts1_virt_ori"], sc["pts1_virt_ori"][:, :60]), 1).contiguous()  # M = 160
    big["pts2_virt_ori"] = torch.cat((sc["pts2_virt_ori"], sc["pts2_virt_ori"][:, :60]), 1).contiguous()
    o = dfepe.pipeline.hot_path_step(big, IMAGE_SIZE, 3, 0.02, qt=True, balance_F=0.3)
    ref = dfepe.pipeline.hot_path_step(big, IMAGE_SIZE, 3, 0.02, qt=True, fused=False)
    want = 0.3 * ref["loss_F"].item() + ref["loss_qt"].item()
    assert abs(o["loss"].item() - want) < 1e-6
    assert torch.isfinite(o["grad_logits"]).all()
