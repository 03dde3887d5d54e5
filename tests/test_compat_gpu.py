"""The reference-API mirror (pytorch-deepfepe_amd/compat) against golden vectors produced by the reference itself
and against the CPU oracle.  GPU box only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
IMAGE_SIZE = [376, 1241, 3]
DEV = "cuda:0"


def T(x, dt=torch.float32):
    return torch.from_numpy(np.asarray(x)).to(dt)


def unit_align(a, ref):
    a = np.asarray(a, dtype=np.float64).reshape(a.shape[0], -1)
    r = np.asarray(ref, dtype=np.float64).reshape(ref.shape[0], -1)
    a = a / np.linalg.norm(a, axis=1, keepdims=True)
    r = r / np.linalg.norm(r, axis=1, keepdims=True)
    s = np.sign((a * r).sum(1, keepdims=True))
    s[s == 0] = 1
    return a * s, r, s[:, 0]


def test_fit_module_matches_reference_golden(dfepe, golden):
    g = golden("fit")
    fit = dfepe.compat.DeepFNet.Fit(is_cuda=True, if_cpu_svd=True)
    for kind in ("general", "outlier40", "dense1000"):
        out, res = fit(T(g[f"{kind}_f32_pts1"]).to(DEV), T(g[f"{kind}_f32_pts2"]).to(DEV), T(g[f"{kind}_f32_weights"]).to(DEV))
        assert out.shape == g[f"{kind}_f32_out"].shape and res.shape == g[f"{kind}_f32_residual"].shape
        a, r, s = unit_align(out.cpu().numpy(), g[f"{kind}_f32_out"])
        assert np.linalg.norm(a - r, axis=1).max() < 2e-5
        np.testing.assert_allclose(res.cpu().numpy() * s[:, None], g[f"{kind}_f32_residual"], atol=3e-6, rtol=2e-3)
    norm = dfepe.compat.DeepFNet.NormalizeAndExpand_HW(IMAGE_SIZE)
    p1, p2, T1, T2 = norm(T(g["general_f32_matches"]).to(DEV))
    np.testing.assert_allclose(p1.permute(0, 2, 1).cpu().numpy(), g["general_f32_pts1"], atol=2e-6)
    np.testing.assert_allclose(T2.cpu().numpy(), g["general_f32_T_hw"], atol=1e-7)


def test_deepfnet_full_model_matches_reference_golden(dfepe, golden):
    """Whole recurrent model with seeded estimators (tests/golden/pipeline.npz 'net_*', depth 3): same state_dict keys,
    same parameters (checksum), same logits / F / weights per layer, same F-loss and same parameter gradients.
    The SVD sign gauge of the reference (LAPACK's, arbitrary) is imposed on our fit outputs so that the next
    estimator layer sees the same `residual` channel the reference saw."""
    g = golden("pipeline")
    depth = 3
    net = dfepe.compat.DeepFNet.DeepFNet(depth=depth, image_size=IMAGE_SIZE, if_quality=False, if_cpu_svd=True)
    dfepe.synth.fill_params_deterministic(net, seed=5)
    assert sorted(net.state_dict().keys()) == [str(k) for k in g["net_state_keys"]]
    chk = np.array([float(p.detach().double().abs().sum()) for _, p in sorted(net.named_parameters())])
    np.testing.assert_allclose(chk, g["net_param_checksum"], rtol=1e-6)
    net = net.to(DEV)
    layer = {"i": 0}
    orig_fit = net._fit

    def fit_with_reference_gauge(matches, logits, data_batch, want_epi, dst=None):
        outs = orig_fit(matches, logits, data_batch, want_epi, dst)
        ref = torch.from_numpy(g["net_out_layers"][layer["i"]]).to(DEV)
        s = torch.sign((outs[0].detach() * ref).flatten(1).sum(1))
        layer["i"] += 1
        return (outs[0] * s[:, None, None], outs[1] * s[:, None]) + tuple(outs[2:])

    net._fit = fit_with_reference_gauge
    batch = {"matches_xy_ori": T(g["net_matches_xy_ori"]).to(DEV), "matches_good_unique_nums": None, "t_scene_scale": None}
    outs = net(batch)
    assert set(outs.keys()) == {"logits", "logits_layers", "F_est", "epi_res_layers", "T1", "T2", "out_layers", "pts1", "pts2",
                                "weights", "residual_layers", "weights_layers"}
    # What sets the tolerances below: the estimator is ~0.8 GFLOP of fp32 GEMMs + InstanceNorms per pair, and two fp32
    # implementations of the SAME network differ by their summation order.  Measured here: this package's fused estimator
    # (split-bf16 MFMA GEMMs) against its own stock-PyTorch estimator (conv1d, MIOpen) with identical parameters -- the distance
    # to the reference's run is required to stay within a small multiple of that fp32 reordering noise.
    net_b = dfepe.compat.DeepFNet.DeepFNet(depth=depth, image_size=IMAGE_SIZE, if_quality=False, if_cpu_svd=True, fused_estimator=False)
    net_b.load_state_dict(net.state_dict())
    net_b = net_b.to(DEV)
    layer_b = {"i": 0}
    fit_b = net_b._fit

    def fit_b_gauge(matches, logits, data_batch, want_epi, dst=None):
        o = fit_b(matches, logits, data_batch, want_epi, dst)
        ref = torch.from_numpy(g["net_out_layers"][layer_b["i"]]).to(DEV)
        sg = torch.sign((o[0].detach() * ref).flatten(1).sum(1))
        layer_b["i"] += 1
        return (o[0] * sg[:, None, None], o[1] * sg[:, None]) + tuple(o[2:])

    net_b._fit = fit_b_gauge
    with torch.no_grad():
        outs_b = net_b(batch)
    noise = max(float((outs["logits_layers"][l].detach() - outs_b["logits_layers"][l]).abs().max()) for l in range(depth))
    dist = max(float(np.abs(outs["logits_layers"][l].detach().cpu().numpy() - g["net_logits_layers"][l]).max()) for l in range(depth))
    print(f"logits: |fused - stock estimator| (same parameters, both fp32) = {noise:.2e}; |ours - reference| = {dist:.2e}")
    assert dist < max(20 * noise, 2e-3)
    for l in range(depth):
        a, r, _ = unit_align(outs["out_layers"][l].detach().cpu().numpy(), g["net_out_layers"][l])
        # tolerances = ~10-20 x the distances measured on an MI355X (scripts/measure_golden_tolerances.py: F 1e-5, logits 8e-5,
        # weights 2e-6, residual 8e-8, epipolar residual 3e-5, loss 9e-7 relative, gradient norms 8e-6, cosine 0.99999994)
        assert np.linalg.norm(a - r, axis=1).max() < 2e-4, l
        np.testing.assert_allclose(outs["logits_layers"][l].detach().cpu().numpy(), g["net_logits_layers"][l], atol=1e-3, rtol=1e-3)
        np.testing.assert_allclose(outs["weights_layers"][l].detach().cpu().numpy(), g["net_weights_layers"][l], atol=3e-5, rtol=2e-3)
        np.testing.assert_allclose(outs["residual_layers"][l].detach().cpu().numpy(), g["net_residual_layers"][l], atol=2e-6, rtol=2e-3)
    for l in range(depth - 1):
        np.testing.assert_allclose(outs["epi_res_layers"][l].detach().cpu().numpy(), g["net_epi_res_layers"][l], atol=5e-4, rtol=2e-3)
    loss_params = {"depth": depth, "clamp_at": 0.02, "if_tri_depth": False, "if_sample_loss": False, "topK": 8, "matches_good_unique_nums": None}
    losses, E_ests, F_ests, _, _, _, E_layers = dfepe.compat.train_good_utils.get_all_loss_DeepF(
        outs, T(g["net_pts1_virt_ori"]).to(DEV), T(g["net_pts2_virt_ori"]).to(DEV), T(g["net_Ks"]).to(DEV), loss_params,
        get_residual_summaries=False)
    np.testing.assert_allclose(losses["loss_F"].item(), g["net_loss_F"], rtol=1e-4)
    losses["loss_F"].backward()
    # conv biases that feed an InstanceNorm cancel exactly; the fused estimator gives them an exact zero gradient (the reference's ~0)
    gn = {n: (0.0 if p.grad is None else float(p.grad.double().norm())) for n, p in net.named_parameters()}
    ours = np.array([gn[n] for n in sorted(gn)])
    np.testing.assert_allclose(ours, g["net_grad_norms"], rtol=5e-3, atol=1e-4 * g["net_grad_norms"].max())
    ga = net.input_weights.fw[0].weight.grad.cpu().numpy().ravel()
    gr = g["net_grad_first_conv"].ravel()
    assert (ga * gr).sum() / (np.linalg.norm(ga) * np.linalg.norm(gr)) > 0.99999


def test_deepfnet_sign_gauge_departure_is_bounded_and_documented(dfepe, golden):
    """The reference feeds the SIGNED residual of each fit to the next estimator layer, and the sign is LAPACK's arbitrary
    one (it has no convention); this library orients f by its largest component.  The recurrent outputs therefore differ
    from a reference run wherever the two signs differ: this test measures by how much on the
    golden model (seeded random weights) -- layer 1 is gauge-free and agrees tightly; later layers agree on the sign-invariant
    quantities only as far as the estimator's sensitivity to that one input channel allows."""
    g = golden("pipeline")
    depth = 3
    net = dfepe.compat.DeepFNet.DeepFNet(depth=depth, image_size=IMAGE_SIZE, if_quality=False, if_cpu_svd=True)
    dfepe.synth.fill_params_deterministic(net, seed=5)
    net = net.to(DEV)
    batch = {"matches_xy_ori": T(g["net_matches_xy_ori"]).to(DEV), "matches_good_unique_nums": None, "t_scene_scale": None}
    with torch.no_grad():
        outs = net(batch)
    a, r, s = unit_align(outs["out_layers"][0].cpu().numpy(), g["net_out_layers"][0])
    assert np.linalg.norm(a - r, axis=1).max() < 1e-3  # first fit: its weights do not depend on any sign
    flipped = (s < 0).mean()
    dep = []
    for l in range(1, depth):
        a, r, _ = unit_align(outs["out_layers"][l].cpu().numpy(), g["net_out_layers"][l])
        dep.append(np.linalg.norm(a - r, axis=1).max())
    # with every pair's sign equal to the reference's there is no departure beyond fp32 noise; otherwise it is finite and
    # of the size the estimator's response to a flipped residual channel gives (recorded here, not asserted to be small)
    assert all(np.isfinite(dep))
    if flipped == 0.0:
        assert max(dep) < 1e-3
    print(f"sign gauge: {100 * flipped:.0f} % of the pairs oriented opposite to the reference's LAPACK run; "
          f"max unit-Frobenius departure of the later layers' F: {max(dep):.3e}")


def test_shape_validation(dfepe):
    """Wrong layouts raise ValueError before anything is launched (the kernels index raw pointers)."""
    B, N = 3, 20
    w = torch.rand(B, N, device=DEV)
    with pytest.raises(ValueError):
        dfepe.ops.w8pt(torch.rand(B, N, 2, device=DEV), torch.rand(B, N, 2, device=DEV), w)  # [B,N,2] instead of homogeneous [B,N,3]
    with pytest.raises(ValueError):
        dfepe.ops.w8pt_raw(torch.rand(B, N, 3, device=DEV), w, 1241, 376)
    with pytest.raises(ValueError):
        dfepe.ops.floss(torch.rand(2, B, 3, 3, device=DEV), torch.eye(3, device=DEV), torch.eye(3, device=DEV), torch.rand(3, 3, device=DEV),
                        torch.rand(B, 10, 3, device=DEV), torch.rand(B, 10, 3, device=DEV), 0.02)  # one shared K instead of [B,3,3]
    with pytest.raises(ValueError):
        dfepe.ops.floss(torch.rand(2, B, 3, 3, device=DEV), torch.eye(3, device=DEV), torch.eye(3, device=DEV), torch.rand(B, 3, 3, device=DEV),
                        torch.rand(B, 10, 3, device=DEV), torch.rand(B, 12, 3, device=DEV), 0.02)
    with pytest.raises(ValueError):
        dfepe.ops.pose_errors(torch.rand(2, B, 3, 3, device=DEV), torch.rand(B + 1, 4, device=DEV), torch.rand(B, 3, device=DEV), torch.rand(B, 3, 3, device=DEV))
    with pytest.raises(ValueError):
        dfepe.ops.cheirality(torch.rand(B, 3, 3, device=DEV), torch.rand(1, 3, 3, device=DEV), torch.rand(B, N, 4, device=DEV))
    with pytest.raises(ValueError):
        dfepe.ops.epi_residual(torch.rand(B, N, 3, device=DEV), torch.rand(B, N, 3, device=DEV), torch.rand(B + 1, 3, 3, device=DEV))


def test_legacy_helpers_are_differentiable_like_the_references(dfepe, oracle):
    """The reference's _sym_epi_dist, _sampson_dist, _epi_distance, compute_epi_residual, _get_M2s, _R_to_q, _F_from_XY, _E_from_XY
    are differentiable torch code (utils_F.py:104-155,223-275,291-361,400-413,478-498; utils_geo.py:58-86; used with gradients by
    train_good_utils.py:55-61 and dsac_tools/dsac.py:138-176).  The mirrors launch kernels without an adjoint, so a call whose
    inputs require grad is evaluated by compat._autograd_paths instead (VERDICT r4 missing #3): its VALUES must equal the kernels'
    and its GRADIENTS float64 autograd of the oracle."""
    uF, uG = dfepe.compat.utils_F, dfepe.compat.utils_geo
    sc = dfepe.synth.make_scene(4, 60, seed=12, outlier_ratio=0.1, noise_px=0.5)
    m, K = sc["matches_xy_ori"], sc["Ks"]
    Fgt = sc["F_gt"] / sc["F_gt"].flatten(1).norm(dim=1)[:, None, None]

    def both(fn_dev, fn_ref, args, pick=lambda o: o):
        """fn(*args) on the device with every arg requiring grad vs the oracle in float64: values and all gradients."""
        dev_args = [a.float().to(DEV).requires_grad_(True) for a in args]
        ref_args = [a.double().requires_grad_(True) for a in args]
        od, orf = pick(fn_dev(*dev_args)), pick(fn_ref(*ref_args))
        plain = pick(fn_dev(*[a.detach() for a in dev_args]))  # the kernel path
        scale = float(orf.detach().abs().max()) + 1e-30
        assert float((od.detach().cpu().double() - orf.detach()).abs().max()) < 2e-4 * scale
        assert float((plain.cpu().double() - orf.detach()).abs().max()) < 2e-4 * scale
        gen = torch.Generator().manual_seed(3)
        wgt = torch.rand(orf.shape, generator=gen, dtype=torch.float64)
        (od * wgt.float().to(DEV)).sum().backward()
        (orf * wgt).sum().backward()
        for a, b in zip(dev_args, ref_args):
            assert a.grad is not None
            gs = float(b.grad.abs().max()) + 1e-30
            assert float((a.grad.cpu().double() - b.grad).abs().max()) < 2e-3 * gs, (fn_dev.__name__, float((a.grad.cpu().double() - b.grad).abs().max()), gs)

    X, Y = m[..., :2], m[..., 2:]
    both(uF._sym_epi_dist, oracle.sym_epi_dist, [Fgt, X, Y])
    both(uF._sampson_dist, oracle.sampson_dist, [Fgt, X, Y])
    both(uF._epi_distance, oracle.epi_distance, [Fgt, X, Y], pick=lambda o: o[0])
    both(uF._sym_epi_dist, oracle.sym_epi_dist, [Fgt[0], X[0], Y[0]])
    p1, p2, _ = oracle.normalize_hw(m.double(), IMAGE_SIZE)
    both(lambda a, b, F: uF.compute_epi_residual(a, b, F, 0.5), lambda a, b, F: oracle.compute_epi_residual(a, b, F, 0.5), [p1, p2, Fgt])
    # decomposition / quaternion: the torch gauge on both sides (the reference's own)
    # a true essential matrix has two EQUAL singular values: the gradient of its SVD is singular there (torch's own, in the reference
    # too); the comparison uses a generic matrix with separated singular values
    E = sc["E_gt"][0] / sc["E_gt"][0].norm() + 0.3 * torch.randn(3, 3, generator=torch.Generator().manual_seed(2))
    # (the ORDER of the four candidates follows the SVD's sign gauge -- rocSOLVER's on the device, LAPACK's in the oracle --, their SET
    # does not: compare a symmetric function of the four [R|t])
    cvec = torch.linspace(-1.0, 1.0, 12, dtype=torch.float64)

    def sym(Ms):
        Ms = torch.stack(Ms).reshape(4, 12)
        proj = (Ms * cvec.to(Ms.device, Ms.dtype)).sum(1)
        return torch.stack((proj.square().sum(), proj.pow(4).sum()))

    both(lambda e: sym(uF._get_M2s(e)[2]), lambda e: sym([torch.cat((R, t), 1) for R in oracle.get_M2s(e)[0] for t in oracle.get_M2s(e)[1]]), [E])
    R = torch.linalg.inv(sc["delta_Rtijs_4_4"].double())[0, :3, :3]
    both(uG._R_to_q, oracle.R_to_q, [R])
    # textbook solvers, gradients w.r.t. the points (sign gauge: torch.linalg.svd on both sides)
    Ki = torch.linalg.inv(K[0].double())
    xn = (torch.cat((X[0].double(), torch.ones(60, 1, dtype=torch.float64)), 1) @ Ki.T)[:, :2]
    yn = (torch.cat((Y[0].double(), torch.ones(60, 1, dtype=torch.float64)), 1) @ Ki.T)[:, :2]

    def unit(Mx):
        Mx = Mx / Mx.norm()
        return Mx * torch.sign(Mx.flatten()[Mx.abs().flatten().argmax()])

    dev_x = xn.float().to(DEV).requires_grad_(True)
    ref_x = xn.clone().requires_grad_(True)
    Fd = unit(uF._F_from_XY(dev_x, yn.float().to(DEV)))
    Fr = unit(oracle.F_from_XY(ref_x, yn))
    assert float((Fd.detach().cpu().double() - Fr.detach()).abs().max()) < 1e-3
    Fk = unit(uF._F_from_XY(dev_x.detach(), yn.float().to(DEV)))  # the kernel
    assert float((Fk.cpu().double() - Fr.detach()).abs().max()) < 1e-3
    Fd[0, 1].backward()
    Fr[0, 1].backward()
    assert float((dev_x.grad.cpu().double() - ref_x.grad).abs().max()) < 2e-2 * float(ref_x.grad.abs().max())
    Ed = uF._E_from_XY(dev_x.detach().requires_grad_(True), yn.float().to(DEV), K[0].to(DEV), if_normzliedK=True)
    assert Ed.requires_grad and torch.isfinite(Ed).all()


def test_epipolar_metrics_on_homogeneous_points(dfepe, oracle):
    """if_homo=True: homogeneous points are used as they are (third coordinate != 1 included), clamp_at=0.0 clamps to zero,
    clamp_at=None does not clamp -- like utils_F.py:291-361."""
    uF = dfepe.compat.utils_F
    g = torch.Generator().manual_seed(5)
    B, N = 3, 50
    F = torch.randn(B, 3, 3, generator=g)
    w1, w2 = 0.5 + torch.rand(B, N, 1, generator=g), 0.5 + torch.rand(B, N, 1, generator=g)
    X, Y = torch.randn(B, N, 2, generator=g), torch.randn(B, N, 2, generator=g)
    Xh, Yh = torch.cat((X, torch.ones(B, N, 1)), 2) * w1, torch.cat((Y, torch.ones(B, N, 1)), 2) * w2
    Fd, Xd, Yd = F.double(), Xh.double(), Yh.double()
    num = (Yd @ Fd @ Xd.transpose(1, 2)).diagonal(dim1=1, dim2=2)
    Fx1, Fx2 = Fd @ Xd.transpose(1, 2), Fd.transpose(1, 2) @ Yd.transpose(1, 2)
    a, b = Fx1[:, 0] ** 2 + Fx1[:, 1] ** 2, Fx2[:, 0] ** 2 + Fx2[:, 1] ** 2
    sym = num ** 2 * (1 / (a + 1e-10) + 1 / (b + 1e-10))
    out = uF._sym_epi_dist(F.to(DEV), Xh.to(DEV), Yh.to(DEV), if_homo=True)
    np.testing.assert_allclose(out.cpu().numpy(), sym.numpy(), rtol=2e-5)
    np.testing.assert_allclose(uF._sampson_dist(F.to(DEV), Xh.to(DEV), Yh.to(DEV), if_homo=True).cpu().numpy(), (num ** 2 / (a + b)).numpy(), rtol=2e-5)
    d = uF._epi_distance(F.to(DEV), Xh.to(DEV), Yh.to(DEV), if_homo=True)
    np.testing.assert_allclose(d[1].cpu().numpy(), (num.abs() / a.sqrt()).numpy(), rtol=2e-5)
    assert uF._sym_epi_dist(F.to(DEV), X.to(DEV), Y.to(DEV), clamp_at=0.0).abs().max().item() == 0.0
    assert uF._sym_epi_dist(F.to(DEV), X.to(DEV), Y.to(DEV), clamp_at=None).max().item() > 1.0
    np.testing.assert_allclose(uF._sym_epi_dist(F.to(DEV), X.to(DEV), Y.to(DEV)).cpu().numpy(),
                               oracle.sym_epi_dist(F.double(), X.double(), Y.double()).numpy(), rtol=2e-5)
    with pytest.raises(ValueError):
        uF._sym_epi_dist(F.to(DEV), X.to(DEV), Y.to(DEV), if_homo=True)  # 2-D points announced as homogeneous


def test_loss_functions_match_reference_golden(dfepe, golden):
    """get_all_loss_DeepF and get_Rt_loss fed with the reference's own per-layer F (golden 'solver_*')."""
    g = golden("pipeline")
    pre = "solver_"
    L = 5
    B = g[pre + "Ks"].shape[0]
    Thw = torch.tensor([[2.0 / 1241, 0, -1.0], [0, 2.0 / 376, -1.0], [0, 0, 1.0]]).expand(B, 3, 3).to(DEV)
    outs = {"weights": T(g[pre + "weights_layers"][-1]).to(DEV), "F_est": T(g[pre + "F_est"]).to(DEV), "T1": Thw, "T2": Thw,
            "out_layers": [T(x).to(DEV) for x in g[pre + "out_layers"]], "residual_layers": [T(x).to(DEV) for x in g[pre + "residual_layers"]],
            "weights_layers": [T(x).to(DEV) for x in g[pre + "weights_layers"]], "epi_res_layers": [T(x).to(DEV) for x in g[pre + "epi_res_layers"]]}
    loss_params = {"depth": L, "clamp_at": 0.02, "if_tri_depth": False, "if_sample_loss": False, "topK": 8,
                   "matches_good_unique_nums": [100] * B}
    ret = dfepe.compat.train_good_utils.get_all_loss_DeepF(outs, T(g[pre + "pts1_virt_ori"]).to(DEV), T(g[pre + "pts2_virt_ori"]).to(DEV),
                                                           T(g[pre + "Ks"]).to(DEV), loss_params, get_residual_summaries=True)
    assert len(ret) == 7
    losses, E_ests, F_ests, logits_softmax, rn, rnm, E_layers = ret
    np.testing.assert_allclose(torch.stack(losses["loss_layers"]).cpu().numpy(), g[pre + "loss_layers"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(losses["loss_F"].item(), g[pre + "loss_F"], rtol=1e-4)
    np.testing.assert_allclose(losses["loss_min_layers"].cpu().numpy(), g[pre + "loss_min_layers"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(losses["loss_min_batch"].cpu().numpy(), g[pre + "loss_min_batch"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(losses["loss_epi_res"].item(), g[pre + "loss_epi_res"], rtol=1e-4)
    np.testing.assert_allclose(torch.stack(E_layers).cpu().numpy(), g[pre + "E_layers"], rtol=2e-4, atol=2e-4 * np.abs(g[pre + "E_layers"]).max())
    np.testing.assert_allclose(E_ests.cpu().numpy(), g[pre + "E_ests"], rtol=2e-4, atol=2e-4 * np.abs(g[pre + "E_ests"]).max())
    np.testing.assert_allclose(F_ests.cpu().numpy(), g[pre + "F_ests"], rtol=2e-4, atol=2e-4 * np.abs(g[pre + "F_ests"]).max())
    for k in ("loss_residual", "loss_residual_topK", "loss_regW_clip", "loss_regW_entro", "loss_regW_entro_topK"):
        assert torch.isfinite(torch.as_tensor(losses[k])).all()
    rt = dfepe.compat.train_good_utils.get_Rt_loss([T(x).to(DEV) for x in g[pre + "E_layers"]], T(g[pre + "Ks"]), None, None,
                                                   T(g[pre + "delta_Rtijs_4_4"]), T(g[pre + "qs_cam"]).to(DEV), T(g[pre + "ts_cam"]).to(DEV), device=DEV)
    assert set(rt.keys()) == {"t_l2_error_mean", "q_l2_error_mean", "t_l2_error_list", "q_l2_error_list", "R_angle_error_mean",
                              "R_angle_error_list", "t_angle_error_mean", "t_angle_error_list", "R_angle_error_layers_list",
                              "t_angle_error_layers_list", "t_l2_error_layers_list", "q_l2_error_layers_list"}
    np.testing.assert_allclose(torch.stack(rt["q_l2_error_layers_list"]).cpu().numpy(), g[pre + "q_l2_layers"], atol=5e-6, rtol=2e-4)
    np.testing.assert_allclose(torch.stack(rt["t_l2_error_layers_list"]).cpu().numpy(), g[pre + "t_l2_layers"], atol=5e-5, rtol=2e-4)
    np.testing.assert_allclose(np.stack(rt["t_angle_error_layers_list"]), g[pre + "t_angle_layers"], atol=2e-2, rtol=1e-4)
    np.testing.assert_allclose(np.stack(rt["R_angle_error_layers_list"]), g[pre + "stubcv2_R_angle_layers"], atol=5e-3, rtol=1e-3)
    np.testing.assert_allclose(rt["t_l2_error_mean"].item(), g[pre + "t_l2_error_mean"], rtol=2e-4)
    np.testing.assert_allclose(rt["q_l2_error_mean"].item(), g[pre + "q_l2_error_mean"], rtol=2e-4)
    np.testing.assert_allclose(rt["q_l2_error_list"].cpu().numpy(), g[pre + "q_l2_error_list"], rtol=2e-4)  # the reference's slip: holds the t means
    np.testing.assert_allclose(rt["t_l2_error_list"].cpu().numpy(), g[pre + "t_l2_error_list"], rtol=2e-4)


def test_dsac_tools_functions_match_reference_golden(dfepe, oracle, golden):
    g = golden("geometry")
    uF, uG = dfepe.compat.utils_F, dfepe.compat.utils_geo
    x1, x2, F, K = T(g["x1"]).to(DEV), T(g["x2"]).to(DEV), T(g["F_in"]).to(DEV), T(g["K"]).to(DEV)
    # golden is the reference in fp64 on fp64 pixels; here the pixels (~1e3) are fp32, so y^T F x carries ~1e-4 px of input rounding
    np.testing.assert_allclose(uF._sym_epi_dist(F, x1, x2).cpu().numpy(), g["sym_epi_b"], rtol=2e-3, atol=3e-4)
    np.testing.assert_allclose(uF._sym_epi_dist(F[0], x1[0], x2[0]).cpu().numpy(), g["sym_epi_2d"], rtol=2e-3, atol=3e-4)
    np.testing.assert_allclose(uF._sampson_dist(F, x1, x2).cpu().numpy(), g["sampson_b"], rtol=2e-3, atol=3e-4)
    np.testing.assert_allclose(torch.stack(uF._epi_distance(F, x1, x2)).cpu().numpy(), g["epi_dist_b"], rtol=2e-3, atol=3e-3)
    a, r, _ = unit_align(uF._F_to_E(F[0], K).cpu().numpy()[None], g["F_to_E"][None])
    assert np.abs(a - r).max() < 1e-4
    np.testing.assert_allclose(uF._E_to_F(T(g["E_in"]).to(DEV), K.expand(8, 3, 3)).cpu().numpy().shape, (8, 3, 3))
    for b in range(8):
        R2s, t2s, M2s = uF._get_M2s(T(g["E_in"][b]).to(DEV))
        assert len(R2s) == 2 and len(t2s) == 2 and len(M2s) == 4 and M2s[0].shape == (3, 4)
        d11 = np.abs(R2s[0].cpu().numpy() - g["M2s_R1"][b]).max() + np.abs(R2s[1].cpu().numpy() - g["M2s_R2"][b]).max()
        d12 = np.abs(R2s[0].cpu().numpy() - g["M2s_R2"][b]).max() + np.abs(R2s[1].cpu().numpy() - g["M2s_R1"][b]).max()
        assert min(d11, d12) < 2e-5
        assert min(np.abs(t2s[0].cpu().numpy() - g["M2s_t"][b]).max(), np.abs(t2s[1].cpu().numpy() - g["M2s_t"][b]).max()) < 2e-5
    q = uG._R_to_q(T(g["Rq_in"]).to(DEV))
    np.testing.assert_allclose(q.cpu().numpy(), g["Rq_q"], atol=2e-6)  # all four trace-method branches
    assert uG._R_to_q(T(g["Rq_in"][0]).to(DEV)).shape == (4, 1)
    for b in range(8):
        t_cam = np.linalg.inv(g["delta_Rtijs_4_4"][b])[:3, 3]
        assert abs(uG.vector_angle(g["M2s_t"][b], t_cam) - g["vec_angle"][b]) < 2e-2
    # textbook 8-point variants (same kernel, flags): fp32 inputs vs the reference's fp64 run
    Kd = K.double()
    for ours, key in ((uF._F_from_XY(x1[0], x2[0]), "F_from_XY"), (uF._E_from_XY(x1[0], x2[0], K), "E_from_XY"),
                      (uF._E_from_XY(x1[0], x2[0], K, W=torch.diag(T(g["W_diag"]).to(DEV))), "E_from_XY_W")):
        a, r, _ = unit_align(ours.cpu().numpy()[None], g[key][None])
        assert np.abs(a - r).max() < 5e-4, (key, np.abs(a - r).max())
    # normalize=False (no Hartley step) against the fp64 oracle on K^-1-normalised points, where it is well conditioned
    x1n = (torch.cat((x1[0], torch.ones(x1.shape[1], 1, device=DEV)), 1) @ torch.linalg.inv(K).t())[:, :2].contiguous()
    x2n = (torch.cat((x2[0], torch.ones(x2.shape[1], 1, device=DEV)), 1) @ torch.linalg.inv(K).t())[:, :2].contiguous()
    for ess in (False, True):
        ours = (uF._E_from_XY(x1n, x2n, K, if_normzliedK=True, normalize=False) if ess else uF._F_from_XY(x1n, x2n, normalize=False))
        ref = (oracle.E_from_XY(x1n.cpu().double(), x2n.cpu().double(), Kd.cpu(), if_normzliedK=True, normalize=False) if ess
               else oracle.F_from_XY(x1n.cpu().double(), x2n.cpu().double(), normalize=False))
        a, r, _ = unit_align(ours.cpu().numpy()[None], ref.numpy()[None])
        assert np.abs(a - r).max() < 2e-5, (ess, np.abs(a - r).max())
    # batched forms (the reference needs the external batch_svd extension for these)
    Eb = uF._E_from_XY_batch(x1, x2, K.expand(8, 3, 3))
    a, r, sgn = unit_align(Eb[:1].cpu().numpy(), g["E_from_XY"][None])
    assert np.abs(a - r).max() < 5e-4
    E1 = uF._E_from_XY(x1[0], x2[0], K)
    assert (Eb[0] + E1).abs().max().item() < 1e-6  # the batched reference function returns the negated matrix (utils_F.py:221)
    Fb, Ab = T(g["E_in"]).to(DEV), (K.expand(8, 3, 3) + torch.arange(8, device=DEV).float().reshape(8, 1, 1) * 1e-3).contiguous()
    np.testing.assert_allclose(dfepe.ops.congruence(Fb, Ab).cpu().numpy(),
                               (Ab.double().transpose(1, 2) @ Fb.double() @ Ab.double()).cpu().numpy(), rtol=1e-6, atol=1e-6 * float(Ab.abs().max()) ** 2)
    Rs, ts = uF._get_M2s_batch(T(g["E_in"]).to(DEV))
    assert Rs[0].shape == (8, 3, 3) and ts[0].shape == (8, 3, 1) and (ts[0] + ts[1]).abs().max().item() == 0.0
    d3, d1, d2 = uF.epi_distance_np(g["F_in"][0], g["x1"][0], g["x2"][0])
    np.testing.assert_allclose(d3, 2 * g["epi_dist_b"][0][0], rtol=2e-3, atol=6e-3)


def test_compute_epi_residual_standalone(dfepe, oracle, golden):
    g = golden("fit")
    p1, p2, F = T(g["general_f32_pts1"]).to(DEV), T(g["general_f32_pts2"]).to(DEV), T(g["general_f32_out"]).to(DEV)
    for c, key in ((0.5, "epi_0p5"), (0.02, "epi_0p02")):
        np.testing.assert_allclose(dfepe.compat.utils_F.compute_epi_residual(p1, p2, F, c).cpu().numpy(), g[f"general_f32_{key}"], atol=2e-6, rtol=1e-4)
    Fd = F.clone().requires_grad_(True)
    G = torch.randn(8, 100, generator=torch.Generator().manual_seed(0)).to(DEV)
    (dfepe.compat.utils_F.compute_epi_residual(p1, p2, Fd, 0.5) * G).sum().backward()
    Fo = F.cpu().double().requires_grad_(True)
    (oracle.compute_epi_residual(p1.cpu().double(), p2.cpu().double(), Fo, 0.5) * G.cpu().double()).sum().backward()
    assert np.abs(Fd.grad.cpu().numpy() - Fo.grad.numpy()).max() / np.abs(Fo.grad.numpy()).max() < 1e-4


@pytest.mark.parametrize("N", [64, 1000])
def test_cheirality_recovers_generating_pose(dfepe, oracle, N):
    """OpenCV's triangulation is unpinned (absent here): validate by geometric ground truth and against the oracle's DLT."""
    B = 12
    sc = dfepe.synth.make_scene(B, N, seed=17, noise_px=0.3, outlier_ratio=0.1)
    E = sc["E_gt"] / sc["E_gt"].flatten(1).norm(dim=1)[:, None, None]
    Rt, win, cnt = dfepe.ops.cheirality(E.to(DEV), sc["Ks"].to(DEV), sc["matches_xy_ori"].to(DEV), 50.0)
    Rt, win, cnt = Rt.cpu().numpy(), win.cpu().numpy(), cnt.cpu().numpy()
    cam = np.linalg.inv(sc["delta_Rtijs_4_4"].numpy())
    for b in range(B):
        assert win[b] >= 0 and cnt[b].max() == cnt[b, win[b]]
        assert oracle.rotation_angle_deg(Rt[b, :, :3], cam[b, :3, :3]) < 0.05
        assert oracle.vector_angle_deg(Rt[b, :, 3], cam[b, :3, 3]) < 0.5
    for b in range(3):  # counts against the CPU DLT (same algorithm, fp64 SVD of the 4x4)
        _, _, counts = oracle.cheirality_select(E[b].double(), sc["Ks"][b].numpy(), sc["matches_xy_ori"][b, :, :2].double().numpy(),
                                                sc["matches_xy_ori"][b, :, 2:].double().numpy(), 50.0)
        assert sorted(counts) == sorted(cnt[b].tolist()) or np.abs(np.sort(counts) - np.sort(cnt[b])).max() <= max(1, N // 200)
    # compat wrapper returns the reference's triple
    _, err, Rt_cam = dfepe.compat.utils_F._E_to_M_train(E[0].to(DEV), sc["Ks"][0].numpy(), sc["matches_xy_ori"][0, :, :2].numpy(),
                                                        sc["matches_xy_ori"][0, :, 2:].numpy(), delta_Rt_gt_cam=cam[0], show_result=False)
    assert Rt_cam.shape == (3, 4) and err[0] < 0.05 and err[1] < 0.5


@pytest.mark.parametrize("case,pad_to", [("mixed", 0), ("dense1000", 0), ("dense1000", 2100), ("garbage", 0)])
def test_cheirality_matches_the_references_own_logic(dfepe, golden, case, pad_to):
    """dfepe_cheirality against tests/golden/cheirality.npz -- the reference's own _E_to_M_train (utils_F.py:679-763:
    candidate order, 0 < Z < depth_thres in both cameras, first arg-max, _inv_Rt of the winner) run with a DLT stand-in for
    cv2.triangulatePoints.  The winner's count and Rt_cam must agree; a count may differ by the odd correspondence whose
    depth sits on a bound (the 4x4 eigen-solver here is not numpy's SVD; OpenCV's own triangulation is unpinned anyway).
    pad_to > 2048 embeds the fixture pairs in a large batch: that selects the one-wavefront-per-pair variant at N = 1000."""
    g = golden("cheirality")
    E, K, m = (torch.from_numpy(g[f"{case}_{k}"]).float() for k in ("E", "K", "matches"))
    thr = float(g[f"{case}_depth_thres"])
    Bf, N = m.shape[0], m.shape[1]
    if pad_to:
        sc = dfepe.synth.make_scene(pad_to - Bf, N, seed=9, outlier_ratio=0.2)
        Ef = sc["E_gt"] / sc["E_gt"].flatten(1).norm(dim=1)[:, None, None]
        E, K, m = torch.cat((E, Ef.float())), torch.cat((K, sc["Ks"].float())), torch.cat((m, sc["matches_xy_ori"].float()))
    Rt, win, cnt = dfepe.ops.cheirality(E.to(DEV), K.to(DEV), m.to(DEV), thr)
    Rt, win, cnt = Rt.cpu().numpy()[:Bf], win.cpu().numpy()[:Bf], cnt.cpu().numpy()[:Bf]
    gc, gw, gR = g[f"{case}_counts"], g[f"{case}_winner"], g[f"{case}_Rt_cam"]
    # The ORDER of the four candidates follows the SVD gauge (the signs LAPACK happens to give u3 and (u1, v1) decide which
    # candidate is "(R1, t)"; any valid SVD yields the same SET), so per-candidate counts are compared as a multiset and the
    # winner through what it selects: its count and its pose, both gauge-free.
    slack = max(1, N // 250)
    assert np.abs(np.sort(cnt, axis=1) - np.sort(gc, axis=1)).max() <= slack, (cnt, gc)
    decided = 0
    for b in range(Bf):
        top2 = np.sort(gc[b])[-2:]
        if gw[b] < 0:
            assert win[b] < 0
            continue
        assert win[b] >= 0 and abs(int(cnt[b, win[b]]) - int(gc[b, gw[b]])) <= slack
        if top2[1] - top2[0] > 2 * slack:  # a decided vote: the winner is not up to a boundary point
            decided += 1
            np.testing.assert_allclose(Rt[b], gR[b], atol=2e-5)
    assert decided >= Bf // 2


def test_cheirality_with_fused_E_from_F(dfepe):
    """pre = T K: the launch decomposes (T K)^T F (T K) -- same result as the stand-alone congruence followed by cheirality."""
    B, N = 9, 300
    sc = dfepe.synth.make_scene(B, N, seed=4, outlier_ratio=0.2)
    d = {k: v.to(DEV) for k, v in sc.items()}
    w = torch.softmax(d["logits_layers"][0], 1).contiguous()
    F = dfepe.ops.w8pt_forward(d["matches_xy_ori"], None, w, True, 1241.0, 376.0, 0.5, False, False)[0]
    T = torch.tensor([[2.0 / 1241, 0.0, -1.0], [0.0, 2.0 / 376, -1.0], [0.0, 0.0, 1.0]], device=DEV)
    TK = (T @ d["Ks"]).contiguous()
    a = dfepe.ops.cheirality(dfepe.ops.congruence(F, TK), d["Ks"], d["matches_xy_ori"], 50.0)
    b = dfepe.ops.cheirality(F, d["Ks"], d["matches_xy_ori"], 50.0, pre=TK)
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[0], b[0])


def test_validation_pose_path(dfepe, oracle):
    """goodCorr_eval_nondecompose / val_rt_batch (the cv2.recoverPose path of the reference, unpinned): recover the
    generating pose from the ground-truth E and from an E estimated by the solver on noisy matches."""
    B, N = 16, 200
    sc = dfepe.synth.make_scene(B, N, seed=23, noise_px=0.3)
    dev = dfepe.pipeline.scene_to_device(sc, DEV)
    out = dfepe.compat.train_good_utils.val_rt_batch(dev["Ks"], dev["matches_xy_ori"], dev["E_gt"], dev["delta_Rtijs_4_4"])
    assert out["err_R_deg"].max().item() < 0.05 and out["err_t_deg"].max().item() < 0.5
    assert (out["winner"] >= 0).all()
    # E from the solver (uniform weights) on the same matches
    w = torch.full((B, N), 1.0 / N, device=DEV)
    F, _, _ = dfepe.ops.w8pt_raw(dev["matches_xy_ori"], w, 1241, 376)
    T = oracle.hw_matrix(IMAGE_SIZE).to(DEV)
    E = dev["Ks"].transpose(1, 2) @ T.t() @ F @ T @ dev["Ks"]
    out2 = dfepe.compat.train_good_utils.val_rt_batch(dev["Ks"], dev["matches_xy_ori"], E, dev["delta_Rtijs_4_4"])
    assert out2["err_R_deg"].median().item() < 0.5 and out2["err_t_deg"].median().item() < 10.0
    # single-pair reference signature
    cam = np.linalg.inv(sc["delta_Rtijs_4_4"][0].numpy())[:3]
    M, (eq, et) = dfepe.compat.utils_F.goodCorr_eval_nondecompose(sc["matches_xy_ori"][0, :, :2].numpy(), sc["matches_xy_ori"][0, :, 2:].numpy(),
                                                                sc["E_gt"][0].numpy().astype(np.float64), cam, sc["Ks"][0].numpy(), None)
    assert M.shape == (3, 4) and eq < 0.05 and et < 0.5
    np.testing.assert_allclose(M[:, :3], sc["delta_Rtijs_4_4"][0, :3, :3].numpy(), atol=2e-3)  # scene convention: x2 ~ R x1 + t
    M2, errs = dfepe.compat.utils_F.goodCorr_eval_nondecompose(np.zeros((3, 2)), np.zeros((3, 2)), np.eye(3), cam, sc["Ks"][0].numpy(), None)
    assert errs == (180.0, 90.0)


def test_deepfnet_learned_offsets_branch(dfepe):
    """if_learn_offsets (DeepFNet.py:490-507): with a zeroed offset head the model equals the plain one; the gradient of
    the F-loss reaches the offset head through the solver's d/d(matches) and agrees with a finite difference."""
    torch.manual_seed(0)
    B, N, depth = 3, 100, 3
    sc = dfepe.synth.make_scene(B, N, seed=51, outlier_ratio=0.2)
    dev = dfepe.pipeline.scene_to_device(sc, DEV)
    batch = {"matches_xy_ori": dev["matches_xy_ori"], "matches_good_unique_nums": None, "t_scene_scale": None}
    plain = dfepe.compat.DeepFNet.DeepFNet(depth=depth, image_size=IMAGE_SIZE, if_quality=False).to(DEV)
    dfepe.synth.fill_params_deterministic(plain, seed=2)
    net = dfepe.compat.DeepFNet.DeepFNet(depth=depth, image_size=IMAGE_SIZE, if_quality=False, if_learn_offsets=True).to(DEV)
    net.input_weights.load_state_dict(plain.input_weights.state_dict())
    net.update_weights.load_state_dict(plain.update_weights.state_dict())
    with torch.no_grad():
        for p in net.update_offsets.parameters():
            p.zero_()
    o0, o1 = plain(batch), net(batch)
    assert "offsets" in o1 and o1["offsets"].shape == (B, 4, N) and o1["offsets"].abs().max().item() == 0.0
    for a, b in zip(o0["out_layers"], o1["out_layers"]):  # the stock estimator is not bit-reproducible run to run (MIOpen): ~1e-6 drift
        assert (a - b).abs().max().item() < 1e-4 * a.abs().max().item()
    # now a small non-zero head: analytic gradient vs central finite difference along the last bias
    dfepe.synth.fill_params_deterministic(net.update_offsets, seed=9)
    with torch.no_grad():
        for p in net.update_offsets.parameters():
            p.mul_(0.05)
    lp = {"depth": depth, "clamp_at": 0.5, "if_tri_depth": False, "if_sample_loss": False, "topK": 8, "matches_good_unique_nums": None}

    def loss_of():
        outs = net(batch)
        losses = dfepe.compat.train_good_utils.get_all_loss_DeepF(outs, dev["pts1_virt_ori"], dev["pts2_virt_ori"], dev["Ks"], lp, get_residual_summaries=False)[0]
        return losses["loss_F"]

    bias = net.update_offsets.fw[15].bias
    net.zero_grad()
    loss_of().backward()
    g = bias.grad.clone()
    assert torch.isfinite(g).all() and g.abs().max().item() > 0
    d = torch.tensor([1.0, -0.5, 0.7, 0.3], device=DEV)
    eps = 0.05  # pixels
    with torch.no_grad():
        bias.add_(eps * d); lp_ = loss_of().item(); bias.sub_(2 * eps * d); lm_ = loss_of().item(); bias.add_(eps * d)
    num = (lp_ - lm_) / (2 * eps)
    ana = (g * d).sum().item()
    assert abs(num - ana) < 0.05 * max(abs(num), abs(ana)) + 1e-7, (num, ana)


def test_cheirality_batch_size_variants_agree(dfepe):
    """B >= 2048 runs one wavefront per pair, smaller batches four (they split the correspondences): same results."""
    B, N = 2048, 150
    sc = dfepe.synth.make_scene(B, N, seed=21, outlier_ratio=0.3, noise_px=1.0)
    E, K, m = sc["E_gt"].to(DEV), sc["Ks"].to(DEV), sc["matches_xy_ori"].to(DEV)
    Rt, win, cnt = dfepe.ops.cheirality(E, K, m, 50.0)
    Rt2, win2, cnt2 = dfepe.ops.cheirality(E[:200].contiguous(), K[:200].contiguous(), m[:200].contiguous(), 50.0)
    assert torch.equal(win[:200], win2) and torch.equal(cnt[:200], cnt2) and torch.equal(Rt[:200], Rt2)
    assert (win >= 0).float().mean().item() > 0.95


def test_dsac_hypothesis_loop(dfepe, oracle):
    """compat.dsac.DSAC (all hypotheses per launch) against the oracle's per-hypothesis loop on the same minimal sets
    (Python's `random` seeded identically): per-correspondence average scores, per-hypothesis scores, and the refined E
    of every hypothesis up to sign."""
    import random

    sc = dfepe.synth.make_scene(1, 200, seed=8, outlier_ratio=0.3, noise_px=0.5)
    X, Y, K = sc["matches_xy_ori"][0, :, :2], sc["matches_xy_ori"][0, :, 2:], sc["Ks"][0]
    hyps, thr, beta = 24, 2.0, 5.0
    losses = []
    d = dfepe.compat.dsac.DSAC(hyps, thr, beta, 0.5, K, lambda H, Xa, Ya: losses.append(1) or H.abs().sum())
    random.seed(5)
    out = d(X.to(DEV), Y.to(DEV), None)
    random.seed(5)
    idx_list = [random.sample(range(200), 10) for _ in range(hyps)]
    ref, ref_scores, ref_E = oracle.dsac_scores(X.double(), Y.double(), K.double(), hyps, thr, beta, idx_list)
    assert out.shape == (200, 1) and len(losses) == hyps and d.hyp_losses.shape == (hyps,)
    np.testing.assert_allclose(d.hyp_scores.cpu().numpy(), ref_scores.numpy(), rtol=2e-3, atol=1e-3)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=2e-3, atol=1e-3)
    assert d.best_H_idx == int(ref_scores.argmax()) and d.best_corres_idx == idx_list[d.best_H_idx]
    a, r, _ = unit_align(d.best_H.reshape(1, 3, 3).cpu().numpy(), ref_E[d.best_H_idx].reshape(1, 3, 3).numpy())
    assert np.linalg.norm(a - r, axis=1).max() < 1e-3


class _Recorder:
    def __init__(self):
        self.scalars, self.hists = {}, {}

    def add_scalar(self, tag, value, n_iter):
        self.scalars[tag] = float(value)

    def add_histogram(self, tag, values, n_iter):
        self.hists[tag] = np.asarray(values)


def test_write_metrics_summary_matches_reference_golden(dfepe, golden):
    """compat.train_good_utils.write_metrics_summary (counts, F1, medians, maxima, histogram ratios reduced on the device)
    against every scalar the reference's own write_metrics_summary logged for the same inputs (tests/golden/metrics.npz)."""
    g = golden("metrics")
    want = dict(zip([str(t) for t in g["tags"]], g["values"]))
    d = {}
    for k in g.files:
        if k.startswith("in_"):
            metric, tag = k[3:].rsplit("_", 1)
            # half as numpy arrays (the reference's layout), half as device tensors
            d.setdefault(metric, {})[tag] = [a if i % 2 else torch.from_numpy(a).to(DEV) for i, a in enumerate(g[k])]
    rec = _Recorder()
    dfepe.compat.train_good_utils.write_metrics_summary(rec, d, "val", 7)
    assert sorted(rec.scalars) == sorted(want)
    for tag, v in want.items():
        assert abs(rec.scalars[tag] - v) < 2e-6 * max(1.0, abs(v)), (tag, rec.scalars[tag], v)
    assert len(rec.hists) == 9


def test_validation_summary_end_to_end(dfepe, oracle):
    """val_rt_batch + epipolar distances + the summary reductions in one call, against the oracle's numpy summary of the
    same per-pair tensors."""
    B, N = 64, 200
    sc = dfepe.synth.make_scene(B, N, seed=3, outlier_ratio=0.2, noise_px=0.5)
    E = sc["E_gt"] + 2e-3 * torch.randn(B, 3, 3, generator=torch.Generator().manual_seed(1)) * sc["E_gt"].abs().max()
    F_est = sc["F_gt"] + 2e-5 * torch.randn(B, 3, 3, generator=torch.Generator().manual_seed(2)) * sc["F_gt"].abs().amax(dim=(1, 2), keepdim=True)
    sm, pairs = dfepe.compat.train_good_utils.validation_summary(sc["Ks"].to(DEV), sc["matches_xy_ori"].to(DEV), E.to(DEV), F_est.to(DEV),
                                                                 sc["F_gt"].to(DEV), sc["delta_Rtijs_4_4"].to(DEV))
    ref = oracle.metrics_summary_np(pairs["epi_dists"].cpu().numpy(), pairs["epi_dists_gt"].cpu().numpy(), pairs["err_R_deg"].cpu().numpy(),
                                    pairs["err_t_deg"].cpu().numpy())
    for k, v in ref.items():
        np.testing.assert_allclose(np.asarray(sm[k]), np.asarray(v), rtol=2e-6, atol=1e-7, err_msg=k)
    assert 0.0 <= sm["ratio_0.1"] <= sm["ratio_1"] <= 1.0 and np.isfinite(sm["median_err_q"]) and sm["ratio_q"][-1] <= 1.0


def test_metrics_summary_propagates_nan_and_bins_like_numpy(dfepe):
    """A diverged pair must stay visible: np.median / np.amax of a vector holding a NaN are NaN (train_good_utils.py:799-816);
    np.histogram compares float64 values with float64 edges, so a float32 error just below a float-rounded edge stays below it."""
    q = torch.tensor([0.5, 2.0, float("nan"), 1.0, 7.0], device=DEV)
    t = torch.tensor([0.5, 2.0, 3.0, 1.0, 7.0], device=DEV)
    epi = torch.rand(5, 10, device=DEV)
    s = dfepe.ops.metrics_summary(epi, None, q, t)
    assert np.isnan(s["median_err_q"]) and np.isnan(s["max_err_q"])
    assert s["median_err_t"] == 2.0 and s["max_err_t"] == 7.0
    # float32(0.01) = 0.0099999998 < 0.01 (double); float32(0.3) = 0.30000001 > 0.3; float32(0.1) = 0.100000001 > 0.1
    edge = torch.tensor([0.01, 0.3, 0.1, 0.03], dtype=torch.float32, device=DEV)
    s = dfepe.ops.metrics_summary(epi, None, edge, edge)
    e64 = edge.cpu().numpy().astype(np.float64)
    hist, _ = np.histogram(e64, bins=np.array(dfepe.ops.METRIC_THS))
    np.testing.assert_allclose(s["ratio_q"], np.cumsum(hist) / 4.0)


def _decisive(counts):
    c = np.sort(np.asarray(counts))[::-1]
    return c[0] > 0 and c[0] > c[1]


def test_val_rt_matches_the_references_own_code(dfepe, golden):
    """compat.train_good_utils.val_rt / val_rt_batch / validation_summary and compat.utils_F.goodCorr_eval_nondecompose against
    tests/golden/valrt.npz = the reference's own val_rt (train_good_utils.py:553-646) and goodCorr_eval_nondecompose
    (utils_F.py:909-954) run with a stand-in for cv2.recoverPose: score mask, < 5 points fall-back, invert_Rt, angles,
    epi_distance_np.  Pairs whose in-front counts tie at the top are decided by the SVD sign gauge and are skipped for the pose."""
    g = golden("valrt")
    tgu, uF = dfepe.compat.train_good_utils, dfepe.compat.utils_F
    B = g["valrt_K"].shape[0]
    dec = [b for b in range(B) if _decisive(g["valrt_counts_est"][b])]
    assert len(dec) >= 8
    for b in range(B):
        m = g["valrt_matches"][b]
        r = tgu.val_rt(b, g["valrt_K"][b], m[:, :2], m[:, 2:], g["valrt_E_est"][b], g["valrt_E_gt"][b], g["valrt_F_est"][b], g["valrt_F_gt"][b],
                       g["valrt_delta"][b], five_point=False, if_opencv=False)
        assert r[6] == b and r[2] is None and r[3] is None and r[8] is None
        # float32 evaluation of |y^T F x| with pixel coordinates (the reference's is float32 too): relative to the distance scale
        np.testing.assert_allclose(r[1], g["valrt_epi_est"][b], rtol=2e-3, atol=2e-3 * np.abs(g["valrt_epi_est"][b]).max())
        np.testing.assert_allclose(r[5], g["valrt_epi_gt"][b], rtol=2e-3, atol=2e-3 * max(np.abs(g["valrt_epi_gt"][b]).max(), 1.0))
        if b in dec:
            np.testing.assert_allclose(r[0], g["valrt_err_est"][b], atol=0.05, rtol=1e-3)
            np.testing.assert_allclose(r[7], g["valrt_M_est"][b], atol=2e-4)
        if _decisive(g["valrt_counts_gt"][b]):
            np.testing.assert_allclose(r[4], g["valrt_err_gt"][b], atol=0.05, rtol=1e-3)
    # the batched forms on the same pairs
    t = lambda k: torch.from_numpy(g[k]).to(DEV)
    pairs = tgu.val_rt_batch(t("valrt_K"), t("valrt_matches"), t("valrt_E_est"), t("valrt_delta"))
    np.testing.assert_allclose(pairs["err_R_deg"].cpu().numpy()[dec], g["valrt_err_est"][dec, 0], atol=0.05, rtol=1e-3)
    np.testing.assert_allclose(pairs["err_t_deg"].cpu().numpy()[dec], g["valrt_err_est"][dec, 1], atol=0.05, rtol=1e-3)
    sm, per = tgu.validation_summary(t("valrt_K"), t("valrt_matches"), t("valrt_E_est"), t("valrt_F_est"), t("valrt_F_gt"), t("valrt_delta"))
    np.testing.assert_allclose(per["epi_dists"].cpu().numpy(), g["valrt_epi_est"], rtol=2e-3, atol=2e-3 * np.abs(g["valrt_epi_est"]).max())
    ref_ratio = float((g["valrt_epi_est"] < 1.0).mean())
    assert abs(sm["ratio_1"] - ref_ratio) <= 2.0 / g["valrt_epi_est"].size
    # goodCorr_eval_nondecompose with scores: only the top decile (ties kept by `>=`) decides the pose
    for b in range(g["scores_K"].shape[0]):
        if not _decisive(g["scores_counts"][b]):
            continue
        m = g["scores_matches"][b]
        dinv = np.linalg.inv(g["scores_delta"][b])[:3]
        M, err = uF.goodCorr_eval_nondecompose(m[:, :2], m[:, 2:], g["scores_E"][b].astype(np.float64), dinv, g["scores_K"][b], g["scores_scores"][b])
        np.testing.assert_allclose(err, g["scores_err"][b], atol=0.05, rtol=1e-3)
        np.testing.assert_allclose(M, g["scores_M"][b], atol=2e-4)
    bb = int(g["few_pair"])
    m = g["scores_matches"][bb]
    dinv = np.linalg.inv(g["scores_delta"][bb])[:3]
    for j, n in enumerate(g["few_n"]):
        M, err = uF.goodCorr_eval_nondecompose(m[40:40 + n, :2], m[40:40 + n, 2:], g["scores_E"][bb].astype(np.float64), dinv, g["scores_K"][bb], None)
        np.testing.assert_allclose(err, g["few_err"][j], atol=0.05, rtol=1e-3)
        np.testing.assert_allclose(M, g["few_M"][j], atol=2e-4)


def test_dense_W_and_unnormalised_rows_match_the_reference(dfepe, oracle, golden):
    """The last corners of the mirrored surface against the reference's own outputs (tests/golden/surface.npz):
    _E_from_XY / _F_from_XY with a dense W [N,N] (utils_F.py:129-130,245-246; the explicit-rows kernel) and
    Fit(normalize_SVD=False) (DeepFNet.py:211-214) forward + d/d(weights)."""
    g = golden("surface")
    uF = dfepe.compat.utils_F
    for b in range(g["densew_W"].shape[0]):
        m = T(g["densew_matches"][b]).float().to(DEV)
        K, W = T(g["densew_K"][b]).float().to(DEV), T(g["densew_W"][b]).float().to(DEV)
        for ours, key in ((uF._E_from_XY(m[:, :2], m[:, 2:], K, W=W), "densew_E"), (uF._F_from_XY(m[:, :2], m[:, 2:], W=W), "densew_F")):
            a, r, _ = unit_align(ours.cpu().numpy()[None], g[key][b][None])
            assert np.abs(a - r).max() < 5e-4, (key, b, np.abs(a - r).max())  # fp32 inputs and _normalize_XY vs the fp64 reference run
    # Fit(normalize_SVD=False): module forward and autograd to the weights
    H, W_ = IMAGE_SIZE[0], IMAGE_SIZE[1]
    m = T(g["nosvdnorm_matches"]).float().to(DEV)
    norm = dfepe.compat.DeepFNet.NormalizeAndExpand_HW(IMAGE_SIZE)
    p1, p2, _, _ = norm(m)
    pts1, pts2 = p1.permute(0, 2, 1).contiguous(), p2.permute(0, 2, 1).contiguous()
    w = T(g["nosvdnorm_weights"]).float().to(DEV).requires_grad_(True)
    fit = dfepe.compat.DeepFNet.Fit(normalize_SVD=False)
    out, res = fit(pts1, pts2, w)
    GF, GR = T(g["nosvdnorm_GF"]).float().to(DEV), T(g["nosvdnorm_GR"]).float().to(DEV)
    s = torch.sign((out.detach() * GF).flatten(1).sum(1))
    ((s[:, None, None] * out * GF).sum() + (s[:, None] * res * GR).sum()).backward()
    a, r, sg = unit_align(out.detach().cpu().numpy(), g["nosvdnorm_out_f64"])
    assert np.abs(a - r).max() < 2e-5
    # residual = X f / |f| with un-normalised rows: sign per pair from the F alignment, scale is absolute
    r64 = g["nosvdnorm_residual_f64"]
    sgn = np.sign((out.detach().cpu().numpy().reshape(-1, 9) * g["nosvdnorm_out_f64"].reshape(-1, 9)).sum(1))[:, None]
    assert np.abs(res.detach().cpu().numpy() * sgn - r64).max() < 2e-5 * np.abs(r64).max() + 1e-7
    gw, gw64 = w.grad.cpu().numpy(), g["nosvdnorm_grad_w_f64"]
    assert np.abs(gw - gw64).max() < 2e-3 * np.abs(gw64).max()
    # and the rows really are un-normalised: the default module gives a different F
    out_n, _ = dfepe.compat.DeepFNet.Fit()(pts1, pts2, w.detach())
    a2, r2, _ = unit_align(out_n.cpu().numpy(), g["nosvdnorm_out_f64"])
    assert np.abs(a2 - r2).max() > 1e-4


@pytest.mark.parametrize("B,N", [(24, 1000), (7, 300), (5, 100), (3, 2500)])
def test_fused_fit_and_pose_equals_the_two_launches(dfepe, B, N):
    """ops.fit_pose (dfepe_w8pt_pose_fwd: the cooperative fit's workgroup goes on to the cheirality check of its own pair) against
    w8pt_forward + cheirality(pre=T K): the same F, residuals, counts, winner and camera motion, bit for bit -- also for shapes that
    fall back to the two launches (N <= 128, N > 2048)."""
    sc = dfepe.pipeline.scene_to_device(dfepe.synth.make_scene(B, N, seed=61, outlier_ratio=0.25, noise_px=0.5), DEV)
    m = sc["matches_xy_ori"]
    w = torch.softmax(sc["logits_layers"][0], dim=1).contiguous()
    H, W = float(IMAGE_SIZE[0]), float(IMAGE_SIZE[1])
    T = torch.tensor([[2.0 / W, 0.0, -1.0], [0.0, 2.0 / H, -1.0], [0.0, 0.0, 1.0]], device=DEV)
    TK = (T @ sc["Ks"]).contiguous()
    F0, r0, e0, _, _ = dfepe.ops.w8pt_forward(m, None, w, True, W, H, 0.5, True, False)
    Rt0, win0, cnt0 = dfepe.ops.cheirality(F0, sc["Ks"], m, 50.0, pre=TK)
    F1, r1, e1, _, Rt1, win1, cnt1 = dfepe.ops.fit_pose(m, w, sc["Ks"], W, H, 50.0, pre=TK)
    for a, b in ((F0, F1), (r0, r1), (e0, e1), (Rt0, Rt1), (win0, win1), (cnt0, cnt1)):
        assert torch.equal(a, b)
    assert (win1 >= 0).float().mean().item() > 0.8
