"""One-off randomized sweep of the single-fit adjoints (weights and points, with residual and epipolar upstream gradients)
against fp64 autograd of the oracle, over sizes incl. the cooperative-workgroup regime."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
oracle = importlib.import_module("oracle.deepf_oracle")
IMG = [376, 1241, 3]
def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
t0 = time.time()
for N, B in ((9, 16), (20, 16), (100, 16), (130, 8), (448, 4), (1000, 3)):
    for seed, (outl, noise, use_epi) in enumerate(((0.2, 0.5, True), (0.4, 1.0, False), (0.0, 0.3, True))):
        sc = d.synth.make_scene(B, N, seed=500 + 7 * N + seed, outlier_ratio=outl, noise_px=noise)
        g = torch.Generator().manual_seed(seed)
        GF, GR, GE = torch.randn(B, 3, 3, generator=g), torch.randn(B, N, generator=g), torch.randn(B, N, generator=g)
        w = torch.softmax(sc["logits_layers"][0], 1)
        m = sc["matches_xy_ori"].cuda().requires_grad_(True)
        aw = w.cuda().requires_grad_(True)
        outs = d.ops.w8pt_raw(m, aw, IMG[1], IMG[0], clamp_at=0.5, want_epi=True)
        loss = (outs[0] * GF.cuda()).sum() + (outs[1] * GR.cuda()).sum() + ((outs[2] * GE.cuda()).sum() if use_epi else 0.0)
        loss.backward()
        mo = sc["matches_xy_ori"].double().requires_grad_(True)
        ow = w.double().requires_grad_(True)
        q1, q2, _ = oracle.normalize_hw(mo, IMG)
        o_out, o_res, _ = oracle.fit_forward(q1, q2, ow.unsqueeze(1))
        s = torch.sign((o_out.detach() * outs[0].detach().cpu().double()).flatten(1).sum(1))
        lo = (s[:, None, None] * o_out * GF.double()).sum() + (s[:, None] * o_res * GR.double()).sum()
        if use_epi:
            lo = lo + (oracle.compute_epi_residual(q1, q2, o_out, 0.5) * GE.double()).sum()
        lo.backward()
        per_w = [rel(aw.grad[b].cpu().numpy(), ow.grad[b].numpy()) for b in range(B)]
        per_m = [rel(m.grad[b].cpu().numpy(), mo.grad[b].numpy()) for b in range(B)]
        print(f"N={N:5d} outl={outl} noise={noise} epi={use_epi}: finite {bool(torch.isfinite(m.grad).all() and torch.isfinite(aw.grad).all())}  d/dw rel err median {np.median(per_w):.1e} max {max(per_w):.1e}   d/dmatches median {np.median(per_m):.1e} max {max(per_m):.1e}", flush=True)
print(f"{time.time()-t0:.1f} s")
