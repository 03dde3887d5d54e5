"""Launch the kernels of the other BASELINE configurations and of the weight estimator a few times (for rocprofv3 kernel-trace and
--pmc passes): config 5 (N = 1000: fit + cheirality) at 4096 and at 512 pairs, one split-bf16 estimator call forward + backward at
4096 x 100 points (fused epilogue) and one at 12 x 2000 (plain product + est_norm_fwd_n / est_in_bwd_n)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what in ("all", "c5"):
    for B in (4096, 512):
        sc = d.pipeline.scene_to_device(d.synth.make_scene(B, 1000, seed=1, outlier_ratio=0.2), "cuda:0")
        w = torch.softmax(sc["logits_layers"][0], dim=1).contiguous()
        T = torch.tensor([[2.0 / 1241, 0, -1.0], [0, 2.0 / 376, -1.0], [0, 0, 1.0]], device="cuda:0")
        TK = (T @ sc["Ks"]).contiguous()
        for _ in range(4):
            F, _, _, _, _ = d.ops.w8pt_forward(sc["matches_xy_ori"], None, w, True, 1241.0, 376.0, 0.5, True, False)
            d.ops.cheirality(F, sc["Ks"], sc["matches_xy_ori"], 50.0, pre=TK)
        torch.cuda.synchronize()
if what in ("all", "est"):
    m = d.compat.ErrorEstimators.FusedErrorEstimator(7).cuda(); d.synth.fill_params_deterministic(m, 1)
    x = torch.rand(4096, 7, 100, device="cuda", requires_grad=True)
    G = torch.randn(4096, 1, 100, device="cuda")
    for _ in range(2):
        m.zero_grad(set_to_none=True); x.grad = None
        (m(x) * G).sum().backward()
    torch.cuda.synchronize()
if what in ("all", "estn"):
    m = d.compat.ErrorEstimators.FusedErrorEstimator(7).cuda(); d.synth.fill_params_deterministic(m, 1)
    x = torch.rand(12, 7, 2000, device="cuda", requires_grad=True)
    G = torch.randn(12, 1, 2000, device="cuda")
    for _ in range(2):
        m.zero_grad(set_to_none=True); x.grad = None
        (m(x) * G).sum().backward()
    torch.cuda.synchronize()
