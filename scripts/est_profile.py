"""One split-bf16 estimator call (forward + backward) at B pairs, for rocprofv3 --kernel-trace --stats."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = d.compat.ErrorEstimators.FusedErrorEstimator(7).cuda(); d.synth.fill_params_deterministic(m, 1)
x = torch.rand(B, 7, 100, device="cuda", requires_grad=True)
G = torch.randn(B, 1, 100, device="cuda")
for _ in range(4):
    m.zero_grad(set_to_none=True); x.grad = None
    (m(x) * G).sum().backward()
torch.cuda.synchronize()
