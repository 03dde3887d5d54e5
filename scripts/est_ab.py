"""A/B of the estimator call (forward + backward, B x 100) between builds of the library / environment switches, alternating in ONE
process per variant is not possible (the library is loaded once), so: one process per variant, each timing 3 x 10 steps after warm-up
and printing the median block.   python scripts/est_ab.py [B]      (variants via DFEPE_LIB_PATH / DFEPE_EST_FUSE_DGRAD)"""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = d.compat.ErrorEstimators.FusedErrorEstimator(7).cuda(); d.synth.fill_params_deterministic(m, 1)
torch.manual_seed(0)
x = torch.rand(B, 7, 100, device="cuda")  # no input gradient: what DeepFNet's estimators see
G = torch.randn(B, 1, 100, device="cuda")
def step():
    m.zero_grad(set_to_none=True); (m(x) * G).sum().backward()
for _ in range(5): step()
blocks = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); blocks.append((time.perf_counter() - t0) / 10)
with torch.no_grad():
    for _ in range(3): m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m(x)
    torch.cuda.synchronize(); fwd = (time.perf_counter() - t0) / 10
print(f"lib={os.path.basename(d._lib.LIB_PATH)} fuse={os.environ.get('DFEPE_EST_FUSE_DGRAD','1')}: fwd+bwd {sorted(blocks)[1]*1e3:.3f} ms (blocks {[round(b*1e3,3) for b in blocks]}), no-grad fwd {fwd*1e3:.3f} ms")
