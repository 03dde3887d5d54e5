"""GPU time of ONE estimator call (forward + backward, captured in a hipGraph and replayed: no host in the figure) for a list of batch sizes, under
the plan switch of csrc/est_gemm.hip (DFEPE_EST_SPLITK = 0: fused epilogues wherever they exist, never split; 1: the default plan; 2: plain
products + register-resident normalisation everywhere).   DFEPE_EST_SPLITK=1 python scripts/est_plan_sweep.py N B1 B2 ..."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
N = int(sys.argv[1])
Bs = [int(a) for a in sys.argv[2:]]
DEV = "cuda:0"
est = d.compat.ErrorEstimators.FusedErrorEstimator(7).to(DEV)
d.synth.fill_params_deterministic(est, 1)
params = list(est.parameters())
out = []
for B in Bs:
    x = torch.rand(B, 7, N, device=DEV).requires_grad_(True)
    G = torch.randn(B, 1, N, device=DEV)

    def run():
        y = est(x)
        return torch.autograd.grad((y * G).sum(), [x] + params)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        run()
    for _ in range(5):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    out.append(f"B={B}: {best:.1f} us")
print(f"DFEPE_EST_SPLITK={os.environ.get('DFEPE_EST_SPLITK', '1')} N={N}: " + "  ".join(out))
