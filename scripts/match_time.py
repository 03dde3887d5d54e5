"""Timing of the match-construction row (nn_match_two_way + gather) against its MFMA roofline and the oracle on the host."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
oracle = importlib.import_module("oracle.deepf_oracle")
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n
for B, N, D in ((64, 1024, 256), (64, 1000, 256), (8, 1024, 256), (256, 1024, 256), (64, 2048, 256)):
    g = torch.Generator().manual_seed(0)
    d1 = torch.nn.functional.normalize(torch.randn(B, N, D, generator=g), dim=2)
    d2 = torch.nn.functional.normalize(d1[:, torch.randperm(N, generator=g)] + 0.05 * torch.randn(B, N, D, generator=g), dim=2)
    a, b = d1.cuda(), d2.cuda()
    dt = t(lambda: d.ops.nn_match_two_way(a, b, 0.7))
    flop = 2.0 * B * N * N * D
    line = f"B={B} N={N} D={D}: {dt*1e3:.3f} ms  {B/dt:.0f} pairs/s  {flop/dt/1e12:.1f} TFLOP/s ({100*flop/dt/157.3e12:.0f} % of the fp32 MFMA peak)"
    dt_ref = t(lambda: torch.sqrt((2 - 2 * torch.clamp(torch.bmm(a, b.transpose(1, 2)), -1, 1))).min(dim=2), n=5)
    line += f" | torch bmm+sqrt+min (row side only): {dt_ref*1e3:.3f} ms"
    if (B, N) == (64, 1024):
        t0 = time.perf_counter()
        for k in range(4):
            oracle.nn_match_two_way(d1[k].numpy().T, d2[k].numpy().T, 0.7)
        line += f" | oracle (numpy, host): {4/(time.perf_counter()-t0):.1f} pairs/s"
    print(line, flush=True)
