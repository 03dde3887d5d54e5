"""Forward-fit time against wavefronts per pair (diagnostic flag bits 25-26) for large N."""
import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
for N in (384, 512, 768, 1000, 2000):
    for B in (4096, 1536, 512):
        sc = d.synth.make_scene(B, N, seed=1, outlier_ratio=0.2)
        m = sc["matches_xy_ori"].cuda(); w = torch.softmax(sc["logits_layers"][0], 1).cuda()
        r = [t(lambda: d.ops.w8pt_forward(m, None, w, True, 1241., 376., 0.5, True, True, diag=g)) for g in (0x200, 0x400, 0x600, 0)]
        print(f"N={N:5d} B={B:5d}: wpp1 {r[0]:7.1f}  wpp2 {r[1]:7.1f}  wpp4 {r[2]:7.1f}  auto {r[3]:7.1f} us", flush=True)
