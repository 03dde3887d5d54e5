"""Scratch timing of the forward kernel (HIP events) — superseded by bench.py."""
import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
for B, N in ((4096, 100), (4096, 1000), (512, 1000), (64, 1000), (4096, 512)):
    sc = d.synth.make_scene(B, N, seed=1, outlier_ratio=0.2)
    m = sc["matches_xy_ori"].cuda(); w = torch.softmax(sc["logits_layers"][0], 1).cuda()
    full = t(lambda: d.ops.w8pt_forward(m, None, w, True, 1241., 376., 0.5, True, True))
    nojac = t(lambda: d.ops.w8pt_forward(m, None, w, True, 1241., 376., 0.5, True, True, diag=1))
    if N == 100:
        for S in (1, 4, 5, 6, 8):
            ts = t(lambda: d.ops.w8pt_forward(m, None, w, True, 1241., 376., 0.5, True, True, diag=1 + S))
            tp = t(lambda: d.ops.w8pt_forward(m, None, w, True, 1241., 376., 0.5, True, True, diag=0x100 | (1 + S)))
            print(f"   forced {S} sweeps: no polish {ts:.1f} us, with polish {tp:.1f} us")
    if N >= 512:
        nc = t(lambda: d.ops.w8pt_forward(m, None, w, True, 1241., 376., 0.5, True, True, diag=0x200))
        print(f"   one wavefront per pair (cooperative workgroups off): {nc:.1f} us")
    byts = B * (28 * N + 36)
    print(f"B={B} N={N}: full {full:.1f} us, without Jacobi {nojac:.1f} us  ({B/full:.2f} Mpairs/s, {byts/full/1e3:.1f} GB/s algorithmic)")
