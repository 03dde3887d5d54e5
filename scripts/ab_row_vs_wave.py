"""Same-box A/B: row-per-pair kernels (w8pt16) against the wavefront-per-pair kernels (DFEPE_W8PT_WAVE_PER_PAIR) on the
bench workload (raw matches, fused softmax, epipolar residual, save record), forward and backward, HIP events."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
d = importlib.import_module("pytorch-deepfepe_amd")
DEV = "cuda:0"


def timeit(fn, iters=50, warm=5):
    # a hipGraph of `iters` back-to-back calls: the host launch path cannot be the bottleneck of a 10 us kernel
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(3):
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e3)
    return best  # us


def main():
    for B, N in [(4096, 100), (4096, 1000), (512, 1000), (1024, 300), (4096, 300), (64, 2000)]:
        sc = d.synth.make_scene(min(B, 4096), N, seed=1, outlier_ratio=0.2)
        rep = B // min(B, 4096)
        m = sc["matches_xy_ori"].repeat(rep, 1, 1).to(DEV).contiguous()
        lg = sc["logits_layers"][0].repeat(rep, 1).to(DEV).contiguous()
        gF = torch.randn(B, 3, 3, device=DEV)
        gR = torch.randn(B, N, device=DEV)
        gE = torch.randn(B, N, device=DEV)
        out = {}
        for name, wpp in (("row", False), ("wave", True)):
            F = torch.empty(B, 3, 3, device=DEV)
            fwd = lambda: d.ops.w8pt_forward(m, None, lg, True, 1241., 376., 0.5, True, True, logits=True, F_out=F, wave_per_pair=wpp)
            t_f = timeit(fwd)
            Fo, res, epi, save, w = fwd()
            gw = torch.empty(B, N, device=DEV)
            bwd_F = lambda: d.ops.w8pt_backward(m, None, w, True, 1241., 376., 0.5, save, Fo, gF, None, None, logits=True, out=gw, wave_per_pair=wpp)
            bwd_all = lambda: d.ops.w8pt_backward(m, None, w, True, 1241., 376., 0.5, save, Fo, gF, gR, gE, logits=True, out=gw, wave_per_pair=wpp)
            out[name] = (t_f, timeit(bwd_F), timeit(bwd_all), Fo.clone(), bwd_all().clone())
        r, w_ = out["row"], out["wave"]
        s = torch.sign((r[3] * w_[3]).flatten(1).sum(1))[:, None, None]
        dF = ((r[3] - s * w_[3]).flatten(1).norm(dim=1) / w_[3].flatten(1).norm(dim=1)).max().item()
        dg = ((r[4] - w_[4]).abs().max() / w_[4].abs().max()).item()
        print(f"B={B} N={N}: fwd row {r[0]:.1f} us  wave {w_[0]:.1f} us | bwd(gF) row {r[1]:.1f}  wave {w_[1]:.1f} | "
              f"bwd(gF,gRes,gEpi) row {r[2]:.1f}  wave {w_[2]:.1f} | max|dF| {dF:.1e}  grad rel diff {dg:.1e}", flush=True)


if __name__ == "__main__":
    main()
