"""Forward fit at N = 1000 (pixel matches, weights in), cooperative workgroup per pair vs row-based kernels (DFEPE_W8PT_ROW_PER_PAIR: below 8192
pairs that is two rows of a wavefront per pair), hipGraph of 20 launches between HIP events."""
import ctypes, importlib, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
L = d._lib.lib()
N = 1000
for B in [int(a) for a in sys.argv[1:]] or [512, 1024, 2048, 3072, 4096, 8192]:
    sc = d.synth.make_scene(B, N, seed=1, outlier_ratio=0.2)
    m = sc["matches_xy_ori"].cuda().contiguous(); w = torch.softmax(sc["logits_layers"][0], 1).cuda().contiguous()
    Fo = torch.empty(B, 9, device="cuda"); res = torch.empty(B, N, device="cuda"); epi = torch.empty(B, N, device="cuda")
    out = []
    for flags in (1, 65):
        def launch():
            assert L.dfepe_w8pt_fwd(m.data_ptr(), None, w.data_ptr(), B, N, 1, flags, 1241.0, 376.0, 0.5, Fo.data_ptr(), res.data_ptr(), epi.data_ptr(), None, None,
                                    torch.cuda.current_stream().cuda_stream) == 0
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3): launch()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(20): launch()
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(7):
                e0.record(); g.replay(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 50.0)
        torch.cuda.current_stream().wait_stream(side)
        out.append(statistics.median(ts))
    print(f"B={B:5d}  default {out[0]:8.2f} us   row-based {out[1]:8.2f} us", flush=True)
