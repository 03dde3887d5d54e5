"""Same-box A/B of the forward fit of two library builds at several batch sizes (training shape: pixel matches + logits in, epipolar
residual + save record + softmax weights out), hipGraph of 20 launches between HIP events, builds alternating:
    python scripts/ab_fit_sizes.py ab_libs/libdfepe_r4.so pytorch-deepfepe_amd/libdfepe_hip.so 4096 8192 32768
DFEPE_FIT_LEAN=0/1 in the environment forces the <= 256-register build of the NEW library off / on (default: from 8192 pairs)."""
import ctypes, importlib, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
P, I, U, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_float
libs = []
for path in sys.argv[1:3]:
    L = ctypes.CDLL(os.path.abspath(path))
    L.dfepe_w8pt_fwd.restype = I
    L.dfepe_w8pt_fwd.argtypes = [P, P, P, I, I, I, U, F, F, F, P, P, P, P, P, P]
    libs.append(L)
sizes = [int(a) for a in sys.argv[3:]] or [4096, 8192, 32768]
N = int(os.environ.get("AB_N", "100"))
print("DFEPE_FIT_LEAN =", os.environ.get("DFEPE_FIT_LEAN"), " N =", N)
for B in sizes:
    sc = d.synth.make_scene(B, N, seed=1, outlier_ratio=0.2)
    m = sc["matches_xy_ori"].cuda().contiguous(); lg = sc["logits_layers"][0].cuda().contiguous()
    outs = [(torch.empty(B, 9, device="cuda"), torch.empty(B, N, device="cuda"), torch.empty(B, N, device="cuda"), torch.empty(B, 128, device="cuda"),
             torch.empty(B, N, device="cuda")) for _ in libs]

    def launch(k):
        Fo, res, epi, sv, wo = outs[k]
        rc = libs[k].dfepe_w8pt_fwd(m.data_ptr(), None, lg.data_ptr(), B, N, 1, 3, 1241.0, 376.0, 0.5, Fo.data_ptr(), res.data_ptr(), epi.data_ptr(),
                                    sv.data_ptr(), wo.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0

    def timed(k, reps=20, rounds=7):
        for _ in range(3):
            launch(k)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                launch(k)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(rounds):
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / reps)
        return statistics.median(ts)

    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ts = [[], []]
        for rnd in range(4):
            for k in range(2):
                ts[k].append(timed(k))
        torch.cuda.synchronize()
    torch.cuda.current_stream().wait_stream(side)
    a, b = (statistics.median(t) for t in ts)
    same = all(torch.equal(torch.nan_to_num(x), torch.nan_to_num(y)) for x, y in zip(outs[0], outs[1]))
    print(f"B={B:6d}  A {a:8.2f} us  B {b:8.2f} us  B/A {b / a:.3f}   per 4096 pairs: A {a * 4096 / B:6.2f}  B {b * 4096 / B:6.2f} us   outputs bit-identical: {same}", flush=True)
