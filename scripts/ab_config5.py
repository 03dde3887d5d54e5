"""Same-box A/B of BASELINE config 5's kernels in two builds of the library (ctypes on both, no package binding: the round-4 build
lacks the newer symbols):  python scripts/ab_config5.py ab_libs/libdfepe_r4.so pytorch-deepfepe_amd/libdfepe_hip.so [pairs ...]
For every batch size: the cheirality launch alone (E = the fit's F, pre = T K: the benchmark's scene), the fit + pose call
(dfepe_w8pt_pose_fwd), each as a hipGraph of 20 calls replayed between HIP events, alternating the builds; then the in-front counts of
the two builds compared.  DFEPE_POSE_LAUNCHES=2 in the environment makes the NEW build run fit and pose as two launches."""
import ctypes
import importlib
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
P, I, F, U = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_uint
libs = []
for path in sys.argv[1:3]:
    L = ctypes.CDLL(os.path.abspath(path))
    L.dfepe_cheirality.restype = I
    L.dfepe_cheirality.argtypes = [P, P, P, P, I, I, F, P, P, P, P]
    L.dfepe_w8pt_pose_fwd.restype = I
    new_abi = hasattr(L, "dfepe_cheirality_ex")
    L.dfepe_w8pt_pose_fwd.argtypes = [P, P, I, I, U, F, F, F, P, P, F, P, P, P, P, P, P, P] + ([P, P] if new_abi else [P])
    if new_abi:
        L.dfepe_cheirality_ex.restype = I
        L.dfepe_cheirality_ex.argtypes = [P, P, P, P, I, I, F, U, P, P, P, P, P]
        L.dfepe_cheirality_workspace_bytes.restype = ctypes.c_size_t
        L.dfepe_cheirality_workspace_bytes.argtypes = [I]
    L.new_abi = new_abi
    L.dfepe_w8pt_fwd.restype = I
    L.dfepe_w8pt_fwd.argtypes = [P, P, P, I, I, I, U, F, F, F, P, P, P, P, P, P]
    libs.append(L)
sizes = [int(a) for a in sys.argv[3:]] or [4096, 512]
N, W, H = 1000, 1241.0, 376.0
dev = "cuda:0"


def timed(fn, reps=20, rounds=7):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(rounds):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return statistics.median(ts)


for B in sizes:
    sc = d.pipeline.scene_to_device(d.synth.make_scene(B, N, seed=1000, outlier_ratio=0.2, noise_px=0.5), dev)
    m = sc["matches_xy_ori"].contiguous()
    w0 = torch.softmax(sc["logits_layers"][0], dim=1).contiguous()
    T = torch.tensor([[2.0 / W, 0.0, -1.0], [0.0, 2.0 / H, -1.0], [0.0, 0.0, 1.0]], device=dev)
    TK = (T @ sc["Ks"]).contiguous()
    K = sc["Ks"].contiguous()
    Fo = torch.empty(B, 9, device=dev); res = torch.empty(B, N, device=dev); epi = torch.empty(B, N, device=dev)
    outs = []
    for L in libs:
        outs.append((torch.empty(B, 12, device=dev), torch.empty(B, device=dev, dtype=torch.int32), torch.empty(B, 4, device=dev, dtype=torch.int32)))

    def st():
        return torch.cuda.current_stream().cuda_stream

    def fit(L):
        assert L.dfepe_w8pt_fwd(m.data_ptr(), None, w0.data_ptr(), B, N, 1, 1, W, H, 0.5, Fo.data_ptr(), res.data_ptr(), epi.data_ptr(), None, None, st()) == 0

    wsb = [torch.empty(B * 56, device=dev, dtype=torch.float64) if L.new_abi else None for L in libs]

    def cheir(k):
        L, (Rt, win, cnt) = libs[k], outs[k]
        if L.new_abi:  # with the preparation launch (the package's default)
            assert L.dfepe_cheirality_ex(Fo.data_ptr(), TK.data_ptr(), K.data_ptr(), m.data_ptr(), B, N, 50.0, 0, wsb[k].data_ptr(), Rt.data_ptr(), win.data_ptr(), cnt.data_ptr(), st()) == 0
        else:
            assert L.dfepe_cheirality(Fo.data_ptr(), TK.data_ptr(), K.data_ptr(), m.data_ptr(), B, N, 50.0, Rt.data_ptr(), win.data_ptr(), cnt.data_ptr(), st()) == 0

    def cheir_noprep(k):
        L, (Rt, win, cnt) = libs[k], outs[k]
        assert L.dfepe_cheirality(Fo.data_ptr(), TK.data_ptr(), K.data_ptr(), m.data_ptr(), B, N, 50.0, Rt.data_ptr(), win.data_ptr(), cnt.data_ptr(), st()) == 0

    def both(k):
        L, (Rt, win, cnt) = libs[k], outs[k]
        tail = (wsb[k].data_ptr(), st()) if L.new_abi else (st(),)
        assert L.dfepe_w8pt_pose_fwd(m.data_ptr(), w0.data_ptr(), B, N, 1, W, H, 0.5, K.data_ptr(), TK.data_ptr(), 50.0, Fo.data_ptr(), res.data_ptr(),
                                     epi.data_ptr(), None, Rt.data_ptr(), win.data_ptr(), cnt.data_ptr(), *tail) == 0

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fit(libs[1])
        res_t = {}
        for name, fn in (("fit alone", lambda k: fit(libs[k])), ("cheirality alone", cheir), ("cheirality, in-kernel prep", cheir_noprep), ("fit + pose call", both)):
            ts = [[], []]
            for rnd in range(3):
                for k in range(2):
                    ts[k].append(timed(lambda: fn(k)))
            res_t[name] = [statistics.median(t) for t in ts]
        torch.cuda.synchronize()
    torch.cuda.current_stream().wait_stream(side)
    for name, (a, b) in res_t.items():
        print(f"B={B:5d} N={N}  {name:27s}  A {a:8.2f} us   B {b:8.2f} us   B/A {b / a:.3f}", flush=True)
    cheir(0); cheir(1)
    torch.cuda.synchronize()
    ca, cb = outs[0][2], outs[1][2]
    dif = (ca != cb).any(1)
    print(f"B={B:5d}  counts differ on {int(dif.sum())} of {B} pairs (max |diff| {int((ca - cb).abs().max())}); winners differ on {int((outs[0][1] != outs[1][1]).sum())}", flush=True)
