"""Same-box A/B timing of two builds of libdfepe_hip.so (box-to-box variance is ~1 us, more than most kernel changes).
usage (GPU box): python scripts/ab_time.py pytorch-deepfepe_amd/libdfepe_hip.so pytorch-deepfepe_amd/libdfepe_hip_b.so
Times dfepe_w8pt_fwd (B=4096, N=100, weights in, epi + save out) alternately with both libraries."""
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
B, N = 4096, 100
sc = d.synth.make_scene(B, N, seed=1, outlier_ratio=0.2)
m = sc["matches_xy_ori"].cuda(); w = torch.softmax(sc["logits_layers"][0], 1).cuda().contiguous()
Fo = torch.empty(B, 9, device="cuda"); res = torch.empty(B, N, device="cuda"); epi = torch.empty(B, N, device="cuda")
sv = torch.empty(B, 128, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def launch(L):
    rc = L.dfepe_w8pt_fwd(m.data_ptr(), None, w.data_ptr(), B, N, 1, 1, 1241.0, 376.0, 0.5, Fo.data_ptr(), res.data_ptr(), epi.data_ptr(),
                          sv.data_ptr(), None, st)
    assert rc == 0
def t(L, n=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): launch(L)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
for L in libs:
    for _ in range(20): launch(L)
outs = []
for L in libs:
    launch(L); torch.cuda.synchronize(); outs.append(Fo.clone())
print("max |F_A - F_B| =", (outs[0] - outs[1]).abs().max().item())
ts = [[], []]
for rnd in range(12):
    for k, L in enumerate(libs):
        ts[k].append(t(L))
for k in range(2):
    print(f"lib {'AB'[k]}: median {statistics.median(ts[k]):.2f} us  min {min(ts[k]):.2f}  max {max(ts[k]):.2f}")
print(f"B - A = {statistics.median(ts[1]) - statistics.median(ts[0]):+.2f} us")
