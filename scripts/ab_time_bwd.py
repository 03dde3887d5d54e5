"""Same-box A/B timing of dfepe_w8pt_bwd of two library builds (fused-step configuration: logits mode, g_F only)."""
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
    L.dfepe_w8pt_bwd.restype = I
    L.dfepe_w8pt_bwd.argtypes = [P, P, P, I, I, I, U, F, F, F, P, P, P, P, P, P, P, P, P, P, P, P]
    libs.append(L)
B, N = 4096, 100
sc = d.synth.make_scene(B, N, seed=1, outlier_ratio=0.2)
m = sc["matches_xy_ori"].cuda(); lg = sc["logits_layers"][0].cuda().contiguous()
Fo = torch.empty(B, 9, device="cuda"); res = torch.empty(B, N, device="cuda"); epi = torch.empty(B, N, device="cuda")
sv = torch.empty(B, 128, device="cuda"); wo = torch.empty(B, N, device="cuda")
gF = torch.randn(B, 9, device="cuda"); gl = torch.empty(B, N, device="cuda")
st = torch.cuda.current_stream().cuda_stream
assert libs[0].dfepe_w8pt_fwd(m.data_ptr(), None, lg.data_ptr(), B, N, 1, 3, 1241.0, 376.0, 0.5, Fo.data_ptr(), res.data_ptr(), epi.data_ptr(),
                               sv.data_ptr(), wo.data_ptr(), st) == 0
def launch(L):
    rc = L.dfepe_w8pt_bwd(m.data_ptr(), None, wo.data_ptr(), B, N, 1, 3, 1241.0, 376.0, 0.5, sv.data_ptr(), Fo.data_ptr(), gF.data_ptr(), None, None,
                          None, None, gl.data_ptr(), None, None, None, st)  # g_weights_extra, g_scale, g_weights, g_pts1, g_pts2, pending head
    assert rc == 0
def t(L, n=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): launch(L)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
outs = []
for L in libs:
    for _ in range(20): launch(L)
    torch.cuda.synchronize(); outs.append(gl.clone())
print("max |g_A - g_B| / max|g| =", ((outs[0] - outs[1]).abs().max() / outs[0].abs().max()).item())
ts = [[], []]
for rnd in range(12):
    for k, L in enumerate(libs):
        ts[k].append(t(L))
for k in range(2):
    print(f"lib {'AB'[k]}: median {statistics.median(ts[k]):.2f} us  min {min(ts[k]):.2f}")
print(f"B - A = {statistics.median(ts[1]) - statistics.median(ts[0]):+.2f} us")
