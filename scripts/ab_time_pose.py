"""Same-box A/B timing of dfepe_pose_fwd / dfepe_pose_bwd of two library builds (L=5, B=4096)."""
import ctypes, importlib, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
libs = []
for path in sys.argv[1:3]:
    L = ctypes.CDLL(os.path.abspath(path))
    L.dfepe_pose_fwd.restype = I
    L.dfepe_pose_fwd.argtypes = [P, I, I, P, P, P, P, P, P, P, P, P]
    L.dfepe_pose_bwd.restype = I
    L.dfepe_pose_bwd.argtypes = [P, I, I, P, P, P, P, F, F, F, F, P, P, P]
    libs.append(L)
Ln, B = 5, 4096
sc = d.pipeline.scene_to_device(d.synth.make_scene(B, 100, seed=1, outlier_ratio=0.2), "cuda:0")
out = d.pipeline.hot_path_step(sc, [376, 1241, 3], Ln, 0.02, qt=True)
E = out["E_layers"].contiguous()
q_gt = sc["qs_cam"].reshape(B, 4).contiguous(); t_gt = sc["ts_cam"].reshape(B, 3).contiguous(); R_gt = sc["R_gt"].contiguous()
ql = torch.empty(Ln, B, device="cuda"); tl = torch.empty(Ln, B, device="cuda"); Rd = torch.empty(Ln, B, device="cuda"); td = torch.empty(Ln, B, device="cuda")
sel = torch.empty(Ln, B, device="cuda", dtype=torch.int32); gE = torch.empty(Ln, B, 9, device="cuda"); gs = torch.ones(1, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def fwd(L):
    assert L.dfepe_pose_fwd(E.data_ptr(), Ln, B, q_gt.data_ptr(), t_gt.data_ptr(), R_gt.data_ptr(), ql.data_ptr(), tl.data_ptr(), Rd.data_ptr(), td.data_ptr(), sel.data_ptr(), st) == 0
def bwd(L):
    assert L.dfepe_pose_bwd(E.data_ptr(), Ln, B, q_gt.data_ptr(), t_gt.data_ptr(), None, None, 1.0 / (Ln * B), 0.1, 0.1 / (Ln * B), 0.5, gs.data_ptr(), gE.data_ptr(), st) == 0
def t(f, L, n=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f(L)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
res = []
for L in libs:
    for _ in range(10): fwd(L); bwd(L)
    torch.cuda.synchronize(); res.append((ql.clone(), tl.clone(), Rd.clone(), td.clone(), gE.clone()))
for name, a, b in zip(("q_l2", "t_l2", "R_deg", "t_deg", "g_E"), res[0], res[1]):
    print(f"max |{name}_A - {name}_B| = {(a - b).abs().max().item():.2e}")
for f, nm in ((fwd, "pose_fwd"), (bwd, "pose_bwd")):
    ts = [[], []]
    for rnd in range(10):
        for k, L in enumerate(libs): ts[k].append(t(f, L))
    print(f"{nm}: A {statistics.median(ts[0]):.2f} us  B {statistics.median(ts[1]):.2f} us  B - A = {statistics.median(ts[1]) - statistics.median(ts[0]):+.2f} us")
