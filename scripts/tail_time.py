"""Where the fused loss tail spends its time: dfepe_loss_tail at B = 4096, M = 100, L = 5 with parts switched off
(hipGraph of 20 launches, HIP events).   python scripts/tail_time.py"""
import importlib
import os
import statistics
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
d = importlib.import_module("pytorch-deepfepe_amd")
_lib, ops = d._lib, d.ops
DEV = torch.device("cuda", 0)
B, L, M = 4096, 5, 100
sc = d.pipeline.scene_to_device(d.synth.make_scene(B, 100, seed=1000, outlier_ratio=0.2, noise_px=0.5, depth_layers=L), DEV)
H, W = 376.0, 1241.0
hw_T = torch.tensor([[2.0 / W, 0.0, -1.0], [0.0, 2.0 / H, -1.0], [0.0, 0.0, 1.0]], device=DEV)
m = sc["matches_xy_ori"]
Fs = torch.stack([d.ops.w8pt_forward(m, None, torch.softmax(sc["logits_layers"][l], 1).contiguous(), True, W, H, 0.5, True, False)[0] for l in range(L)]).contiguous()
lib = _lib.lib()
loss_sum = torch.empty(L, B, device=DEV); E = torch.empty(L, B, 3, 3, device=DEV)
q_l2 = torch.empty(L, B, device=DEV); t_l2 = torch.empty(L, B, device=DEV); R_deg = torch.empty(L, B, device=DEV); t_deg = torch.empty(L, B, device=DEV)
sel = torch.empty(L, B, device=DEV, dtype=torch.int32); gF = torch.empty(L, B, 3, 3, device=DEV)
packed = torch.empty(L + 4, device=DEV, dtype=torch.float64); scalars = torch.empty(4 + L, device=DEV)
ws = torch.empty((lib.dfepe_loss_tail_workspace_bytes(B) + 7) // 8, dtype=torch.float64, device=DEV)
v1, v2 = sc["pts1_virt_ori"], sc["pts2_virt_ori"]


def launch(pose=True, grad=True, balance_F=1.0, st=None):
    P = ops._ptr
    rc = lib.dfepe_loss_tail(Fs.data_ptr(), L, B, hw_T.data_ptr(), hw_T.data_ptr(), 0, sc["Ks"].data_ptr(), v1.data_ptr(), v2.data_ptr(), M, 0.02,
                             P(sc["qs_cam"] if pose else None), P(sc["ts_cam"] if pose else None), P(sc["R_gt"] if pose else None), 0.1, 0.5,
                             balance_F, 1.0, 0.1, float(B), loss_sum.data_ptr(), E.data_ptr(), P(q_l2 if pose else None), P(t_l2 if pose else None),
                             P(R_deg if pose else None), P(t_deg if pose else None), P(sel if pose else None), gF.data_ptr() if grad else None,
                             packed.data_ptr(), scalars.data_ptr(), ws.data_ptr(), 1, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


def timed(**kw):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        launch(**kw)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            launch(**kw)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(7):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 20)
    return statistics.median(ts)


for name, kw in [("full (pose + F-loss, forward + adjoint)", {}), ("no pose items", dict(pose=False)), ("F-loss adjoint off (balance_F = 0)", dict(balance_F=0.0)),
                 ("forward only (no g_F)", dict(grad=False)), ("no pose, forward only", dict(pose=False, grad=False))]:
    print(f"{name:45s} {timed(**kw):7.2f} us per launch (incl. the dispatch gap between dependent launches)")
