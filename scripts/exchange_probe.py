#!/usr/bin/env python3
"""What does each way of getting the (L+4)-double loss vector to the other ranks cost the solver stream?  One-rank RCCL group,
the captured config-3 step (B = 4096), us per step over 400 replays:
  none        the step alone
  record      + one event record on the solver stream per step
  rec+wait    + a wait on the event another stream recorded two steps ago
  sync        + all_reduce(packed) in stream order
  lagged      two alternately replayed graphs (own output buffers each); all_reduce(async_op=True) of the finished step's packed on
              RCCL's stream; its completion is waited for two steps later, just before the same graph is replayed again
  branch      the all-reduce captured as a branch of the step's graph (hot_path_fused(loss_exchange=...))
GPU box:  python scripts/exchange_probe.py"""
import importlib
import os
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
IMAGE_SIZE = [376, 1241, 3]
dfepe = importlib.import_module("pytorch-deepfepe_amd")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
L, B, N = 5, 4096, 100
sc = dfepe.pipeline.scene_to_device(dfepe.synth.make_scene(B, N, seed=1000, outlier_ratio=0.2, noise_px=0.5, depth_layers=L), dev)
H, W = float(IMAGE_SIZE[0]), float(IMAGE_SIZE[1])
hw_T = torch.tensor([[2.0 / W, 0.0, -1.0], [0.0, 2.0 / H, -1.0], [0.0, 0.0, 1.0]], device=dev)
logits = sc["logits_layers"][:L].clone().requires_grad_(True)


def capture(loss_exchange=None, branch=False):
    st = {}

    def body():
        out = dfepe.pipeline.hot_path_fused(sc["matches_xy_ori"], logits, sc["Ks"], sc["pts1_virt_ori"], sc["pts2_virt_ori"], sc["qs_cam"],
                                            sc["ts_cam"], sc["R_gt"], IMAGE_SIZE, 0.02, True, hw_T=hw_T, grad_pairs=B, defer_loss_head=True,
                                            loss_exchange=loss_exchange, exchange_branch=branch)
        st["g"], = torch.autograd.grad(out["loss"], logits, grad_outputs=st.setdefault("seed", torch.ones_like(out["loss"])))
        return out

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = body()
    return g, out, st


def timeit(step, n=400, warm=200):
    for i in range(warm):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


gA, outA, _ = capture()
gB, outB, _ = capture()
res = {}
res["none"] = timeit(lambda i: gA.replay())
ev = [torch.cuda.Event() for _ in range(4)]


def rec(i):
    gA.replay()
    ev[i % 4].record()


res["record"] = timeit(rec)
other = torch.cuda.Stream()
ev2 = [torch.cuda.Event() for _ in range(4)]
for e in ev2:
    e.record(other)


def recwait(i):
    torch.cuda.current_stream().wait_event(ev2[i % 4])
    gA.replay()
    ev[i % 4].record()
    other.wait_event(ev[i % 4])
    ev2[i % 4].record(other)


res["rec+wait"] = timeit(recwait)


def sync(i):
    gA.replay()
    dist.all_reduce(outA["packed"])


res["sync"] = timeit(sync)
works = [None, None]
graphs = [(gA, outA), (gB, outB)]


def lagged(i):
    k = i & 1
    if works[k] is not None:
        works[k].wait()  # the all-reduce of this graph's previous packed (two steps ago) before the graph rewrites it
    g, o = graphs[k]
    g.replay()
    works[k] = dist.all_reduce(o["packed"], async_op=True)


res["lagged"] = timeit(lagged)
for w in works:
    if w is not None:
        w.wait()
gC, outC, _ = capture(lambda p: dist.all_reduce(p), branch=True)
res["branch"] = timeit(lambda i: gC.replay())
# the same all-reduce captured in stream order at the end of the step's graph (no branch)
st = {}


def body_tail():
    out = dfepe.pipeline.hot_path_fused(sc["matches_xy_ori"], logits, sc["Ks"], sc["pts1_virt_ori"], sc["pts2_virt_ori"], sc["qs_cam"],
                                        sc["ts_cam"], sc["R_gt"], IMAGE_SIZE, 0.02, True, hw_T=hw_T, grad_pairs=B, defer_loss_head=True)
    st["g"], = torch.autograd.grad(out["loss"], logits, grad_outputs=st.setdefault("seed", torch.ones_like(out["loss"])))
    dist.all_reduce(out["packed"])
    return out


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    body_tail()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
gD = torch.cuda.CUDAGraph()
with torch.cuda.graph(gD):
    body_tail()
res["in-graph, end of step"] = timeit(lambda i: gD.replay())
for k, v in res.items():
    print(f"{k:24s} {v:8.1f} us/step   (+{v - res['none']:.1f})")
dist.destroy_process_group()
