"""Single-GPU batch sweep that predicts the 1 -> 8 GPU curves.

    python scripts/batch_sweep.py [out.md]

For B in {512 ... 32768}: the captured training step of configs 3 / 4 (N = 100, depth 5, fwd+bwd; config 4 = 40 % outliers,
qt-only objective) and the config-5 step (N = 1000: fit + E-from-F + cheirality) are timed as hipGraph replays (HIP events, median
of 5 rounds).  A one-rank RCCL all-reduce of the packed (L+4)-double loss vector is timed the same way: its latency on 8 ranks
over xGMI is larger (a few tens of us), so the table carries both the measured one-rank figure and a 30 us allowance.
The weak curve is t(B_per_gpu) constant in N by construction (no data-path collective); strong scaling of a total batch B_tot over G
ranks is t(B_tot) / (t(B_tot / G) + all-reduce).
"""
import importlib
import os
import statistics
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
d = importlib.import_module("pytorch-deepfepe_amd")
IMG = [376, 1241, 3]
DEV = torch.device("cuda", 0)
BS = [512, 1024, 2048, 4096, 8192, 16384, 32768]


def graph_time_us(body, reps=20, rounds=5):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(rounds):
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return statistics.median(ts)


def train_step(B, outl, balance_F, exchange=None):
    L = 5
    sc = d.pipeline.scene_to_device(d.synth.make_scene(B, 100, seed=1000, outlier_ratio=outl, noise_px=0.5, depth_layers=L), DEV)
    H, W = float(IMG[0]), float(IMG[1])
    hw_T = torch.tensor([[2.0 / W, 0.0, -1.0], [0.0, 2.0 / H, -1.0], [0.0, 0.0, 1.0]], device=DEV)
    logits = sc["logits_layers"][:L].clone().requires_grad_(True)

    seed = torch.ones((), device=DEV)

    def body():
        out = d.pipeline.hot_path_fused(sc["matches_xy_ori"], logits, sc["Ks"], sc["pts1_virt_ori"], sc["pts2_virt_ori"], sc["qs_cam"], sc["ts_cam"],
                                        sc["R_gt"], IMG, clamp_at=0.02, qt=True, hw_T=hw_T, balance_F=balance_F, grad_pairs=B, defer_loss_head=True,
                                        loss_exchange=exchange)
        torch.autograd.grad(out["loss"], logits, grad_outputs=seed.reshape(out["loss"].shape))

    return graph_time_us(body)


def pose_step(B):
    sc = d.pipeline.scene_to_device(d.synth.make_scene(B, 1000, seed=1000, outlier_ratio=0.2, noise_px=0.5), DEV)
    H, W = float(IMG[0]), float(IMG[1])
    hw_T = torch.tensor([[2.0 / W, 0.0, -1.0], [0.0, 2.0 / H, -1.0], [0.0, 0.0, 1.0]], device=DEV)
    TK = (hw_T @ sc["Ks"]).contiguous()
    w0 = torch.softmax(sc["logits_layers"][0], dim=1).contiguous()
    m = sc["matches_xy_ori"]

    def body():  # bench.py --config 5: one launch for the cooperative shapes (<= 3072 pairs), two otherwise
        d.ops.fit_pose(m, w0, sc["Ks"], W, H, 50.0, pre=TK)

    fit = graph_time_us(lambda: d.ops.w8pt_forward(m, None, w0, True, W, H, 0.5, True, False))
    return graph_time_us(body), fit


def allreduce_us():
    """Eager in-stream all-reduce of the 9-double vector on the (already initialised) one-rank group."""
    import torch.distributed as dist

    v = torch.zeros(9, dtype=torch.float64, device=DEV)
    for _ in range(5):
        dist.all_reduce(v)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(50):
            dist.all_reduce(v)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 50)
    return statistics.median(ts)


def main():
    import torch.distributed as dist

    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "gpurun_out", "r04_batch_sweep.md")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=DEV)
    ex = lambda p: dist.all_reduce(p)
    t3, t4, t5, f5, t4x = {}, {}, {}, {}, {}
    for B in BS:
        t3[B] = train_step(B, 0.2, 1.0)
        t4[B] = train_step(B, 0.4, 0.0)
        t4x[B] = train_step(B, 0.4, 0.0, exchange=ex)  # the all-reduce captured as the last node of the step's graph (bench.py's default)
        if B <= 8192:
            t5[B], f5[B] = pose_step(B)
        print(B, t3[B], t4[B], t4x[B], t5.get(B), f5.get(B), flush=True)
    ar = allreduce_us()
    dist.destroy_process_group()
    one_rank = t4x[4096] - t4[4096]
    lines = ["# Batch sweep on one MI355X (hipGraph replays, HIP events, median of 5 x 20 replays)", "",
             f"The data-parallel exchange (all-reduce of the packed 9-double loss vector) is the last node of the step's hipGraph "
             f"(`hot_path_fused(loss_exchange=...)`, bench.py's default).  On a ONE-rank RCCL group it costs **{one_rank:+.1f} us** per step at 4096 "
             f"pairs (column `+ exchange`; the same all-reduce launched eagerly in stream order: {ar:.1f} us; as a graph branch parallel to the "
             "backward: +33 us, `scripts/exchange_probe.py`).  What it costs on EIGHT ranks over xGMI cannot be measured here (no 8-GPU node "
             "under this lease): the predictions carry the measured one-rank figure and, beside it, allowances of 15 and 30 us for the "
             "small-message latency of an 8-rank all-reduce that runs in stream order at the end of the step.", "",
             "| pairs B | config 3 step (us) | Mpairs/s | config 4 step (us) | + exchange (us) | Mpairs/s | config 5 step (us) | of which fit (us) | Mpairs/s |",
             "|---|---|---|---|---|---|---|---|---|"]
    for B in BS:
        c5 = f"{t5[B]:.1f} | {f5[B]:.1f} | {B / t5[B]:.2f}" if B in t5 else "– | – | –"
        lines.append(f"| {B} | {t3[B]:.1f} | {B / t3[B]:.2f} | {t4[B]:.1f} | {t4x[B]:.1f} | {B / t4[B]:.2f} | {c5} |")
    lines += ["", "## Predicted curves", ""]
    for label, AR8 in (("the measured one-rank cost of the captured exchange", max(one_rank, 0.0)), ("15 us for the 8-rank all-reduce", 15.0),
                       ("30 us for the 8-rank all-reduce", 30.0)):
        lines += [f"### with {label}", "",
                  "Weak scaling (configs 3 / 4: 4096 pairs per GPU, no data-path collective):", "",
                  "| GPUs | step (us) | Mpairs/s (config 3) | efficiency |", "|---|---|---|---|"]
        for G in (1, 2, 4, 8):
            t = t3[4096] + (AR8 if G > 1 else 0.0)
            lines.append(f"| {G} | {t:.1f} | {G * 4096 / t:.1f} | {t3[4096] / t:.2f} |")
        lines += ["", "Strong scaling, config 4 as one global batch of 32768 pairs (`bench.py --config 4 --scaling strong`): t(32768) / (t(32768 / G) + exchange):", "",
                  "| GPUs | pairs per GPU | step (us) | speed-up vs 1 GPU |", "|---|---|---|---|"]
        for G in (1, 2, 4, 8):
            b = 32768 // G
            t = t4[b] + (AR8 if G > 1 else 0.0)
            lines.append(f"| {G} | {b} | {t:.1f} | {t4[32768] / t:.2f} |")
        lines.append("")
    lines += ["The ratio t(32768) / t(4096) is the ceiling of the 8-GPU strong-scaling figure (exchange free): one GPU runs the 32768-pair batch "
              "at a higher rate per pair than the 4096-pair one (eight wavefronts per SIMD hide each other's issue bubbles; at 4096 pairs every "
              "SIMD holds ONE), so every improvement of the large-batch rate lowers the ratio.", "",
              "Strong scaling, config 5 (4096 pairs in total, N = 1000; no loss, hence no collective): t(4096) / t(4096 / G):", "",
              "| GPUs | pairs per GPU | step (us) | speed-up vs 1 GPU |", "|---|---|---|---|"]
    for G in (1, 2, 4, 8):
        b = 4096 // G
        lines.append(f"| {G} | {b} | {t5[b]:.1f} | {t5[4096] / t5[b]:.2f} |")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
