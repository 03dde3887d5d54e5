#!/usr/bin/env python3
"""bench.py — throughput of the deepFEPE weighted-8-point hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {2,3,4,5}] [--scaling {weak,strong}]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Default (no flags) = BASELINE.json config 3, the configuration the metric is quoted on; one "step" = one pass of the hot
path over one resident batch of synthetic pairs:  depth=5 x [softmax -> weighted normalised 8-point fit + in-loop epipolar
residual]  ->  F-loss on 100 virtual points  ->  E = K^T F K  ->  quaternion/translation pose loss, then backward to the
per-layer logits.  Inputs already live in HBM when the timed region starts.  The other BASELINE configs are selectable:
    2  B=1024, N=100, single weighted-8-point fit (forward) + E-from-F
    4  B=4096 per GPU (x8 = 32768), 40 % outliers, qt pose-loss objective (the reference's if_qt_loss mixing: the F-loss is
       evaluated but dropped from the objective, Train_model_pipeline.py:580-587), forward + backward
    5  B=4096 in total, N=1000, one fit + E-from-F + cheirality-checked pose; strong scaling (the batch is split over ranks)
Data parallel over N ranks: every rank owns its shard of independent pairs (the pairs never interact), the only exchange
is one all-reduce of the small loss vector per step (RCCL) where the step has a loss.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel against the HBM roofline,
measured with HIP events on the launch stream) and `cpu_baseline` (the CPU oracle on a bounded sample of the same workload).
"""
import argparse
import importlib
import json
import os

# before the HIP runtime initialises: see pytorch-deepfepe_amd/__init__.py (ROCm 7.2 hipGraph replays are wrong for the full model's
# captured step with the runtime's graph packet capture, and 3 % slower for the timed step); logged in config.env
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import statistics
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

IMAGE_SIZE = [376, 1241, 3]
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
METRIC = "image-pairs/sec (F+E+pose+loss) at B=4096 N=100; median R/t angular err vs ref"

CONFIGS = {
    2: dict(B=1024, N=100, depth=1, outliers=0.2, scaling="weak", kind="fit",
            what="BASELINE config 2: B={B}/GPU, N={N}: one weighted-8-point fit (forward, in-loop epipolar residual) + E-from-F"),
    3: dict(B=4096, N=100, depth=5, outliers=0.2, scaling="weak", kind="train", balance_F=1.0,
            what="BASELINE config 3: B={B}/GPU, N={N}, depth={L} weighted-8-point fits with fixed per-layer logits + in-loop epipolar "
                 "residual + F-loss (100 virtual pts) + E-from-F + qt pose loss, forward+backward to the logits"),
    4: dict(B=4096, N=100, depth=5, outliers=0.4, scaling="weak", kind="train", balance_F=0.0,
            what="BASELINE config 4: B={B}/GPU (x8 GPUs = 32768), N={N}, 40 % outliers, depth={L}, qt pose-loss objective (F-loss evaluated, "
                 "dropped from the objective like the reference's if_qt_loss), forward+backward to the logits"),
    5: dict(B=4096, N=1000, depth=1, outliers=0.2, scaling="strong", kind="pose",
            what="BASELINE config 5: B={B} pairs in total, N={N}: one weighted-8-point fit + E-from-F + cheirality-checked R,t (depth_thres 50)"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--spinup", type=int, default=200, help="untimed steps in front of the warm-up steps (clock / cache settling)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS))
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None, help="default: weak for configs 2-4, strong for config 5")
    ap.add_argument("--batch", type=int, default=None, help="pairs per GPU (weak) or in total (strong); default: the config's")
    ap.add_argument("--npoints", type=int, default=None)
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--outliers", type=float, default=None)
    ap.add_argument("--blocks", type=int, default=10, help="informational: repeated 20-step blocks after the timed region (median, spread)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-model", action="store_true", help="skip the secondary whole-DeepFNet measurement")
    ap.add_argument("--no-defer-head", action="store_true", help="launch the loss head as a kernel of its own instead of finishing it inside the first backward launch")
    ap.add_argument("--no-extras", action="store_true", help="skip every informational field (layers_batched, full_model, match_construction ...)")
    ap.add_argument("--cpu-sample", type=int, default=512, help="pairs in the CPU-baseline sample (~10 s of host work per pass)")
    ap.add_argument("--force-dist", action="store_true",
                    help="single-process smoke test of the multi-GPU code path: a 1-rank RCCL group and the overlapped exchange")
    ap.add_argument("--launcher", choices=("auto", "torchrun", "none"), default="auto",
                    help="auto: a plain `python bench.py --gpus N` (no RANK in the environment) with N > 1 re-executes itself under "
                         "torch.distributed.run with N ranks; torchrun: do that even for N = 1 (exercises the launch path on a 1-GPU box); "
                         "none: never re-execute")
    return ap.parse_args(argv)


def capture_mode():
    """"thread_local" once an RCCL group exists (its watchdog thread must be free to poll events while this thread captures:
    pytorch-deepfepe_amd/dist.py graph_capture_mode), torch's default otherwise."""
    import torch.distributed as dist

    return "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_command(argv, gpus, port):
    """The command a plain `python bench.py --gpus N` turns itself into: the driver's own N > 1 invocation."""
    argv = list(argv)
    while "--launcher" in argv:  # the ranks must not launch again
        i = argv.index("--launcher")
        del argv[i:i + 2]
    argv = [a for a in argv if not a.startswith("--launcher=")]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__), *argv, "--launcher", "none"]


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, RCCL).  The
    reference's counterpart is one process driving every GPU through nn.DataParallel (deepFEPE/train_good.py:311-312).
    Rank 0's JSON line passes through this process's stdout; the launcher's exit code is returned."""
    import subprocess

    have = torch.cuda.device_count()
    if have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: this node exposes {have} GPU(s)")
    cmd = launch_command(sys.argv[1:], args.gpus, _free_port())
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    log("self-launch:", " ".join(cmd))
    return subprocess.run(cmd, env=env).returncode


def resolve_workload(args, rank, world, shard_range):
    """What this rank runs: the BASELINE config's shape with the command-line overrides, and the rank's share of it.  Weak scaling:
    `batch` pairs per GPU (config 4's 8 x 4096 = 32768 at --gpus 8); strong: `batch` pairs in total, contiguous shards
    (dist.shard_range), `--config 4 --scaling strong` = BASELINE config 4 as north_star words it (32768 pairs over the ranks).
    grad_pairs = the pairs the batch means of the loss run over = the GLOBAL batch (the reference's means divide by it)."""
    cfg = dict(CONFIGS[args.config])
    scaling = args.scaling or cfg["scaling"]
    B_cfg = args.batch if args.batch is not None else cfg["B"]
    if args.config == 4 and scaling == "strong" and args.batch is None:
        B_cfg = 32768
    N = args.npoints if args.npoints is not None else cfg["N"]
    L = args.depth if args.depth is not None else cfg["depth"]
    outl = args.outliers if args.outliers is not None else cfg["outliers"]
    if scaling == "strong":
        a, b = shard_range(B_cfg, rank, world)
        B, B_total = b - a, B_cfg
    else:
        B, B_total = B_cfg, B_cfg * world
    return {"cfg": cfg, "scaling": scaling, "B_cfg": B_cfg, "N": N, "L": L, "outliers": outl, "kind": cfg["kind"], "B": B, "B_total": B_total,
            "grad_pairs": B_total}


def log(*a):
    if os.environ.get("DFEPE_BENCH_VERBOSE"):
        print(f"[bench {time.perf_counter():.2f}]", *a, file=sys.stderr, flush=True)


def event_time_us(fn, reps=50, rounds=5, warm=5, graph=True):
    """Average GPU time of one call of `fn`: `reps` back-to-back calls captured in a hipGraph (so that the host launch path --
    ctypes, tensor allocation -- cannot be the bottleneck of a 10 us kernel), replayed between two HIP events on the launch
    stream; median over `rounds` replays.  Includes the ~1-2 us dispatch gap between dependent launches."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = None
    if graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode=capture_mode()):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    durs = []
    for _ in range(rounds):
        e0.record()
        if g is not None:
            g.replay()
        else:
            for _ in range(reps):
                fn()
        e1.record()
        torch.cuda.synchronize()
        durs.append(e0.elapsed_time(e1) * 1e3 / reps)
    return statistics.median(durs)


def count_launches(fn):
    """Kernel launches of one call of `fn` (torch.profiler sees every HIP kernel of the process, the ctypes launches included);
    None when the profiler is unavailable."""
    try:
        from torch.profiler import ProfilerActivity, profile

        fn()
        torch.cuda.synchronize()
        names, n, n_api = {}, 0, 0
        for _ in range(3):  # the ROCm tracer behind torch.profiler drops records now and then (round 6: 164, 94 and 17 "launches" for the
            # same step on three boxes): a count can only come out too LOW, so the largest of three passes is kept
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
                fn()
                torch.cuda.synchronize()
            seen = {}
            api = 0  # the launch calls themselves, as the tracer saw them on the host side (hipLaunchKernel, hipExtModuleLaunchKernel ...)
            for ev in prof.events():
                if str(getattr(ev, "device_type", "")).endswith("CUDA") and "memcpy" not in ev.name.lower() and "memset" not in ev.name.lower():
                    seen[ev.name] = seen.get(ev.name, 0) + 1
                elif ev.name.startswith("hip") and "Launch" in ev.name and "Graph" not in ev.name:
                    api += 1
            if sum(seen.values()) > n:
                names, n = seen, sum(seen.values())
            n_api = max(n_api, api)
        ours = sum(c for k, c in names.items() if any(t in k for t in ("w8pt", "loss_tail", "loss_stats", "floss", "pose_", "geo_misc", "deepf_input", "row_dot")))
        # device-side records get lost (long kernels, full buffers), host-side launch calls do not: the larger count is the step's
        return {"total": max(n, n_api), "hip_kernels_of_this_library": ours, "torch_glue": n - ours, "kernel_records": n, "launch_calls": n_api} if (n or n_api) else None
    except Exception:
        return None


def measure_api_path(dfepe, scene, logits, L, balance_F, args, B, fused_ms, fused_grad):
    """The reference's call sequence on the benchmark's batch, three ways: ground truth only at get_Rt_loss (the reference's
    signature, stand-alone F-loss and pose kernels), ground truth also in loss_params (one fused tail launch), and the latter
    with a linear probe estimator that gives every layer but the last the recurrent model's backward.  Each eager and as a
    hipGraph replay."""
    pl = dfepe.pipeline
    out = {"what": "compat.DeepFNet.forward (estimators replaced by the benchmark's fixed per-layer logits) + compat.get_all_loss_DeepF + "
                   "compat.get_Rt_loss + clamp(stack(...)).mean() * balance mixing in torch + backward to the logits; same batch, same "
                   "objective as `value`"}
    for key, gt_in, probe in (("reference_signature", False, False), ("pose_gt_in_loss_params", True, False),
                              ("recurrent_shape", True, True)):
        rows = [logits[l].detach().clone().unsqueeze(1).requires_grad_(True) for l in range(L)]
        net = pl.make_api_net(L, IMAGE_SIZE, rows, recurrent_probe=probe)
        st = {}

        def body():
            loss, outs, losses, geo = pl.reference_call_sequence(net, scene, L, balance_F=balance_F, pose_gt_in_loss_params=gt_in)
            st["g"] = torch.autograd.grad(loss, rows, grad_outputs=st.setdefault("seed", torch.ones_like(loss)))
            st["loss"] = loss

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        rec = {"launches": count_launches(body)}
        if not probe and fused_grad is not None:
            g = torch.stack([x.squeeze(1) for x in st["g"]])
            rec["max_abs_grad_diff_vs_value_run"] = float((g - fused_grad).abs().max())
            rec["grad_scale"] = float(fused_grad.abs().max())
        n_e = max(20, min(args.steps, 100))
        tgu = dfepe.compat.train_good_utils

        def eager_ms():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_e):
                body()
            torch.cuda.synchronize()
            return round((time.perf_counter() - t0) * 1e3 / n_e, 4)

        # the package default returns the reference's host types from get_Rt_loss (one device-to-host copy per step, like the
        # reference's .cpu().numpy()); LAZY_HOST_METRICS defers that copy to the first read (no synchronisation in the step)
        rec["eager_ms_per_step"] = eager_ms()
        tgu.LAZY_HOST_METRICS = True
        try:
            rec["eager_lazy_host_metrics_ms_per_step"] = eager_ms()
            # compat.CapturedStep: what a caller of the reference's eager sequence gets without building a graph itself -- the helper
            # captures forward + losses + the caller's mixing + backward on its third call and replays from then on, copying every
            # batch into its static inputs (two different batches alternate here, so each replay pays its copy-in)
            if not probe:
                def fwd_loss(sc):
                    return pl.reference_call_sequence(net, sc, L, balance_F=balance_F, pose_gt_in_loss_params=gt_in)[0], None
                keys = ("matches_xy_ori", "pts1_virt_ori", "pts2_virt_ori", "Ks", "delta_Rtijs_4_4", "qs_cam", "ts_cam")
                two = [{k: scene[k] for k in keys}, {k: scene[k].clone() for k in keys}]
                helper = dfepe.compat.CapturedStep(fwd_loss, rows, warmup=2)
                for k in range(args.warmup + 8):
                    helper(two[k & 1])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in range(args.steps):
                    helper(two[k & 1])
                torch.cuda.synchronize()
                hms = (time.perf_counter() - t0) * 1e3 / args.steps
                rec["captured_helper_ms_per_step"] = round(hms, 4)
                rec["captured_helper"] = {"eager_steps": helper.n_eager, "captures": helper.n_captures, "replays": helper.n_replays,
                                          "note": "compat.CapturedStep(forward_and_loss, params): copy-in of a fresh batch + one hipGraph replay per call"}
                if fused_grad is not None:
                    gh = torch.stack([x.grad.squeeze(1) for x in rows])
                    rec["captured_helper"]["max_abs_grad_diff_vs_value_run"] = float((gh - fused_grad).abs().max())
                for x in rows:
                    x.grad = None
                del helper
        finally:
            tgu.LAZY_HOST_METRICS = False
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, capture_error_mode=capture_mode()):
            body()
        for _ in range(args.warmup + 20):
            gr.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            gr.replay()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / args.steps
        rec["hipgraph_ms_per_step"] = round(ms, 4)
        rec["pairs_per_s"] = round(B / ms * 1e3, 1)
        rec["vs_fused_entry_point"] = round(ms / fused_ms, 3)
        out[key] = rec
        del gr, net
    return out


def main():
    args = parse()
    launched = "RANK" in os.environ or "LOCAL_RANK" in os.environ
    if not launched and args.launcher != "none" and (args.gpus > 1 or args.launcher == "torchrun"):
        raise SystemExit(self_launch(args))
    # stdout carries exactly ONE line, the result JSON.  RCCL prints a version banner to the C-level stdout (flushed at exit,
    # i.e. after the JSON), so descriptor 1 is pointed at stderr for the whole run and the line is written to the saved one.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (plain `python bench.py --gpus N` does it itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU implementation)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # the steps are captured on side streams while the parameters' AccumulateGrad nodes were made on the default one: intended
    quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
    if quiet is not None:
        quiet(False)
    dist = None
    if world > 1 or args.force_dist or launched:
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        dist = dist_mod

    dfepe = importlib.import_module("pytorch-deepfepe_amd")
    wl = resolve_workload(args, rank, world, dfepe.dist.shard_range)
    cfg, scaling, B_cfg, N, L, outl, kind, B, B_total = (wl[k] for k in ("cfg", "scaling", "B_cfg", "N", "L", "outliers", "kind", "B", "B_total"))
    # every rank generates only its own pairs (seeded per rank): the shards are independent by construction
    scene = dfepe.pipeline.scene_to_device(
        dfepe.synth.make_scene(B, N, seed=1000 + rank, outlier_ratio=outl, noise_px=0.5, depth_layers=L), dev)
    H, W = float(IMAGE_SIZE[0]), float(IMAGE_SIZE[1])
    hw_T = torch.tensor([[2.0 / W, 0.0, -1.0], [0.0, 2.0 / H, -1.0], [0.0, 0.0, 1.0]], device=dev)
    logits = scene["logits_layers"][:L].clone().requires_grad_(kind == "train")
    state = {}  # tensors produced inside the captured graph are static: replays rewrite them in place
    m = scene["matches_xy_ori"]
    w0 = torch.softmax(scene["logits_layers"][0], dim=1).contiguous()
    TK = (hw_T @ scene["Ks"]).contiguous()  # per-pair constant of E = (T K)^T F (T K), formed once

    # the only exchange of the data-parallel path: (L+4) doubles over RCCL/xGMI per step (steps without a loss exchange nothing).
    # Default ("graph"): the all-reduce is the last node of the step's hipGraph (pipeline.hot_path_fused(loss_exchange=...)): the host
    # enqueues nothing per step but the replay.  scripts/exchange_probe.py on a one-rank RCCL group, us per step over the plain step:
    # in the graph +0, eager in stream order after the replay (DFEPE_BENCH_EXCHANGE=sync, round 3) +10, as a graph branch parallel
    # to the backward (=branch) +33, lagged on RCCL's stream with event waits (=overlap: dist.OverlappedLossExchange) +26..29 --
    # every cross-stream edge costs this stack more than the 72-byte collective it would hide.
    has_loss = kind == "train"
    exchange_mode = os.environ.get("DFEPE_BENCH_EXCHANGE", "graph") if (dist is not None and has_loss) else "none"
    loss_exchange = (lambda p: dist.all_reduce(p)) if exchange_mode in ("graph", "branch") else None

    if kind == "train":
        def step_body():
            out = dfepe.pipeline.hot_path_fused(m, logits, scene["Ks"], scene["pts1_virt_ori"], scene["pts2_virt_ori"], scene["qs_cam"],
                                                  scene["ts_cam"], scene["R_gt"], IMAGE_SIZE, clamp_at=0.02, qt=True, hw_T=hw_T,
                                                  balance_F=cfg["balance_F"], grad_pairs=B_total, defer_loss_head=not args.no_defer_head,
                                                  loss_exchange=loss_exchange, exchange_branch=exchange_mode == "branch")
            # the seed d loss / d loss = 1 is a constant of the loop: allocated once (first eager warm-up step) instead of the
            # ones_like() fill that autograd would otherwise launch in every step
            if "seed" not in state:
                state["seed"] = torch.ones_like(out["loss"])
            g, = torch.autograd.grad(out["loss"], logits, grad_outputs=state["seed"])
            state["grad_logits"] = g           # d loss / d logits: what the estimator's backward / an optimizer consumes
            state["loss_vec"] = out["packed"]  # dist.pack_loss_sums layout (L+4 doubles): the ONLY data exchanged between ranks
            return out
    elif kind == "fit":
        def step_body():
            F, res, epi, _, _ = dfepe.ops.w8pt_forward(m, None, w0, True, W, H, 0.5, True, False)
            state["E"] = dfepe.ops.congruence(F, TK)
            return {"F_layers": F.unsqueeze(0), "residual": res, "epi": epi}
    else:  # "pose": fit + E-from-F + cheirality-checked decomposition (the (1,1,0) projection is implied by the decomposition)
        def step_body():
            # E = (T K)^T F (T K) is formed inside; one launch when a cooperative workgroup serves the pair (small batches), else two
            F, res, epi, _, Rt, winner, counts = dfepe.ops.fit_pose(m, w0, scene["Ks"], W, H, 50.0, pre=TK)
            state["Rt"], state["winner"] = Rt, winner
            return {"F_layers": F.unsqueeze(0), "Rt": Rt, "winner": winner, "counts": counts}

    # eager warm-up (also sizes the caching allocator), then optional graph capture of the whole step
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            last = step_body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    log("eager warm-up done")
    graph = None
    exchange_fallback = None
    if not args.no_graph:
        captured = 1
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode=capture_mode()):
                last = step_body()
        except Exception as exc:  # only a captured collective is allowed to fail: the step then keeps it outside the graph
            if exchange_mode not in ("graph", "branch"):
                raise
            captured, graph, exchange_fallback = 0, None, repr(exc)[:200]
            log("capturing the step with the all-reduce inside failed:", exchange_fallback)
            torch.cuda.synchronize()
        if exchange_mode in ("graph", "branch"):
            # every rank must run the same variant: agree on the weakest outcome, outside any capture
            flag = torch.tensor([captured], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                exchange_fallback = exchange_fallback or "another rank failed to capture the all-reduce"
                exchange_mode, loss_exchange = "sync", None  # step_body reads loss_exchange at call time
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode=capture_mode()):
                    last = step_body()
    log("graph captured" if graph is not None else "eager mode")

    exchange = dfepe.dist.OverlappedLossExchange(L + 4, dev, depth=2) if exchange_mode == "overlap" else None

    def run_step():
        if graph is not None:
            graph.replay()
        else:
            step_body()
        if exchange is not None:
            exchange.exchange(state["loss_vec"])
        elif exchange_mode == "sync":
            dist.all_reduce(state["loss_vec"])

    def barrier():
        if exchange is not None:
            exchange.drain()  # every outstanding all-reduce is part of the timed region
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed spin-up in front of the W warm-up steps: a few ms of the same step so that clocks and caches are those of a running
    # job whatever W and K the caller picked (reported as "spinup_steps"; the timed region is exactly K steps of full work)
    # (synchronised every 20 steps: a host that has queued all of them and then sleeps 20 ms in the barrier below comes back slow --
    # with the driver's K = 20 the timed 2 ms then measured 0.104-0.114 ms per step on one box where the ten 20-step blocks behind
    # it, identical code after a 2 ms wait, all gave 0.103: scripts/bench_jitter.sh)
    for i in range(args.spinup):
        run_step()
        if i % 20 == 19 and os.environ.get("DFEPE_BENCH_SPINUP_SYNC", "1") != "0":
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        run_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank_ms = [round(elapsed * 1e3 / args.steps, 4)]
    rccl_world = None
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        per_rank_ms = [round(float(e.item()) * 1e3 / args.steps, 4) for e in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        rccl_world = dist.get_world_size()
    ms_per_step = elapsed * 1e3 / args.steps
    log("timed region done", ms_per_step, "ms/step")
    value = B_total * args.steps / elapsed

    # informational: spread of repeated short blocks (the timed region above is the contract number)
    block_stats = None
    if args.blocks > 0:
        bl = []
        for _ in range(args.blocks):
            barrier()
            tb = time.perf_counter()
            for _ in range(20):
                run_step()
            barrier()
            bl.append((time.perf_counter() - tb) * 1e3 / 20)
        block_stats = {"blocks": args.blocks, "steps_per_block": 20, "median_ms_per_step": round(statistics.median(bl), 4),
                       "min": round(min(bl), 4), "max": round(max(bl), 4)}

    result = None
    if rank == 0:
        extras = not args.no_extras
        # ---- roofline of the dominant kernel (the weighted 8-point fit), HIP events on the launch stream ---------------
        fit_call = lambda: dfepe.ops.w8pt_forward(m, None, w0, True, W, H, 0.5, True, kind == "train")
        kdur_us = event_time_us(fit_call)
        alg_bytes = B * (28 * N + 36)  # read 16N matches + 4N weights; write 36 F + 4N residual + 4N epi  
        achieved = alg_bytes / (kdur_us * 1e-6) / 1e9
        # which kernel of the family the library launches for this shape (csrc/w8pt16.hip: use_coop / use_pair2 / LEAN)
        if N <= dfepe._lib.W8PT16_MAX_N:
            kname = ("w8pt16_fwd_lean_kernel<raw> (one 16-lane row per pair, 234 registers: two wavefronts per SIMD)" if B >= 12288
                     else "w8pt16_fwd_kernel<raw> (one 16-lane row per pair, correspondences in registers)")
        elif N <= 2048 and B <= 1280:
            kname = "w8pt16_coop_fwd_kernel (a four-wavefront workgroup per pair, correspondences in registers)"
        elif N <= 2048 and B < 8192:
            kname = "w8pt16_pair2_fwd_kernel (two 16-lane rows of one wavefront per pair, correspondences re-read per phase)"
        else:
            kname = "w8pt16_fwd_kernel<0, raw> (one 16-lane row per pair, correspondences re-read per phase)"
        row_kernel = N <= dfepe._lib.W8PT16_MAX_N
        traffic = issue = rocprof_us = rocprof_src = None
        tpath = os.path.join(REPO, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(f"fit_fwd_B{B}_N{N}")
                rocprof_us, rocprof_src = tj.get(f"fit_fwd_rocprof_avg_us_B{B}_N{N}"), tj.get("fit_fwd_rocprof_source")
                valu = tj.get(f"fit_fwd_valu_insts_per_wave_B{B}_N{N}")
                if valu and row_kernel:
                    # what actually limits this kernel: four pairs per wavefront, ceil(B / 4 / 1024 SIMDs) wavefronts per SIMD,
                    # 4 issue cycles per wave64 VALU instruction; instruction count from the committed PMC pass
                    wps = -(-B // 4096)
                    cyc = wps * valu * 4.0
                    clk = tj.get("fit_fwd_sustained_clock_ghz")
                    # ~5.1 cycles of SIMD time per VALU instruction of this mix (scripts/ubench/lat.hip: fp64 FMA, DPP moves,
                    # conversions, selects; the same with one or four wavefronts per SIMD) at the sustained clock of the PMC pass
                    real_us = wps * valu * 5.1 / ((clk or 2.0) * 1e3)
                    issue = {"valu_insts_per_wave": valu, "waves_per_simd": wps, "cycles": round(cyc),
                             "floor_us_at_2.4GHz": round(cyc / 2.4e3, 2), "frac_of_kernel_time_at_2.4GHz": round(cyc / 2.4e3 / kdur_us, 3),
                             "sustained_clock_ghz": clk, "issue_bound_us_at_measured_rate": round(real_us, 2),
                             "frac_of_kernel_time_at_measured_rate": round(real_us / kdur_us, 3),
                             "source": "profiles/traffic.json (rocprofv3 --pmc SQ_INSTS_VALU, SQ_BUSY_CU_CYCLES); 5.1 cycles per instruction "
                                       "from scripts/ubench/lat.hip",
                             "note": "fraction of this kernel's own instruction stream, NOT a roofline; a lone wavefront per SIMD also pays for "
                                     "every scalar instruction and branch (~9 cycles each, scripts/ubench/lat2.hip): in round 3 a 14 % cut of "
                                     "the vector instructions moved the kernel by 3 %, 13 scalar instructions and 3 branches out of each "
                                     "multisection round by 8 %"}
            except Exception:
                traffic = None
        # the same launch against the ALU peaks ("report both"): arithmetic this algorithm performs per pair, counted from
        # the kernel: per correspondence ~250 flop (decode, Hartley, 36 fp64 moment FMAs, residual, epipolar residual),
        # per pair ~9 kflop for the eigen route (Householder tridiagonalisation ~2 k, 16 probes x ~11 rounds x 40 flop of multisection
        # ~7 k, twisted factorisation + back-transformation + rank-2 step ~1 k); about 70 % of it is fp64
        flops_pair = 250.0 * N + 9000.0
        alu_tf = B * flops_pair / (kdur_us * 1e-6) / 1e12
        alu = {"flops_per_pair": flops_pair, "achieved_TFLOPs": round(alu_tf, 2), "peak_fp64_vector_TFLOPs": 78.6, "frac_of_fp64_peak": round(alu_tf / 78.6, 4),
               "peak_fp32_vector_TFLOPs": 157.3, "frac_of_fp32_peak": round(alu_tf / 157.3, 4),
               "note": "useful arithmetic of the tridiagonal eigen route, not issued instructions: the kernel is bound by the in-order issue of ONE "
                       "wavefront per SIMD (see vector_issue) -- vector instructions, of which data movement inside the row (DPP), selects and "
                       "conversions are about half, plus the dependency, scalar and branch bubbles nothing else on the SIMD can fill"}
        roofline = {"bound": "hbm", "kernel": kname, "alu": alu, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                    "avg_kernel_us": round(kdur_us, 2), "algorithmic_bytes_per_launch": alg_bytes,
                    # the same kernel's average in the committed rocprofv3 kernel trace (another box, under the profiler): the two
                    # methods differ by a few per cent ("say so in the line") -- `frac` is the live HIP-event one
                    "rocprof_avg_kernel_us": rocprof_us, "rocprof_source": rocprof_src,
                    "frac_from_rocprof_avg": (round(alg_bytes / (rocprof_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 5) if rocprof_us else None),
                    "launches_per_step": L, "vector_issue": issue,
                    "traffic_note": "profiles/traffic.json: PMC FETCH_SIZE/WRITE_SIZE of this probe launch, which in the training configs "
                                    "also writes the 512-B save record per pair on top of the 28N+36 algorithmic bytes",
                    "method": "HIP events around a hipGraph replay of 50 back-to-back launches on the launch stream, median of 5 replays"}
        log("roofline probe done", kdur_us)

        acc = {}
        if kind == "train":
            acc = {"median_R_deg": round(float(last["R_deg"][-1].median().item()), 5),
                   "median_t_deg": round(float(last["t_deg"][-1].median().item()), 5)}
        elif kind == "pose":
            # geometric truth of the synthetic scene: camera motion = inverse of the scene pose
            Rt = state["Rt"].reshape(B, 3, 4)
            Rg = scene["R_gt"]
            tg = torch.nn.functional.normalize(scene["ts_cam"].reshape(B, 3), dim=1)
            ok = state["winner"] >= 0
            cosr = ((Rt[:, :, :3] @ Rg.transpose(1, 2)).diagonal(dim1=1, dim2=2).sum(1) - 1.0) / 2.0
            Rdeg = torch.rad2deg(torch.acos(cosr.clamp(-1, 1)))
            te = torch.nn.functional.normalize(Rt[:, :, 3], dim=1)
            tdeg = torch.rad2deg(torch.acos((te * tg).sum(1).clamp(-1, 1)))
            acc = {"median_R_deg_vs_scene_truth": round(float(Rdeg[ok].median().item()), 5),
                   "median_t_deg_vs_scene_truth": round(float(tdeg[ok].median().item()), 5), "pairs_with_a_valid_pose": int(ok.sum().item())}

        # ---- informational (config 3 only): variants and neighbours of the step that are never `value` -----------------
        layers_batched = recurrent_bwd = full_model = match_row = None
        if extras and kind == "train" and world == 1:
            # the backward a recurrent DeepFNet actually runs: the next estimator layer consumes residual and epi_res, so
            # w8pt_bwd also gets g_residual and g_epi (pass A over the correspondences); the fixed-logits step has g_F only
            Fo, res, epi, save, wout = dfepe.ops.w8pt_forward(m, None, scene["logits_layers"][0].contiguous(), True, W, H, 0.5, True, True, logits=True)
            gF, gR, gE = torch.randn_like(Fo), torch.randn_like(res), torch.randn_like(epi)
            gw = torch.empty_like(wout)
            t_f = event_time_us(lambda: dfepe.ops.w8pt_backward(m, None, wout, True, W, H, 0.5, save, Fo, gF, None, None, logits=True, out=gw))
            t_a = event_time_us(lambda: dfepe.ops.w8pt_backward(m, None, wout, True, W, H, 0.5, save, Fo, gF, gR, gE, logits=True, out=gw))
            recurrent_bwd = {"w8pt_bwd_us_gF_only": round(t_f, 2), "w8pt_bwd_us_gF_gResidual_gEpi": round(t_a, 2),
                             "note": "eager launches, HIP events; the timed step's logits are inputs, so its backward has g_F only; "
                                     "a recurrent DeepFNet adds (second - first) per layer"}
            if graph is not None:
                # all L fits of a layer stack in ONE grid (n_weight_sets = L).  Legal for this solver-only workload because
                # the per-layer logits are inputs; the recurrent DeepFNet cannot do it.
                def body_b():
                    o = dfepe.pipeline.hot_path_fused(m, logits, scene["Ks"], scene["pts1_virt_ori"], scene["pts2_virt_ori"], scene["qs_cam"],
                                                      scene["ts_cam"], scene["R_gt"], IMAGE_SIZE, clamp_at=0.02, qt=True, hw_T=hw_T,
                                                      balance_F=cfg["balance_F"], layers_batched=True)
                    return torch.autograd.grad(o["loss"], logits)[0]
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    gb = body_b()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                same = bool(torch.equal(gb, state["grad_logits"]))
                gr_b = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr_b, capture_error_mode=capture_mode()):
                    gb = body_b()
                for _ in range(args.warmup):
                    gr_b.replay()
                torch.cuda.synchronize()
                tb = time.perf_counter()
                for _ in range(args.steps):
                    gr_b.replay()
                torch.cuda.synchronize()
                tb = time.perf_counter() - tb
                layers_batched = {"ms_per_step": round(tb * 1e3 / args.steps, 4), "pairs_per_s": round(B * args.steps / tb, 1),
                                  "grad_bit_identical_to_value_run": same,
                                  "note": "all L layers' fits in one launch; only legal with fixed logits, not `value`"}
            log("variants done")

        # ---- the same step through the reference's OWN call sequence: compat.DeepFNet (estimator replaced
        #      by the same fixed per-layer logits) -> get_all_loss_DeepF -> get_Rt_loss -> the caller's clamp / balance lines ->
        #      backward (Train_model_pipeline.py:495-595).  Never `value`; it says what a train_good.py user gets of it. ------
        api_path = None
        if extras and kind == "train" and world == 1:
            try:
                api_path = measure_api_path(dfepe, scene, logits, L, cfg["balance_F"], args, B, ms_per_step, state.get("grad_logits"))
            except Exception as exc:  # informational only
                api_path = {"error": repr(exc)[:300]}
            log("api path done", api_path)

        # ---- CPU baseline: the oracle on a bounded sample of the same workload, on this box's host cores -----------------
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # contract: CPU baseline on rank 0 at N=1 only
            import numpy as np

            oracle = importlib.import_module("oracle.deepf_oracle")

            def _ferr(Fa, Fb):
                a = Fa.reshape(Fa.shape[0], -1).double(); b = Fb.reshape(Fb.shape[0], -1).double()
                a = a / a.norm(dim=1, keepdim=True); b = b / b.norm(dim=1, keepdim=True)
                sgn = torch.sign((a * b).sum(1, keepdim=True))
                return (a * sgn - b).norm(dim=1)

            threads = torch.get_num_threads()
            if kind == "train":
                Bc = min(args.cpu_sample, B)
                cpu_scene = {k: (v[:Bc] if k != "logits_layers" else v[:, :Bc]).cpu() for k, v in scene.items()}
                warm = {k: (v[:8] if k != "logits_layers" else v[:, :8]) for k, v in cpu_scene.items()}
                kw = dict(balance_F=cfg["balance_F"])
                oracle.hot_path_step(warm, IMAGE_SIZE, L, 0.02, qt=True, mode="loop", **kw)
                loop_t = []
                for _ in range(2):  # two passes: the spread between boxes was 40 ... 61 pairs/s in round 1
                    c0 = time.perf_counter()
                    ref = oracle.hot_path_step(cpu_scene, IMAGE_SIZE, L, 0.02, qt=True, mode="loop", **kw)
                    loop_t.append(time.perf_counter() - c0)
                # the batched restatement (one batched LAPACK call per fit; the pose loop stays per sample like the reference's)
                Bb = Bc
                cpu_b = {k: (v[:Bb] if k != "logits_layers" else v[:, :Bb]).cpu() for k, v in scene.items()}
                c0 = time.perf_counter()
                oracle.hot_path_step(cpu_b, IMAGE_SIZE, L, 0.02, qt=True, mode="batched", **kw)
                bt = time.perf_counter() - c0
                cpu = {"value": round(Bc / min(loop_t), 2), "unit": "pairs/s", "cores": threads, "kind": "port",
                       "sample": f"{Bc} pairs of the same workload (N={N}, depth={L}, fwd+bwd), best of 2 passes ({loop_t[0]:.1f} s, {loop_t[1]:.1f} s); "
                                 "oracle/deepf_oracle.py hot_path_step(mode='loop') = per-sample torch SVD + per-sample pose loop like the reference",
                       "batched_restatement": {"value": round(Bb / bt, 2), "unit": "pairs/s", "sample": f"{Bb} pairs, one pass, {bt:.1f} s; "
                                               "mode='batched': batched LAPACK SVDs, per-sample pose loop"}}
                ours_F = last["F_layers"][-1][:Bc].cpu()
                e32 = _ferr(ours_F, ref["outs"]["out_layers"][-1].detach())
                ref64 = oracle.hot_path_step({k: v.double() for k, v in cpu_scene.items()}, IMAGE_SIZE, L, 0.02, qt=False, mode="batched", backward=False)
                e64 = _ferr(ours_F, ref64["outs"]["out_layers"][-1])
                acc.update({"cpu_ref_median_R_deg_sample": round(float(np.median(ref["pose"]["R_deg"][-1])), 5),
                            "cpu_ref_median_t_deg_sample": round(float(np.median(ref["pose"]["t_deg"][-1])), 5),
                            "gpu_median_R_deg_sample": round(float(np.median(last["R_deg"][-1][:Bc].cpu().numpy())), 5),
                            "gpu_median_t_deg_sample": round(float(np.median(last["t_deg"][-1][:Bc].cpu().numpy())), 5),
                            "F_fro_err_vs_fp64_oracle_sample": {"max": float(e64.max()), "median": float(e64.median())},
                            "F_fro_err_vs_fp32_reference_shaped_sample": {"max": float(e32.max()), "median": float(e32.median())}})
            else:
                Bc = min(args.cpu_sample if kind == "fit" else 48, B)
                mc, wc = m[:Bc].cpu(), w0[:Bc].cpu()
                Kc = scene["Ks"][:Bc].cpu()

                def cpu_pass():
                    p1, p2, T = oracle.normalize_hw(mc, IMAGE_SIZE)
                    out, _, _ = oracle.fit_forward(p1, p2, wc.unsqueeze(1), mode="loop")
                    E = Kc.transpose(1, 2) @ T.transpose(1, 2) @ out @ T @ Kc
                    if kind == "pose":
                        for b_ in range(Bc):
                            oracle.cheirality_select(E[b_], Kc[b_].numpy(), mc[b_, :, :2].numpy(), mc[b_, :, 2:].numpy(), 50.0)
                    return out
                cpu_pass()
                c0 = time.perf_counter()
                outc = cpu_pass()
                cdt = time.perf_counter() - c0
                cpu = {"value": round(Bc / cdt, 2), "unit": "pairs/s", "cores": threads, "kind": "port",
                       "sample": f"{Bc} pairs of the same workload (N={N}), one pass, {cdt:.1f} s; oracle fit_forward(mode='loop')"
                                 + (" + cheirality_select per pair (DLT stand-in for cv2.triangulatePoints)" if kind == "pose" else "")}
                e32 = _ferr(last["F_layers"][-1][:Bc].cpu(), outc)
                acc["F_fro_err_vs_fp32_reference_shaped_sample"] = {"max": float(e32.max()), "median": float(e32.median())}
            log("cpu baseline done", cpu)

        # ---- secondary, informational: the whole DeepFNet step (estimator on the bf16 matrix cores with split operands and the
        #      InstanceNorm/LeakyReLU epilogue, solver, F-loss, qt loss, backward to the estimator parameters) -----------
        if extras and not args.no_full_model and world == 1 and kind == "train" and args.config == 3:
            try:
                net = dfepe.compat.DeepFNet.DeepFNet(depth=L, image_size=IMAGE_SIZE, if_quality=False).to(dev)
                dfepe.synth.fill_params_deterministic(net, 1)
                tgu = dfepe.compat.train_good_utils
                lp = {"depth": L, "clamp_at": 0.02, "if_tri_depth": False, "if_sample_loss": False, "topK": 8, "matches_good_unique_nums": None}
                batch = {"matches_xy_ori": m, "matches_good_unique_nums": None, "t_scene_scale": None}

                def full_step():
                    net.zero_grad(set_to_none=True)
                    outs = net(batch)
                    losses, _, _, _, _, _, E_layers = tgu.get_all_loss_DeepF(outs, scene["pts1_virt_ori"], scene["pts2_virt_ori"], scene["Ks"], lp, get_residual_summaries=False)
                    rt = tgu.get_Rt_loss(E_layers, None, None, None, scene["delta_Rtijs_4_4"], scene["qs_cam"], scene["ts_cam"], device=dev)
                    lq = torch.clamp(torch.stack(rt["q_l2_error_layers_list"]), 0, 0.1).mean()
                    lt = torch.clamp(torch.stack(rt["t_l2_error_layers_list"]), 0, 0.5).mean()
                    (losses["loss_F"] + lq + 0.1 * lt).backward()

                for _ in range(2):
                    full_step()
                torch.cuda.synchronize()
                f0 = time.perf_counter()
                nfull = 3
                for _ in range(nfull):
                    full_step()
                torch.cuda.synchronize()
                fdt = (time.perf_counter() - f0) / nfull
                full_model = {"value": round(B / fdt, 1), "unit": "pairs/s", "ms_per_step": round(fdt * 1e3, 2),
                              "what": "compat.DeepFNet (seeded random weights) forward + F-loss + qt loss + backward to the estimator parameters; the estimator runs "
                                      "on the 16-bit matrix cores with fp32-accurate split operands (csrc/est_gemm.hip, SURVEY row f-1); not part of `value`"}
                # the same model at a loader-sized batch, where the step is launch-bound: the reference's eager sequence against
                # compat.CapturedStep (the whole step -- estimators, solver, losses, backward -- as one replayed hipGraph)
                keys = ("matches_xy_ori", "pts1_virt_ori", "pts2_virt_ori", "Ks", "delta_Rtijs_4_4", "qs_cam", "ts_cam")

                def fwd_loss(b):
                    outs = net({"matches_xy_ori": b["matches_xy_ori"], "matches_good_unique_nums": None, "t_scene_scale": None})
                    losses, _, _, _, _, _, E_layers = tgu.get_all_loss_DeepF(outs, b["pts1_virt_ori"], b["pts2_virt_ori"], b["Ks"], lp, get_residual_summaries=False)
                    rt = tgu.get_Rt_loss(E_layers, None, None, None, b["delta_Rtijs_4_4"], b["qs_cam"], b["ts_cam"], device=dev)
                    lq = torch.clamp(torch.stack(rt["q_l2_error_layers_list"]), 0, 0.1).mean()
                    lt = torch.clamp(torch.stack(rt["t_l2_error_layers_list"]), 0, 0.5).mean()
                    return losses["loss_F"] + lq + 0.1 * lt, None

                def eager_small(b):
                    net.zero_grad(set_to_none=True)
                    fwd_loss(b)[0].backward()

                def small_batch(Bs, Ns=None):
                    if Ns is None or Ns == N:
                        small = [{k: scene[k][i * Bs:(i + 1) * Bs].contiguous() for k in keys} for i in range(2)]
                    else:  # another number of points per pair: scenes of its own (the same generator and seeds family)
                        small = []
                        for i in range(2):
                            sc_ = dfepe.synth.make_scene(Bs, Ns, seed=2000 + i, outlier_ratio=outl, noise_px=0.5)
                            small.append({k: sc_[k].to(dev) for k in keys})
                    for k in range(3):
                        eager_small(small[k & 1])
                    torch.cuda.synchronize()
                    f0 = time.perf_counter()
                    for k in range(10):
                        eager_small(small[k & 1])
                    torch.cuda.synchronize()
                    eager_small_ms = (time.perf_counter() - f0) * 1e2
                    helper = dfepe.compat.CapturedStep(fwd_loss, net, warmup=2)
                    for k in range(6):
                        helper(small[k & 1])
                    torch.cuda.synchronize()
                    f0 = time.perf_counter()
                    for k in range(20):
                        helper(small[k & 1])
                    torch.cuda.synchronize()
                    cap_ms = (time.perf_counter() - f0) * 50.0
                    nl = count_launches(lambda: eager_small(small[0]))
                    return {"B": Bs, "N": Ns or N, "eager_ms_per_step": round(eager_small_ms, 3),
                            "captured_step_ms_per_step": round(cap_ms, 3),
                            "captures": helper.n_captures, "replays": helper.n_replays, "graphs_rejected_by_self_check": helper.n_rejected,
                            "kernel_launches_per_step": None if nl is None else nl["total"],
                            "note": "compat.CapturedStep(forward_and_loss, net): copy-in of a fresh batch + one hipGraph replay per step (no optimizer in either figure)"}

                full_model["small_batch"] = small_batch(64)
                full_model["reference_batch"] = small_batch(8)  # the reference's own configurations train with 4-32 pairs per batch ...
                # ... and 1000-2000 points per pair (deepFEPE/configs/kitti_corr_baseline.yaml:12-13: good_num 1000, batch_size 8)
                full_model["reference_batch_n1000"] = small_batch(8, 1000)
                del net
            except Exception as e:  # never let the secondary measurement break the contract line
                full_model = {"error": repr(e)[:200]} if full_model is None else dict(full_model, small_batch={"error": repr(e)[:200]})

        # ---- informational: the upstream match-construction row, fp32-MFMA two-way descriptor matching ----
        if extras and world == 1 and args.config == 3:
            try:
                gm = torch.Generator().manual_seed(0)
                Bm_, Nm_, Dm_ = 64, 1024, 256
                da = torch.nn.functional.normalize(torch.randn(Bm_, Nm_, Dm_, generator=gm), dim=2)
                db = torch.nn.functional.normalize(da[:, torch.randperm(Nm_, generator=gm)] + 0.05 * torch.randn(Bm_, Nm_, Dm_, generator=gm), dim=2)
                da, db = da.to(dev), db.to(dev)
                tm_ = event_time_us(lambda: dfepe.ops.nn_match_two_way(da, db, 0.7), reps=20, rounds=3, warm=3, graph=False) * 1e-6
                cntm = dfepe.ops.nn_match_two_way(da, db, 0.7)[3]
                fl = 2.0 * Bm_ * Nm_ * Nm_ * Dm_
                match_row = {"workload": f"two-way nearest-neighbour matching of {Bm_} pairs x {Nm_} x {Nm_} descriptors (D={Dm_}, fp32)",
                             "pairs_per_s": round(Bm_ / tm_, 1), "ms": round(tm_ * 1e3, 4),
                             "roofline": {"bound": "mfma", "achieved": round(fl / tm_ / 1e12, 2), "peak": 157.3, "unit": "TFLOP/s",
                                          "frac": round(fl / tm_ / 157.3e12, 4), "dtype": "f32 (v_mfma_f32_32x32x2_f32)"},
                             "mean_matches": float(cntm.float().mean().item()), "note": "upstream of the solver, not part of `value`"}
            except Exception as exc:  # informational only
                match_row = {"error": repr(exc)}
            log("match-construction row done", match_row)

        result = {
            "metric": METRIC if args.config == 3 else f"image-pairs/sec, BASELINE config {args.config}",
            "value": round(value, 1),
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup, "spinup_steps": args.spinup,
            "ms_per_step": round(ms_per_step, 4),
            "ms_per_step_per_rank": per_rank_ms,
            "rccl_world_size": rccl_world,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            # what the path computes in: the solver (moments, tridiagonal eigen-solve, rank-2 step, pose decomposition) is fp64;
            # tensors cross the boundary as fp32 and the per-point epipolar / F-loss terms are fp32 like the reference's
            "dtype": "f64 solver arithmetic, f32 I/O and epipolar terms",
            "data": "synthetic",
            "config": {"workload": cfg["what"].format(B=B_cfg, N=N, L=L), "baseline_config": args.config, "B_per_gpu": B, "B_total": B_total,
                       "N": N, "depth": L, "outlier_ratio": outl, "parallelism": f"dp{world}", "hipgraph": graph is not None,
                       "launches_per_step": ((2 * L + 2) if (args.no_defer_head or exchange_mode == "branch") else (2 * L + 1)) if kind == "train" else 2,
                       "loss_head": ("a launch on the exchange stream, followed by the all-reduce: a branch of the step's graph parallel to the backward fits"
                                     if exchange_mode == "branch" else "a launch of its own" if args.no_defer_head else
                                     "batch sums of the loss finished in spare wavefronts of the first backward launch (defer_loss_head)"),
                       "loss_exchange": {"graph": "all_reduce(SUM) of L+4 doubles captured as the last node of the step's hipGraph",
                                         "branch": "all_reduce(SUM) of L+4 doubles captured in the step's hipGraph as a branch parallel to the backward",
                                         "sync": "all_reduce(SUM) of L+4 doubles in stream order after every step",
                                         "overlap": "double-buffered asynchronous all_reduce (dist.OverlappedLossExchange)",
                                         "none": None}[exchange_mode],
                       "loss_exchange_fallback": exchange_fallback,
                       "env": {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "DEBUG_CLR_GRAPH_PACKET_CAPTURE", "DFEPE_BENCH_EXCHANGE", "NCCL_DEBUG")},
                       # whether the hipGraph packet-capture workaround took effect in THIS process (set before the HIP runtime initialised)
                       "hip_graph_packet_capture_off": bool(getattr(dfepe, "HIP_GRAPH_PACKET_CAPTURE_OFF", False))},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "accuracy": acc,
            "block_stats": block_stats,
            "api_path": api_path,
            "recurrent_backward": recurrent_bwd,
            "full_model": full_model,
            "layers_batched": layers_batched,
            "match_construction": match_row,
        }
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
    if dist is not None:
        dist.barrier()
        # the captured graphs hold the communicator's collective as a node: they go first, and the device is idle, before the process
        # group is torn down (a watchdog thread that still sees work on a destroyed communicator aborts the process)
        graph = None
        state.clear()
        import gc

        gc.collect()
        torch.cuda.synchronize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
