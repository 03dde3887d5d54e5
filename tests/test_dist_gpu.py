"""What the data-parallel path rests on, checked on ONE GPU (the driver runs 8-GPU nodes itself):

* ``grad_pairs``: a rank that owns a shard of the batch must produce exactly the rows of the global-batch gradient;
* the step bench.py times (captured hipGraph, deferred loss head, grad_pairs) is bit-identical to the eager, undeferred step
  the full-size parity test compares with the oracle;
* a real RCCL group (one rank) behind a graph replay through dist.reduce_losses / OverlappedLossExchange;
* ``python bench.py --gpus N`` launches its own ranks (the form the driver uses for N = 1, deepFEPE/train_good.py:311-312 is
  the reference's single-process counterpart).
GPU box only."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
IMAGE_SIZE = [376, 1241, 3]
DEV = "cuda:0"
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fused(dfepe, sc, L, **kw):
    logits = sc["logits_layers"][:L].detach().clone().requires_grad_(True)
    o = dfepe.pipeline.hot_path_fused(sc["matches_xy_ori"], logits, sc["Ks"], sc["pts1_virt_ori"], sc["pts2_virt_ori"], sc["qs_cam"],
                                      sc["ts_cam"], sc["R_gt"], IMAGE_SIZE, 0.02, True, **kw)
    o["loss"].backward()
    torch.cuda.synchronize()
    o["grad_logits"] = logits.grad
    return o


@pytest.mark.parametrize("B,cut,balance_F", [(4096, 2048, 1.0), (301, 100, 0.0)])
def test_grad_pairs_shards_reproduce_the_global_batch_gradient(dfepe, B, cut, balance_F):
    """Two ranks' shards (here: two calls on one GPU) run with grad_pairs = global batch give the rows of the full-batch
    d loss / d logits: the batch means of train_good_utils.py:340-364 / Train_model_pipeline.py:580-586 divide by the GLOBAL
    number of pairs.  Per-pair arithmetic does not depend on the batch, so the rows agree bit for bit; the loss scalars combine
    through dist.reduce_losses from the packed sums."""
    L = 5
    full = dfepe.pipeline.scene_to_device(dfepe.synth.make_scene(B, 100, seed=91, outlier_ratio=0.3, depth_layers=L), DEV)
    whole = _fused(dfepe, full, L, balance_F=balance_F)
    parts, packed = [], []
    for a, b in ((0, cut), (cut, B)):
        shard = {k: (v[:, a:b] if k == "logits_layers" else v[a:b]).contiguous() for k, v in full.items()}
        o = _fused(dfepe, shard, L, balance_F=balance_F, grad_pairs=B)
        parts.append(o["grad_logits"])
        packed.append(o["packed"].clone())
    g = torch.cat(parts, dim=1)
    assert torch.equal(g, whole["grad_logits"])
    # and without grad_pairs the shard gradient is the LOCAL mean's: larger by B / shard size
    shard = {k: (v[:, :cut] if k == "logits_layers" else v[:cut]).contiguous() for k, v in full.items()}
    loc = _fused(dfepe, shard, L, balance_F=balance_F)
    ratio = (loc["grad_logits"].double().norm() / parts[0].double().norm()).item()
    assert abs(ratio - B / cut) < 1e-4 * B / cut
    # the loss scalars of the global batch from the two packed vectors (what the all-reduce sums)
    both = packed[0] + packed[1]
    both[L + 3] = packed[0][L + 3]  # M travels with the vector but is not summed (dist.reduce_losses keeps it out of the all-reduce)
    red = dfepe.dist.reduce_losses(both, L, 1.0, 0.1)
    np.testing.assert_allclose(red["loss_F"].item(), whole["loss_F"].item(), rtol=2e-6)
    np.testing.assert_allclose(red["loss_qt"].item(), whole["loss_qt"].item(), rtol=2e-6)
    assert int(red["n_pairs"].item()) == B


@pytest.mark.parametrize("balance_F,outl", [(1.0, 0.2), (0.0, 0.4)])
def test_captured_deferred_step_is_the_eager_step_bit_for_bit_at_bench_size(dfepe, balance_F, outl):
    """bench.py's timed step -- hot_path_fused(grad_pairs=B_total, defer_loss_head=True) captured in a hipGraph and replayed --
    against the eager, undeferred default that tests/test_fullsize_gpu.py compares with the fp64 oracle, at B=4096, N=100,
    depth 5 (configs 3 and 4): every output and d loss / d logits identical, also after repeated replays."""
    B, N, L = 4096, 100, 5
    sc = dfepe.pipeline.scene_to_device(dfepe.synth.make_scene(B, N, seed=2024, outlier_ratio=outl, noise_px=0.5, depth_layers=L), DEV)
    eager = _fused(dfepe, sc, L, balance_F=balance_F)
    H, W = float(IMAGE_SIZE[0]), float(IMAGE_SIZE[1])
    hw_T = torch.tensor([[2.0 / W, 0.0, -1.0], [0.0, 2.0 / H, -1.0], [0.0, 0.0, 1.0]], device=DEV)
    logits = sc["logits_layers"][:L].clone().requires_grad_(True)
    state = {}

    def step_body():  # bench.py: step_body of kind == "train"
        out = dfepe.pipeline.hot_path_fused(sc["matches_xy_ori"], logits, sc["Ks"], sc["pts1_virt_ori"], sc["pts2_virt_ori"], sc["qs_cam"],
                                            sc["ts_cam"], sc["R_gt"], IMAGE_SIZE, clamp_at=0.02, qt=True, hw_T=hw_T, balance_F=balance_F,
                                            grad_pairs=B, defer_loss_head=True)
        state["g"], = torch.autograd.grad(out["loss"], logits)
        return out

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step_body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode=dfepe.dist.graph_capture_mode()):
        out = step_body()
    for rep in range(3):
        for t in (out["loss"], out["packed"], state["g"], out["F_layers"]):  # poison: a replay must rewrite everything
            t.detach().fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        for k in ("loss", "loss_F", "loss_qt", "loss_layers", "packed", "loss_sum", "q_l2", "t_l2", "R_deg", "t_deg", "F_layers", "E_layers"):
            assert torch.equal(out[k], eager[k]), (k, rep)
        assert torch.equal(state["g"], eager["grad_logits"]), rep


def _body_one_rank_rccl_group_behind_a_graph_replay(dfepe):
    """A real RCCL communicator (world size 1: one GPU under this lease) on the path bench.py --gpus N runs per step: graph replay
    of the fused step -> all-reduce of the packed (L+4)-double vector -> dist.reduce_losses, in-stream and through the
    double-buffered OverlappedLossExchange; the reduced means are the step's own batch means."""
    import torch.distributed as dist

    L, B = 5, 512
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        assert dist.get_backend() == "nccl"
        sc = dfepe.pipeline.scene_to_device(dfepe.synth.make_scene(B, 100, seed=5, outlier_ratio=0.4, depth_layers=L), DEV)
        eager = _fused(dfepe, sc, L)
        logits = sc["logits_layers"][:L].clone().requires_grad_(True)
        state = {}
        H, W = float(IMAGE_SIZE[0]), float(IMAGE_SIZE[1])
        hw_T = torch.tensor([[2.0 / W, 0.0, -1.0], [0.0, 2.0 / H, -1.0], [0.0, 0.0, 1.0]], device=DEV)  # a host copy: not capturable

        def step_body():
            out = dfepe.pipeline.hot_path_fused(sc["matches_xy_ori"], logits, sc["Ks"], sc["pts1_virt_ori"], sc["pts2_virt_ori"], sc["qs_cam"],
                                                sc["ts_cam"], sc["R_gt"], IMAGE_SIZE, 0.02, True, hw_T=hw_T, grad_pairs=B * dist.get_world_size(),
                                                defer_loss_head=True)
            state["g"], = torch.autograd.grad(out["loss"], logits)
            return out

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step_body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode=dfepe.dist.graph_capture_mode()):
            out = step_body()
        ex = dfepe.dist.OverlappedLossExchange(L + 4, torch.device(DEV), depth=2)
        for _ in range(4):
            graph.replay()
            ex.exchange(out["packed"])
        last = ex.drain().clone()
        graph.replay()
        red = dfepe.dist.reduce_losses(out["packed"].clone(), L, 1.0, 0.1)  # in-stream all-reduce
        dist.barrier()
        torch.cuda.synchronize()
        assert torch.equal(last, out["packed"])
        np.testing.assert_allclose(red["loss_F"].item(), eager["loss_F"].item(), rtol=2e-6)
        np.testing.assert_allclose(red["loss_qt"].item(), eager["loss_qt"].item(), rtol=2e-6)
        np.testing.assert_allclose(red["loss_layers"].cpu().numpy(), eager["loss_layers"].double().cpu().numpy(), rtol=2e-6)
        assert torch.equal(state["g"], eager["grad_logits"])
        print(_BODY_OK, flush=True)
    finally:
        dist.destroy_process_group()


def _body_all_reduce_captured_in_the_steps_graph(dfepe):
    """the loss all-reduce as part of the captured step, with a real (one-rank) RCCL communicator -- as the last
    node of the graph (bench.py's default) and as a branch tail -> [loss head -> all_reduce(packed)] || [L x w8pt_bwd] joined at the
    end of the backward; eager and replayed: the packed vector, every batch scalar and d loss / d logits are bit-identical to the
    plain step.  The in-order node must cost (next to) nothing; the branch is measured and printed (+33 us on this stack: cross-stream
    edges of a hipGraph cost more than the collective, scripts/exchange_probe.py)."""
    import time

    import torch.distributed as dist

    L, B = 5, 4096
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        sc = dfepe.pipeline.scene_to_device(dfepe.synth.make_scene(B, 100, seed=6, outlier_ratio=0.2, depth_layers=L), DEV)
        eager = _fused(dfepe, sc, L)
        H, W = float(IMAGE_SIZE[0]), float(IMAGE_SIZE[1])
        hw_T = torch.tensor([[2.0 / W, 0.0, -1.0], [0.0, 2.0 / H, -1.0], [0.0, 0.0, 1.0]], device=DEV)
        logits = sc["logits_layers"][:L].clone().requires_grad_(True)
        calls = {"n": 0}

        def exchange(p):
            calls["n"] += 1
            dist.all_reduce(p)

        def make_step(with_exchange, state):
            def step_body():
                out = dfepe.pipeline.hot_path_fused(sc["matches_xy_ori"], logits, sc["Ks"], sc["pts1_virt_ori"], sc["pts2_virt_ori"], sc["qs_cam"],
                                                    sc["ts_cam"], sc["R_gt"], IMAGE_SIZE, 0.02, True, hw_T=hw_T, grad_pairs=B,
                                                    defer_loss_head=True, loss_exchange=exchange if with_exchange else None,
                                                    exchange_branch=with_exchange == "branch")
                state["g"], = torch.autograd.grad(out["loss"], logits, grad_outputs=state.setdefault("seed", torch.ones_like(out["loss"])))
                return out
            return step_body

        times = {}
        for with_exchange in (False, "in order", "branch"):
            state = {}
            body = make_step(with_exchange, state)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                out = body()  # eager: the branch runs on the exchange stream, the backward joins it
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            for k in ("loss", "loss_F", "loss_qt", "loss_layers", "packed", "q_l2", "F_layers"):
                assert torch.equal(out[k], eager[k]), (with_exchange, k)
            assert torch.equal(state["g"], eager["grad_logits"])
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode=dfepe.dist.graph_capture_mode()):
                out = body()
            for rep in range(3):
                for t in (out["loss"], out["packed"], state["g"]):
                    t.detach().fill_(float("nan"))
                graph.replay()
                torch.cuda.synchronize()
                for k in ("loss", "loss_F", "loss_qt", "loss_layers", "packed", "q_l2", "F_layers"):
                    assert torch.equal(out[k], eager[k]), (with_exchange, k, rep)  # world size 1: the sum over ranks is the rank's own
                assert torch.equal(state["g"], eager["grad_logits"]), (with_exchange, rep)
            for _ in range(200):
                graph.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(300):
                graph.replay()
            torch.cuda.synchronize()
            times[with_exchange] = (time.perf_counter() - t0) / 300 * 1e6
        assert calls["n"] >= 4  # each variant once eagerly, once while capturing
        print(f"captured step: {times[False]:.1f} us without exchange, {times['in order']:.1f} us with the all-reduce as its last node, "
              f"{times['branch']:.1f} us with the all-reduce branch")
        assert times["in order"] < times[False] + 20.0  # measured: +-0.5 us (the bound is loose: a shared box must not fail the suite)
        print(_BODY_OK, flush=True)
    finally:
        dist.destroy_process_group()


# The two tests that hold a real RCCL communicator run their bodies in a CHILD interpreter (python tests/test_dist_gpu.py <body>).
# Round 6 saw the process abort in two whole-suite runs (gpurun_out/r6t, r6v): the communicator's watchdog thread polls the events of
# earlier collectives, and under torch.cuda.graph's default capture_error_mode = "global" that hipEventQuery is illegal while ANY thread
# captures -- "HIP error: operation not permitted when stream is capturing", std::terminate.  A race (the poll has to fall inside a
# capture), so the file passed on its own.  The captures here, in bench.py and in compat.CapturedStep now use
# dist.graph_capture_mode() ("thread_local" once a process group exists); the child keeps an abort of that kind, should one remain,
# from taking the rest of the suite with it.  It prints _BODY_OK after its last assertion.
_BODY_OK = "DFEPE_DIST_BODY_OK"


def _run_body_in_child(name):
    env = dict(os.environ)
    env.pop("RANK", None), env.pop("LOCAL_RANK", None), env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), name], cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    sys.stdout.write(r.stdout[-4000:])
    if _BODY_OK not in r.stdout:
        raise AssertionError(f"{name} failed in the child interpreter (rc {r.returncode}):\n{r.stdout[-6000:]}\n{r.stderr[-6000:]}")
    if r.returncode != 0:
        print(f"[test_dist_gpu] {name}: every assertion passed, the child then left with rc {r.returncode} in its RCCL teardown")


@pytest.mark.timeout(1000)
def test_one_rank_rccl_group_behind_a_graph_replay():
    _run_body_in_child("_body_one_rank_rccl_group_behind_a_graph_replay")


@pytest.mark.timeout(1000)
def test_all_reduce_captured_in_the_steps_graph():
    _run_body_in_child("_body_all_reduce_captured_in_the_steps_graph")


@pytest.mark.timeout(600)
@pytest.mark.parametrize("flags", [["--gpus", "1", "--launcher", "torchrun"], ["--gpus", "1", "--force-dist"]])
def test_bench_launches_its_own_ranks(flags):
    """`python bench.py --gpus N` without a launcher re-executes under torch.distributed.run (here N = 1 forced through the same
    path) and prints ONE JSON line from rank 0 that names the RCCL world it ran in."""
    env = dict(os.environ)
    env.pop("RANK", None), env.pop("LOCAL_RANK", None), env.pop("WORLD_SIZE", None)
    env["MASTER_PORT"] = str(_free_port())
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *flags, "--steps", "20", "--warmup", "3", "--batch", "512", "--no-extras",
                        "--no-cpu-baseline", "--blocks", "0"], cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=580)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["rccl_world_size"] == 1 and len(j["ms_per_step_per_rank"]) == 1
    assert j["value"] > 0 and j["config"]["B_total"] == 512


if __name__ == "__main__":  # child of _run_body_in_child
    import importlib

    os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")  # like tests/conftest.py, before the HIP runtime starts
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    globals()[sys.argv[1]](importlib.import_module("pytorch-deepfepe_amd"))
