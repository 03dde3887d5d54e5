"""Multi-process (gloo, world_size 2) check of the data-parallel host logic: shards partition the batch, and the
all-reduced loss sums reproduce the single-process global means.  The per-pair arithmetic on each rank is done by
the CPU oracle here (there is no GPU in this container); the thing under test is pytorch-deepfepe_amd/dist.py."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
IMAGE_SIZE = [376, 1241, 3]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _local_sums(dfepe, oracle, scene, depth, M):
    outs = oracle.deepf_forward(scene["matches_xy_ori"], IMAGE_SIZE, depth, logits_layers=scene["logits_layers"][:depth])
    losses, _, _, E_layers = oracle.f_loss(outs, scene["pts1_virt_ori"], scene["pts2_virt_ori"], scene["Ks"], depth, 0.02)
    pose = oracle.rt_loss(E_layers, scene["delta_Rtijs_4_4"], scene["qs_cam"], scene["ts_cam"])
    loss_sum = losses["loss_per_pair"].detach() * M
    return loss_sum, pose["q_l2"].detach(), pose["t_l2"].detach()


def _worker(rank, world, port, B, N, depth, q):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dfepe = importlib.import_module("pytorch-deepfepe_amd")
    oracle = importlib.import_module("oracle.deepf_oracle")
    full = dfepe.synth.make_scene(B, N, seed=77, outlier_ratio=0.2, depth_layers=depth, dtype=torch.float64)
    mine = dfepe.dist.shard_scene(full, rank, world)
    a, b = dfepe.dist.shard_range(B, rank, world)
    assert mine["matches_xy_ori"].shape[0] == b - a and mine["logits_layers"].shape[1] == b - a
    M = full["pts1_virt_ori"].shape[1]
    loss_sum, q_l2, t_l2 = _local_sums(dfepe, oracle, mine, depth, M)
    packed = dfepe.dist.pack_loss_sums(loss_sum, M, q_l2, t_l2, 0.1, 0.5)
    red = dfepe.dist.reduce_losses(packed, depth, 1.0, 0.1)
    if rank == 0:
        q.put({k: v.numpy() for k, v in red.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_every_batch():
    sys.path.insert(0, REPO)
    d = importlib.import_module("pytorch-deepfepe_amd")
    for B in (0, 1, 7, 8, 4096, 32768, 10):
        for world in (1, 2, 3, 8):
            covered = []
            for r in range(world):
                a, b = d.dist.shard_range(B, r, world)
                assert 0 <= a <= b <= B
                covered += list(range(a, b))
            assert covered == list(range(B))
            sizes = [d.dist.shard_range(B, r, world)[1] - d.dist.shard_range(B, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        d.dist.shard_range(8, 2, 2)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("B,world", [(7, 2), (19, 8)])  # uneven shards; 8 ranks = the node bench.py --gpus 8 runs on
def test_loss_reduction_over_ranks_matches_single_process(B, world):
    sys.path.insert(0, REPO)
    dfepe = importlib.import_module("pytorch-deepfepe_amd")
    oracle = importlib.import_module("oracle.deepf_oracle")
    N, depth = 40, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, N, depth, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=480)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    full = dfepe.synth.make_scene(B, N, seed=77, outlier_ratio=0.2, depth_layers=depth, dtype=torch.float64)
    ref = oracle.hot_path_step(full, IMAGE_SIZE, depth, 0.02, qt=True, mode="batched", backward=False)
    np.testing.assert_allclose(got["loss_layers"], torch.stack(ref["losses"]["loss_layers"]).numpy(), rtol=1e-10)
    np.testing.assert_allclose(got["loss_F"], ref["losses"]["loss_F"].numpy(), rtol=1e-10)
    lqt = oracle.qt_training_loss(ref["pose"]["q_l2"], ref["pose"]["t_l2"], 0.1, 0.5, 1.0, 0.1)
    np.testing.assert_allclose(got["loss_qt"], lqt.numpy(), rtol=1e-10)
    assert got["n_pairs"] == B


def _exchange_worker(rank, world, port, q):
    import torch.distributed as dist

    sys.path.insert(0, REPO)
    dfepe = importlib.import_module("pytorch-deepfepe_amd")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ex = dfepe.dist.OverlappedLossExchange(5, "cpu", depth=2)
    packed = torch.zeros(5, dtype=torch.float64)
    seen = []
    for step in range(7):
        packed[:] = torch.arange(5, dtype=torch.float64) * (rank + 1) + step   # the producer overwrites its buffer every step
        ex.exchange(packed)
        if step % 3 == 2:
            seen.append(ex.drain().clone())
    final = ex.drain().clone()
    dist.destroy_process_group()
    if rank == 0:
        q.put({"seen": [t.numpy() for t in seen], "final": final.numpy()})


@pytest.mark.timeout(300)
def test_overlapped_loss_exchange_two_ranks():
    """The double-buffered asynchronous exchange bench.py uses between hipGraph replays: every step's vector is summed
    over the ranks although the producer buffer is overwritten right away and buffers are recycled every second step."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    base = np.arange(5, dtype=np.float64)
    expect = lambda step: base * 1 + step + base * 2 + step
    np.testing.assert_allclose(got["seen"][0], expect(2))
    np.testing.assert_allclose(got["seen"][1], expect(5))
    np.testing.assert_allclose(got["final"], expect(6))


def test_overlapped_loss_exchange_without_process_group():
    dfepe = importlib.import_module("pytorch-deepfepe_amd")
    ex = dfepe.dist.OverlappedLossExchange(3, "cpu")
    assert ex.drain() is None
    ex.exchange(torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64))
    np.testing.assert_allclose(ex.drain().numpy(), [1.0, 2.0, 3.0])


def _capture_mode_worker(port, q):
    sys.path.insert(0, REPO)
    dfepe = importlib.import_module("pytorch-deepfepe_amd")
    bench = importlib.import_module("bench")
    before = (dfepe.dist.graph_capture_mode(), bench.capture_mode())
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    during = (dfepe.dist.graph_capture_mode(), bench.capture_mode())
    dist.destroy_process_group()
    q.put((before, during, (dfepe.dist.graph_capture_mode(), bench.capture_mode())))


def test_graph_capture_mode_follows_the_process_group():
    """Round 6: a hipGraph capture next to a process group must be thread-local -- the group's watchdog thread polls events, which the
    default global mode forbids while any thread captures (the process aborted on the GPU box) -- and torch's default otherwise.  The
    library's helper and bench.py's agree."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_capture_mode_worker, args=(_free_port(), q))
    p.start()
    before, during, after = q.get(timeout=120)
    p.join(60)
    assert before == ("global", "global") and during == ("thread_local", "thread_local") and after == ("global", "global")


def test_bench_self_launch_command_is_the_drivers_invocation():
    """`python bench.py --gpus N` (no RANK in the environment) re-executes under torch.distributed.run with N ranks on 127.0.0.1
    -- the command the driver uses for N > 1 -- and the ranks are told not to launch again."""
    sys.path.insert(0, REPO)
    bench = importlib.import_module("bench")
    cmd = bench.launch_command(["--gpus", "8", "--steps", "3", "--launcher", "torchrun", "--warmup", "1"], 8, 29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=8" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    tail = cmd[cmd.index(os.path.join(REPO, "bench.py")) + 1:]
    assert tail == ["--gpus", "8", "--steps", "3", "--warmup", "1", "--launcher", "none"]
    # on this GPU-less container the launch refuses with a message instead of spawning ranks that cannot run
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "GPU" in r.stderr


def test_bench_plumbing_of_the_eight_gpu_runs():
    """The first real 8-GPU run must not fail on plumbing: argument parsing + workload resolution of every rank
    for the driver's weak-scaling command and for BASELINE config 4 / 5 as strong scaling, and the re-launch command a plain
    `python bench.py --gpus 8 --config 4 --scaling strong` turns itself into."""
    sys.path.insert(0, REPO)
    bench = importlib.import_module("bench")
    d = importlib.import_module("pytorch-deepfepe_amd")
    # the driver's command: weak scaling, 4096 pairs per GPU, 8 x 4096 = BASELINE config 4's 32768 in total
    args = bench.parse(["--gpus", "8", "--steps", "20", "--warmup", "5"])
    for r in range(8):
        wl = bench.resolve_workload(args, r, 8, d.dist.shard_range)
        assert (wl["B"], wl["B_total"], wl["N"], wl["L"], wl["scaling"], wl["kind"], wl["grad_pairs"]) == (4096, 32768, 100, 5, "weak", "train", 32768)
    # config 4 as north_star words it: 32768 pairs batch-sharded over the ranks, means over the GLOBAL batch
    args = bench.parse(["--gpus", "8", "--config", "4", "--scaling", "strong"])
    shards = [bench.resolve_workload(args, r, 8, d.dist.shard_range) for r in range(8)]
    assert all(w["B"] == 4096 and w["B_total"] == 32768 and w["grad_pairs"] == 32768 and w["cfg"]["balance_F"] == 0.0 and w["outliers"] == 0.4 for w in shards)
    # config 5: 4096 x 1000 in total, strong by default; an uneven world size still partitions the batch
    args = bench.parse(["--gpus", "8", "--config", "5"])
    assert [bench.resolve_workload(args, r, 8, d.dist.shard_range)["B"] for r in range(8)] == [512] * 8
    args = bench.parse(["--gpus", "3", "--config", "5"])
    assert sum(bench.resolve_workload(args, r, 3, d.dist.shard_range)["B"] for r in range(3)) == 4096
    # an explicit --batch: per GPU when weak, in total when strong
    args = bench.parse(["--gpus", "2", "--batch", "1000"])
    assert bench.resolve_workload(args, 1, 2, d.dist.shard_range)["B_total"] == 2000
    args = bench.parse(["--gpus", "2", "--batch", "1001", "--scaling", "strong"])
    assert [bench.resolve_workload(args, r, 2, d.dist.shard_range)["B"] for r in range(2)] == [501, 500]
    cmd = bench.launch_command(["--gpus", "8", "--config", "4", "--scaling", "strong", "--launcher=auto"], 8, 12345)
    assert cmd[cmd.index(os.path.join(REPO, "bench.py")) + 1:] == ["--gpus", "8", "--config", "4", "--scaling", "strong", "--launcher", "none"]
    assert "--nproc-per-node=8" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
