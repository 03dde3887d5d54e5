"""The reference's only multi-GPU mode is nn.DataParallel (deepFEPE/train_good.py:311-312: one process, one python THREAD per replica
through DeepFNet.forward, parameters re-broadcast as non-leaf copies every step).  On a one-GPU box the same machinery runs with
device_ids=[0, 0]: two replicas, two threads, one device -- everything the drop-in keeps per module, per thread or per process is
exercised (the estimators' shared parameter preparation, the cached image-size transform, the per-thread fused loss tail and pinned
buffers, the module-level caches).  VERDICT r5 item 6.  GPU box only."""
import copy
import threading

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
IMAGE_SIZE = [376, 1241, 3]
DEV = "cuda:0"
KEYS = ("matches_xy_ori", "pts1_virt_ori", "pts2_virt_ori", "Ks", "delta_Rtijs_4_4", "qs_cam", "ts_cam")


def _batch(dfepe, B, N, seed, outliers=0.0):
    sc = dfepe.synth.make_scene(B, N, seed=seed, outlier_ratio=outliers, noise_px=0.5)
    return {k: sc[k].to(DEV) for k in KEYS}


def _loss(dfepe, model, b, depth, pose_gt=False):
    tgu = dfepe.compat.train_good_utils
    lp = {"depth": depth, "clamp_at": 0.02, "if_tri_depth": False, "if_sample_loss": False, "topK": 8, "matches_good_unique_nums": None}
    if pose_gt:
        lp["pose_gt"] = (b["qs_cam"], b["ts_cam"], b["delta_Rtijs_4_4"])
    outs = model({"matches_xy_ori": b["matches_xy_ori"], "matches_good_unique_nums": None, "t_scene_scale": None})
    losses, _, _, _, _, _, E_layers = tgu.get_all_loss_DeepF(outs, b["pts1_virt_ori"], b["pts2_virt_ori"], b["Ks"], lp, get_residual_summaries=False)
    geo = tgu.get_Rt_loss(E_layers, b["Ks"], None, None, b["delta_Rtijs_4_4"], b["qs_cam"], b["ts_cam"], device=DEV)
    lq = torch.clamp(torch.stack(geo["q_l2_error_layers_list"]), 0.0, 0.1).mean()
    lt = torch.clamp(torch.stack(geo["t_l2_error_layers_list"]), 0.0, 0.5).mean()
    return losses["loss_F"] + lq + 0.1 * lt, outs, geo


def test_two_replicas_under_dataparallel_equal_the_unwrapped_model(dfepe):
    """nn.DataParallel(net, device_ids=[0, 0]) through compat.DeepFNet.forward -> both loss functions -> backward, three Adam steps:
    outputs, losses and parameters track the unwrapped model on the same batches (each replica sees half the pairs; every reduction
    of the path is inside a pair, so only the parameter gradients' summation order differs).  The estimators' heads are scaled down
    and the scenes outlier-free (well-conditioned fits: see tests/test_estimator_r6_gpu.py on what random-weight fits do to
    comparisons between two fp32 evaluations)."""
    depth, B, N = 3, 12, 100
    net = dfepe.compat.DeepFNet.DeepFNet(depth=depth, image_size=IMAGE_SIZE, if_quality=False).to(DEV)
    dfepe.synth.fill_params_deterministic(net, 3)
    with torch.no_grad():
        for est in (net.input_weights, net.update_weights):
            est.fw[-1].weight.mul_(0.05)
    solo = copy.deepcopy(net)
    dp = nn.DataParallel(net, device_ids=[0, 0])
    opt_dp = torch.optim.Adam(net.parameters(), lr=1e-4)
    opt_solo = torch.optim.Adam(solo.parameters(), lr=1e-4)
    for stp in range(3):
        b = _batch(dfepe, B, N, 20 + stp)
        opt_dp.zero_grad(set_to_none=True)
        opt_solo.zero_grad(set_to_none=True)
        la, oa, ga = _loss(dfepe, dp, b, depth)
        lb, ob, gb = _loss(dfepe, solo, b, depth)
        assert set(oa.keys()) == set(ob.keys())
        for l in range(depth):
            assert oa["logits_layers"][l].shape == ob["logits_layers"][l].shape == (B, 1, N)
            torch.testing.assert_close(oa["logits_layers"][l], ob["logits_layers"][l], rtol=1e-4, atol=2e-5)
            torch.testing.assert_close(oa["out_layers"][l], ob["out_layers"][l], rtol=1e-3, atol=1e-5 * float(ob["out_layers"][l].abs().max()))
        torch.testing.assert_close(la, lb, rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(np.asarray(ga["R_angle_error_layers_list"][-1]), np.asarray(gb["R_angle_error_layers_list"][-1]), rtol=1e-3, atol=1e-4)
        la.backward()
        lb.backward()
        top = max(float(p.grad.norm()) for p in solo.parameters())
        for (name, pa), (_, pb) in zip(net.named_parameters(), solo.named_parameters()):
            assert pa.grad is not None, name
            dist = float((pa.grad - pb.grad).norm())
            assert dist < 5e-3 * float(pb.grad.norm()) + 1e-4 * top, (stp, name, dist, float(pb.grad.norm()), top)
        opt_dp.step()
        opt_solo.step()
    for (name, pa), (_, pb) in zip(net.named_parameters(), solo.named_parameters()):
        torch.testing.assert_close(pa, pb, rtol=1e-3, atol=2e-5, msg=name)
    # nothing of a replica's scope leaked into the wrapped module
    assert net.update_weights._prepared is None and net.input_weights._prepared is None


def test_loss_functions_hammered_from_two_threads(dfepe):
    """get_all_loss_DeepF (with and without the fused pose tail) + get_Rt_loss from two threads at once on DIFFERENT batches, eight
    rounds: every thread gets, bit for bit, what the same calls return single-threaded (the per-thread fused tail, the pinned
    host-metric buffers, the module-level transform cache)."""
    depth, N = 3, 100
    nets, batches, want = [], [], []
    for t in range(2):
        net = dfepe.compat.DeepFNet.DeepFNet(depth=depth, image_size=IMAGE_SIZE, if_quality=False).to(DEV)
        dfepe.synth.fill_params_deterministic(net, 3 + t)
        nets.append(net)
        batches.append(_batch(dfepe, 8 + 8 * t, N, 50 + t, outliers=0.2))

    def run(t, pose_gt):
        with torch.no_grad():
            loss, outs, geo = _loss(dfepe, nets[t], batches[t], depth, pose_gt=pose_gt)
        return (loss.clone(), torch.stack(geo["q_l2_error_layers_list"]).clone(), np.asarray(geo["R_angle_error_layers_list"][-1]).copy(),
                float(geo["t_angle_error_mean"]))

    for t in range(2):
        want.append({pg: run(t, pg) for pg in (False, True)})
    torch.cuda.synchronize()
    errors = []
    barrier = threading.Barrier(2)

    def worker(t):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for rnd in range(8):
                    barrier.wait(timeout=60)
                    pg = bool((rnd + t) & 1)
                    got = run(t, pg)
                    stream.synchronize()
                    ref = want[t][pg]
                    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), (t, rnd)
                    assert np.array_equal(got[2], ref[2]) and got[3] == ref[3], (t, rnd)
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))
            try:
                barrier.abort()
            except Exception:  # noqa: BLE001
                pass

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errors, errors
