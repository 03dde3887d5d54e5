"""compat.CapturedStep: the reference's eager training step -- net(data_batch) -> get_all_loss_DeepF ->
get_Rt_loss -> the caller's clamp / balance lines -> backward (Train_model_pipeline.py:495-595) -- captured once per batch
signature and replayed.  Its outputs and parameter gradients must equal the eager sequence on every batch it is given, follow the
optimizer's in-place parameter updates, survive zero_grad(set_to_none=True), re-capture on a shape change and accept host-side
(numpy / CPU tensor) ground truth like the reference's loader delivers it.  GPU box only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
IMAGE_SIZE = [376, 1241, 3]
DEV = "cuda:0"


def _batch(dfepe, B, N, seed, host_gt=False):
    sc = dfepe.synth.make_scene(B, N, seed=seed, outlier_ratio=0.2, noise_px=0.5)
    keys = ("matches_xy_ori", "pts1_virt_ori", "pts2_virt_ori", "Ks", "delta_Rtijs_4_4", "qs_cam", "ts_cam")
    b = {k: sc[k].to(DEV) for k in keys}
    if host_gt:  # the reference's loader hands these over on the host (Train_model_pipeline.py:434-447)
        b["qs_cam"], b["ts_cam"] = sc["qs_cam"].numpy(), sc["ts_cam"]
        b["delta_Rtijs_4_4"] = sc["delta_Rtijs_4_4"]
    return b


def _make_step(dfepe, net, depth, pose_gt):
    tgu = dfepe.compat.train_good_utils

    def forward_and_loss(b):
        lp = {"depth": depth, "clamp_at": 0.02, "if_tri_depth": False, "if_sample_loss": False, "topK": 8, "matches_good_unique_nums": None}
        if pose_gt:
            lp["pose_gt"] = (b["qs_cam"], b["ts_cam"], b["delta_Rtijs_4_4"])
        outs = net({"matches_xy_ori": b["matches_xy_ori"], "matches_good_unique_nums": None, "t_scene_scale": None})
        losses, _, _, _, _, _, E_layers = tgu.get_all_loss_DeepF(outs, b["pts1_virt_ori"], b["pts2_virt_ori"], b["Ks"], lp, get_residual_summaries=False)
        geo = tgu.get_Rt_loss(E_layers, b["Ks"], None, None, b["delta_Rtijs_4_4"], b["qs_cam"], b["ts_cam"], device=DEV)
        lq = torch.clamp(torch.stack(geo["q_l2_error_layers_list"]), 0.0, 0.1).mean()
        lt = torch.clamp(torch.stack(geo["t_l2_error_layers_list"]), 0.0, 0.5).mean()
        return losses["loss_F"] + lq + 0.1 * lt, {"losses": losses, "geo": geo}

    return forward_and_loss


def _eager(net, fn, b):
    net.zero_grad(set_to_none=True)
    loss, aux = fn(b)
    loss.backward()
    torch.cuda.synchronize()
    return loss.detach().clone(), [p.grad.detach().clone() for p in net.parameters()], aux


@pytest.mark.parametrize("pose_gt", [False, True])
def test_captured_step_equals_the_eager_sequence_over_batches(dfepe, pose_gt):
    depth, N = 3, 100
    net = dfepe.compat.DeepFNet.DeepFNet(depth=depth, image_size=IMAGE_SIZE, if_quality=False).to(DEV)
    dfepe.synth.fill_params_deterministic(net, 3)
    fn = _make_step(dfepe, net, depth, pose_gt)
    step = dfepe.compat.CapturedStep(fn, net, warmup=2)
    batches = [_batch(dfepe, 48, N, 100 + k, host_gt=(k == 1)) for k in range(3)]
    dev_batches = [{k: (torch.as_tensor(v).to(DEV)) for k, v in b.items()} for b in batches]
    worst = 0.0
    for rnd in range(3):
        for b, bd in zip(batches, dev_batches):
            ref_loss, ref_g, ref_aux = _eager(net, fn, bd)
            ref_R = np.asarray(ref_aux["geo"]["R_angle_error_layers_list"][-1]).copy()
            net.zero_grad(set_to_none=True)  # what optimizer.zero_grad() does: the step must re-attach its static gradients
            loss, aux = step(b)
            torch.cuda.synchronize()
            torch.testing.assert_close(loss.detach(), ref_loss, rtol=1e-6, atol=1e-8)
            for p, g in zip(net.parameters(), ref_g):
                assert p.grad is not None
                torch.testing.assert_close(p.grad, g, rtol=1e-5, atol=1e-7 * float(g.abs().max()) + 1e-12)
                worst = max(worst, float((p.grad - g).abs().max() / g.abs().max().clamp_min(1e-30)))
            host = dfepe.compat.CapturedStep.realise(aux)
            got_R = host["geo"]["R_angle_error_layers_list"][-1]
            assert isinstance(got_R, np.ndarray) and isinstance(host["geo"]["R_angle_error_mean"], float)
            np.testing.assert_allclose(got_R, ref_R, atol=1e-5)
    # two signatures (device ground truth: batches 0 and 2; host ground truth: batch 1), each: two eager steps, then its graph
    assert (step.n_eager, step.n_captures, step.n_replays) == (4, 2, 5), (step.n_eager, step.n_captures, step.n_replays)
    print(f"captured vs eager: worst relative gradient difference {worst:.2e}")
    # another batch size: a second graph, same answers
    b2 = _batch(dfepe, 20, N, 7)
    ref_loss, ref_g, _ = _eager(net, fn, b2)
    for _ in range(3):
        net.zero_grad(set_to_none=True)
        loss, _ = step(b2)
    torch.cuda.synchronize()
    assert step.n_captures == 3
    torch.testing.assert_close(loss.detach(), ref_loss, rtol=1e-6, atol=1e-8)
    for p, g in zip(net.parameters(), ref_g):
        torch.testing.assert_close(p.grad, g, rtol=1e-5, atol=1e-7 * float(g.abs().max()) + 1e-12)
    # and back to the first signature: its graph is still there
    loss, _ = step(batches[0])
    assert step.n_captures == 3


def test_captured_step_follows_the_optimizer(dfepe):
    """Replays read the parameters' current values: three SGD steps through the captured step land on the same parameters as three
    eager ones."""
    depth, N = 2, 100
    nets = []
    for _ in range(2):
        net = dfepe.compat.DeepFNet.DeepFNet(depth=depth, image_size=IMAGE_SIZE, if_quality=False).to(DEV)
        dfepe.synth.fill_params_deterministic(net, 5)
        nets.append(net)
    batches = [_batch(dfepe, 32, N, 40 + k) for k in range(5)]
    fn_e, fn_c = _make_step(dfepe, nets[0], depth, True), _make_step(dfepe, nets[1], depth, True)
    opt_e = torch.optim.SGD(nets[0].parameters(), lr=1e-3)
    opt_c = torch.optim.SGD(nets[1].parameters(), lr=1e-3)
    step = dfepe.compat.CapturedStep(fn_c, nets[1], warmup=1)
    losses = []
    for b in batches:
        opt_e.zero_grad()
        le, _ = fn_e(b)
        le.backward()
        opt_e.step()
        opt_c.zero_grad()
        lc, _ = step(b)
        opt_c.step()
        losses.append((float(le), float(lc)))
    assert step.n_captures == 1 and step.n_replays == 4
    for le, lc in losses:
        assert abs(le - lc) <= 1e-5 * max(1.0, abs(le)), losses
    for pe, pc in zip(nets[0].parameters(), nets[1].parameters()):
        torch.testing.assert_close(pc, pe, rtol=1e-5, atol=1e-7)
