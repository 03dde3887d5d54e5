"""Mirror of the two loss functions of deepFEPE/train_good_utils.py on the hot path:
get_all_loss_DeepF (:298-520) and get_Rt_loss (:64-295), same arguments and return structures."""
import numpy as np
import torch
import torch.nn as nn

from .. import _lib, ops


def mean_list(lst):
    return sum(lst) / len(lst)


def get_unique(xs, topk, matches_good_unique_nums):
    """top-k of the first `unique_num` entries per sample (train_good_utils.py get_unique)."""
    out = []
    for x, n in zip(xs, matches_good_unique_nums):
        out.append(torch.topk(x[: int(n)], topk, dim=0)[0])
    return torch.stack(out)


def get_all_loss_DeepF(outs, pts1_virt_ori, pts2_virt_ori, Ks, loss_params, get_residual_summaries=True):
    """F-loss on the virtual correspondences for every layer + E-from-F.  Returns, like the reference (:511-519),
    (losses_dict, E_ests, F_ests, logits_softmax, residual_norm_layers, residual_norm_max_layers, E_ests_layers)."""
    if loss_params.get("if_tri_depth", False) or loss_params.get("if_sample_loss", False):
        raise NotImplementedError("if_tri_depth / if_sample_loss are outside the built hot path (all shipped configs disable them)")
    logits_softmax = outs["weights"]
    F_est_normalized, T1, T2 = outs["F_est"], outs["T1"], outs["T2"]
    out_layers, residual_layers, weights_layers = outs["out_layers"], outs["residual_layers"], outs["weights_layers"]
    depth = loss_params["depth"]
    F_layers = torch.stack(list(out_layers[:depth]))  # [L,B,3,3]
    M = pts1_virt_ori.shape[1]
    B = F_layers.shape[1]
    loss_sum, E_layers = ops.floss(F_layers, T1, T2, Ks, pts1_virt_ori, pts2_virt_ori, loss_params["clamp_at"])
    per_pair = loss_sum / float(M)  # losses.mean(dim=1) per layer  [L,B]
    loss_layers = [per_pair[i].mean() for i in range(depth)]
    loss_F_all = sum(loss_layers) / len(loss_layers)
    E_ests_layers = [E_layers[i] for i in range(depth)]
    F_ests = T2.permute(0, 2, 1) @ F_est_normalized @ T1
    E_ests = Ks.transpose(1, 2) @ F_ests @ Ks
    losses_dict = {
        "loss_layers": loss_layers,
        "loss_F": loss_F_all,
        "loss_min_layers": per_pair.min(dim=1)[0],
        "loss_min_batch": per_pair.min(dim=0)[0],
    }
    loss_epi_res_all = 0.0
    loss_epi_res_layers = []
    if depth > 1:
        for epi_res, weights in zip(outs["epi_res_layers"], outs["weights_layers"]):
            loss_epi_res_layers.append((epi_res * weights).mean())
        loss_epi_res_all = sum(loss_epi_res_layers) / len(loss_epi_res_layers)
    losses_dict.update({"loss_epi_res_layers": loss_epi_res_layers, "loss_epi_res": loss_epi_res_all})

    residual_norm_layers, residual_norm_max_layers = None, None
    if get_residual_summaries:  # logging-only summaries (:441-509), plain tensor reductions
        topK = loss_params["topK"]
        nums = loss_params["matches_good_unique_nums"]
        residual_norm_layers, residual_norm_topK_layers, residual_norm_max_layers = [], [], []
        for residual in residual_layers:
            norms = residual.norm(p=2, dim=1)
            residual_norm_layers.append(norms.mean())
            residual_norm_max_layers.append(norms.max())
            residual_norm_topK_layers.append(get_unique(residual, topK, nums).mean())
        regW_clip, entro, entro_topK = [], [], []
        for w in weights_layers:
            regW_clip.append(nn.ReLU()(w - 0.01).mean())
            entro.append(torch.distributions.Categorical(probs=w.squeeze(1)).entropy().mean())
            entro_topK.append(torch.distributions.Categorical(probs=get_unique(w.squeeze(1), topK, nums)).entropy().mean())
        losses_dict.update({
            "loss_residual": sum(residual_norm_layers) / len(residual_norm_layers),
            "loss_residual_topK": sum(residual_norm_topK_layers) / len(residual_norm_topK_layers),
            "loss_regW_clip": sum(regW_clip) / len(weights_layers) * 100.0,
            "loss_regW_entro": sum(entro) / len(weights_layers),
            "loss_regW_entro_topK": sum(entro_topK) / len(weights_layers),
        })
    return losses_dict, E_ests, F_ests, logits_softmax, residual_norm_layers, residual_norm_max_layers, E_ests_layers


def get_Rt_loss(E_ests_layers, Ks_cpu, x1_cpu, x2_cpu, delta_Rtijs_4_4_cpu, qs_cam, ts_cam, device="cpu"):
    """Pose loss from the per-layer essential matrices and the ground-truth camera motion.  Same 12-key dict as the
    reference (:272-293).  NB the reference stacks the *translation* list under "q_l2_error_list" (:276); that slip
    is reproduced so downstream logging sees identical values.  Ks/x1/x2 are unused, as in the reference."""
    E_layers = torch.stack(list(E_ests_layers))  # [L,B,3,3]
    if not E_layers.is_cuda:
        raise _lib.DfepeError("get_Rt_loss: E_ests_layers must live on the GPU")
    dev = E_layers.device
    delta = torch.as_tensor(delta_Rtijs_4_4_cpu).to(dev).float()
    R_gt = torch.linalg.inv(delta)[:, :3, :3].contiguous()
    q_l2, t_l2, R_deg, t_deg, _ = ops.pose_errors(E_layers, torch.as_tensor(qs_cam).to(dev), torch.as_tensor(ts_cam).to(dev), R_gt)
    L = E_layers.shape[0]
    R_np, t_np = R_deg.cpu().numpy().astype(np.float64), t_deg.cpu().numpy().astype(np.float64)
    t_l2_layers = [t_l2[i] for i in range(L)]
    q_l2_layers = [q_l2[i] for i in range(L)]
    t_means = [x.mean() for x in t_l2_layers]
    q_means = [x.mean() for x in q_l2_layers]
    R_means = [float(R_np[i].mean()) for i in range(L)]
    tA_means = [float(t_np[i].mean()) for i in range(L)]
    return {
        "t_l2_error_mean": mean_list(t_means),
        "q_l2_error_mean": mean_list(q_means),
        "t_l2_error_list": torch.stack(t_means),
        "q_l2_error_list": torch.stack(t_means),  # sic: reference train_good_utils.py:276
        "R_angle_error_mean": mean_list(R_means),
        "R_angle_error_list": np.array(R_means),
        "t_angle_error_mean": mean_list(tA_means),
        "t_angle_error_list": np.array(tA_means),
        "R_angle_error_layers_list": [R_np[i] for i in range(L)],
        "t_angle_error_layers_list": [t_np[i] for i in range(L)],
        "t_l2_error_layers_list": t_l2_layers,
        "q_l2_error_layers_list": q_l2_layers,
    }


_warned_opencv = False


def val_rt(idx, K_np, x1_single_np, x2_single_np, E_est_np, E_gt_np, F_est_np, F_gt_np, delta_Rtijs_4_4_cpu_np, five_point,
           if_opencv=False):
    """Same call and same 9-tuple as the reference's per-pair validation worker (train_good_utils.py:553-646):
    (error_Rt_estW, epi_dist_mean_estW, error_Rt_opencv, epi_dist_mean_opencv, error_Rt_gt, epi_dist_mean_gt, idx, M_estW, M_opencv).
    The estimated-E and ground-truth-E legs go through utils_F.goodCorr_eval_nondecompose (the cheirality kernel in place of
    cv2.recoverPose) and utils_F.epi_distance_np.  The OpenCV five-point / eight-point RANSAC baseline (``if_opencv``,
    utils_opencv.recover_camera_opencv) is outside this build (OpenCV baselines): its three slots are None.
    ``if_opencv`` defaults to False here (the reference's default is True): a caller that asks for the OpenCV leg gets one
    warning saying that its slots stay None, instead of an unrelated TypeError when it indexes them later.
    One pair per call like the reference; val_rt_batch / validation_summary below are the batched forms."""
    from . import utils_F

    global _warned_opencv
    if if_opencv and not _warned_opencv:
        import warnings

        warnings.warn("val_rt(if_opencv=True): the OpenCV five-point / RANSAC baseline is not part of this build (SURVEY.md §2); "
                      "error_Rt_opencv, epi_dist_mean_opencv and M_opencv are returned as None", RuntimeWarning, stacklevel=2)
        _warned_opencv = True

    delta_Rtij_inv = np.linalg.inv(np.asarray(delta_Rtijs_4_4_cpu_np))[:3]
    M_estW, error_Rt_estW = utils_F.goodCorr_eval_nondecompose(x1_single_np, x2_single_np, np.asarray(E_est_np).astype(np.float64),
                                                                delta_Rtij_inv, K_np, None)
    M_gt, error_Rt_gt = utils_F.goodCorr_eval_nondecompose(x1_single_np, x2_single_np, np.asarray(E_gt_np).astype(np.float64),
                                                            delta_Rtij_inv, K_np, None)
    epi_dist_mean_estW, _, _ = utils_F.epi_distance_np(F_est_np, x1_single_np, x2_single_np, if_homo=False)
    epi_dist_mean_gt, _, _ = utils_F.epi_distance_np(F_gt_np, x1_single_np, x2_single_np, if_homo=False)
    return (error_Rt_estW, epi_dist_mean_estW, None, None, error_Rt_gt, epi_dist_mean_gt, idx, M_estW, None)


def val_rt_batch(Ks, matches_xy, E_ests, delta_Rtijs_4_4, project_E=True, depth_thres=50.0):
    """Batched GPU counterpart of the validation fan-out (Train_model_pipeline.py:954-964,1048-1061 -> val_rt :553-646 ->
    utils_F.goodCorr_eval_nondecompose -> cv2.recoverPose): optional projection of E onto singular values (1,1,0),
    cheirality-checked pose for every pair, rotation / translation angular errors against the ground-truth camera motion.
    Ks, E_ests [B,3,3]; matches_xy [B,N,4] pixels; delta_Rtijs_4_4 [B,4,4] (scene motion, like the dataset).
    Returns dict(err_R_deg [B], err_t_deg [B], Rt_cam [B,3,4], winner [B], counts [B,4]); pairs without a valid
    candidate get the reference's failure values 180 / 90 degrees."""
    E = E_ests.float()
    if not E.is_cuda:
        raise _lib.DfepeError("val_rt_batch: tensors must live on the GPU")
    dev = E.device
    if project_E:
        E = ops.project_essential(E)
    Rt, win, cnt = ops.cheirality(E, Ks.to(dev), matches_xy.to(dev), depth_thres)
    gt = torch.linalg.inv(delta_Rtijs_4_4.to(dev).float())
    err_R = ops.rot_angle_deg(Rt[:, :, :3].contiguous(), gt[:, :3, :3].contiguous())
    err_t = ops.vector_angle_deg(Rt[:, :, 3].contiguous(), gt[:, :3, 3].contiguous())
    bad = win < 0
    err_R = torch.where(bad, torch.full_like(err_R, 180.0), err_R)
    err_t = torch.where(bad, torch.full_like(err_t, 90.0), err_t)
    return {"err_R_deg": err_R, "err_t_deg": err_t, "Rt_cam": Rt, "winner": win, "counts": cnt}


def validation_summary(Ks, matches_xy, E_ests, F_ests, F_gts, delta_Rtijs_4_4, project_E=True, depth_thres=50.0):
    """One validation batch end to end on the device: val_rt_batch (pose errors of every pair), the epipolar distances of
    every correspondence under the estimated and the ground-truth F (epi_distance_np: d1 + d2, utils_F.py:363-385, as val_rt
    calls it, train_good_utils.py:609-614), and the reductions write_metrics_summary applies to them (:758-856).
    Returns (summary dict of python floats for the 'ours' tag, per-pair dict of device tensors)."""
    from . import utils_F

    pairs = val_rt_batch(Ks, matches_xy, E_ests, delta_Rtijs_4_4, project_E=project_E, depth_thres=depth_thres)
    X, Y = matches_xy[:, :, :2].contiguous(), matches_xy[:, :, 2:].contiguous()
    d_est = 2.0 * utils_F._epi_distance(F_ests, X, Y)[0]  # (d1 + d2), the first return value of epi_distance_np
    d_gt = 2.0 * utils_F._epi_distance(F_gts, X, Y)[0]
    pairs.update({"epi_dists": d_est, "epi_dists_gt": d_gt})
    return ops.metrics_summary(d_est, d_gt, pairs["err_R_deg"], pairs["err_t_deg"]), pairs


def write_metrics_summary(writer, dict_of_lists, task, n_iter):
    """Same call and same scalar tags as the reference's write_metrics_summary (train_good_utils.py:758-856).  The values of
    dict_of_lists[metric][exp] may be lists of numpy arrays (the reference's layout) or of device tensors; the counting,
    F1, median, maximum and histogram reductions run on the device (ops.metrics_summary), one small copy per experiment tag.
    Histograms (writer.add_histogram) receive the host copy of the error vectors like the reference's."""
    metric_list = list(dict_of_lists.keys())
    exp_list = list(dict_of_lists[metric_list[0]].keys())
    assert "epi_dists" in metric_list
    dev = torch.device("cuda")

    def flat(v):
        parts = [torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x, dtype=torch.float32).reshape(-1) for x in (v if isinstance(v, (list, tuple)) else [v])]
        return torch.cat([p.to(dev) for p in parts])

    gt_epi = flat(dict_of_lists["epi_dists"]["gt"])
    for tag in exp_list:
        epi = flat(dict_of_lists["epi_dists"][tag])
        err_q, err_t = flat(dict_of_lists["err_q"][tag]), flat(dict_of_lists["err_t"][tag])
        sm = ops.metrics_summary(epi, gt_epi, err_q, err_t)
        writer.add_scalar(task + "-Error-epi_dists/%s-0.1" % tag, sm["ratio_0.1"], n_iter)
        writer.add_scalar(task + "-Error-epi_dists/%s-1" % tag, sm["ratio_1"], n_iter)
        writer.add_scalar(task + "-Error-F1/%s-0.1" % tag, sm["F1_0.1"], n_iter)
        writer.add_scalar(task + "-Error-F1/%s-1" % tag, sm["F1_1"], n_iter)
        for metric in metric_list:
            if metric == "epi_dists":
                continue
            if metric == "err_q":
                med = sm["median_err_q"]
            elif metric == "err_t":
                med = sm["median_err_t"]
            else:  # any further per-pair metric: the same device median
                other = flat(dict_of_lists[metric][tag])
                med = ops.metrics_summary(epi[:0], None, other, other)["median_err_q"]
            writer.add_scalar(task + "-Error-Median/%s-%s" % (metric, tag), med, n_iter)
        writer.add_scalar(task + "-Error-MAX/err_q_MAX_%s" % tag, sm["max_err_q"], n_iter)
        writer.add_scalar(task + "-Error-MAX/err_t_MAX_%s" % tag, sm["max_err_t"], n_iter)
        if hasattr(writer, "add_histogram"):
            q_np, t_np = err_q.cpu().numpy(), err_t.cpu().numpy()
            writer.add_histogram(task + "-Error-hist/err_q_%s" % tag, q_np, n_iter)
            writer.add_histogram(task + "-Error-hist/err_t_%s" % tag, t_np, n_iter)
            writer.add_histogram(task + "-Error-hist/err_t-Clip10.degree_%s" % tag, np.clip(t_np, 0.0, 10.0), n_iter)
        for k, th in enumerate(ops.METRIC_THS[1:]):
            writer.add_scalar(task + "-Error-ratio/ratio_q{}_{}".format(th, tag), sm["ratio_q"][k], n_iter)
            writer.add_scalar(task + "-Error-ratio/ratio_t{}_{}".format(th, tag), sm["ratio_t"][k], n_iter)


def matches_from_SP_outputs(xs_SP, deses_SP, reses_SP, nn_thresh, out_num_points=1000):
    """The per-pair loop of get_matches_from_SP (train_good_utils.py:679-724) for the whole batch on the GPU.
    xs_SP, reses_SP: two tensors [B,N,2] (keypoints, sub-pixel offsets); deses_SP: two tensors [B,N,D] (unit norm).
    One matching launch for all pairs, one D2H copy of the B match counts (the crop/pad permutation is drawn on the host
    from numpy's RNG exactly like utils_misc.crop_or_pad_choice, so seeded runs agree), one gather launch."""
    from .utils_misc import crop_or_pad_choice

    if not deses_SP[0].is_cuda:
        raise _lib.DfepeError("matches_from_SP_outputs: tensors must live on the GPU")
    m1, m2, sc, cnt = ops.nn_match_two_way(deses_SP[0], deses_SP[1], float(nn_thresh))
    counts = cnt.cpu().numpy()
    choice = np.stack([crop_or_pad_choice(int(n), out_num_points, shuffle=True) for n in counts]).astype(np.int32)
    choice_dev = torch.from_numpy(choice).to(m1.device)
    xs, offsets, quality = ops.gather_matches(xs_SP[0], xs_SP[1], reses_SP[0], reses_SP[1], m1, m2, sc, choice_dev)
    xs_all = [x + r for (x, r) in zip(xs_SP, reses_SP)]
    return {"xs": xs, "offsets": offsets, "quality": quality, "num_matches": torch.from_numpy(counts.astype(np.int64)),
            "xs_SP": xs_all}


def get_matches_from_SP(imgs_grey, net_SP, SP_processer, SP_tracker, out_num_points=1000, process_SP_output=None):
    """Same call as the reference's get_matches_from_SP (train_good_utils.py:649-724).  The SuperPoint front-end stays
    the caller's: ``net_SP`` and ``process_SP_output`` (by default the reference's own, importable when this runs inside
    the reference tree, :665) produce keypoints / descriptors / offsets; the matching, crop/pad and gather of all pairs
    then run as two launches instead of a per-pair numpy loop.  ``SP_tracker`` only supplies ``nn_thresh``."""
    if process_SP_output is None:
        from train_good_utils import process_SP_output  # the reference's module (superpoint front-end, out of scope here)
    imgs_grey_float = [img_grey.float().cuda() / 255.0 for img_grey in imgs_grey]
    xs_SP, deses_SP, reses_SP = [], [], []
    for img12_grey_float in imgs_grey_float:
        outs = net_SP(img12_grey_float.unsqueeze(-1).permute(0, 3, 1, 2))  # [batch_size, 1, H, W]
        outs = process_SP_output(outs, SP_processer)
        xs_SP.append(outs["pts_int"])
        deses_SP.append(outs["pts_desc"])
        reses_SP.append(outs["pts_offset"])
    return matches_from_SP_outputs(xs_SP, deses_SP, reses_SP, SP_tracker.nn_thresh, out_num_points)
