"""Mirror of the two loss functions of deepFEPE/train_good_utils.py on the hot path:
get_all_loss_DeepF (:298-520) and get_Rt_loss (:64-295), same arguments and return structures."""
import threading
import weakref

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


# ---------------------------------------------------------------------------------------------------------------------------
# host-side metrics that do not force a device synchronisation
# ---------------------------------------------------------------------------------------------------------------------------
LAZY_HOST_METRICS = False  # False (default): get_Rt_loss copies the angular errors to the host at once and returns exactly the
# reference's types (numpy arrays / python floats; one device synchronisation per call, like the reference's .cpu().numpy()).
# True -- an opt-in of graph-capturing callers (pipeline.CapturedStep, bench.py) -- defers that copy to the first read (_Lazy).
# While the current stream is being captured the values are always lazy: a copy to the host cannot be captured.


class _Lazy:
    """A value of get_Rt_loss's dict that lives on the host in the reference (numpy arrays / python floats of the angular
    errors, train_good_utils.py:173-178,272-293) and is only *logged* there.  Here the device-to-host copy happens on first use:
    the training step has no `.cpu()` synchronisation (and can be captured in a hipGraph); whoever reads the value -- np.asarray,
    float(), indexing, arithmetic, attribute access -- gets the reference's numpy array / float, computed from the device
    buffer as it is at that moment."""

    __array_priority__ = 100.0

    def __init__(self, compute):
        self._compute = compute

    def _v(self):
        return self._compute()

    def __array__(self, dtype=None, copy=None):
        a = np.asarray(self._v())
        return a.astype(dtype) if dtype is not None else a

    def __getattr__(self, name):  # shape, mean(), flatten(), ... of the realised value
        if name.startswith("__"):
            raise AttributeError(name)
        return getattr(self._v(), name)

    def __float__(self):
        return float(self._v())

    def __len__(self):
        return len(self._v())

    def __iter__(self):
        return iter(self._v())

    def __getitem__(self, k):
        return self._v()[k]

    def __repr__(self):
        return repr(self._v())

    def __format__(self, spec):
        return format(self._v(), spec)

    def __bool__(self):
        return bool(self._v())

    def __reduce__(self):  # pickle / torch.save / copy.deepcopy store the realised numpy array or float, not the closure
        return (_realised, (self._v(),))


def _realised(v):
    return v


class _LazyScalar(_Lazy):
    """A lazy python float (the two *_angle_error_mean values): np.isscalar() and tensorboard's make_np accept it.  Only this
    subclass is registered as numbers.Real; the array-valued lazies are not."""


for _name in ("add", "sub", "mul", "truediv", "floordiv", "pow", "mod", "lt", "le", "gt", "ge", "eq", "ne"):
    def _make(n):
        return lambda self, other: getattr(np.asarray(self._v()), f"__{n}__")(other._v() if isinstance(other, _Lazy) else other)
    setattr(_Lazy, f"__{_name}__", _make(_name))
for _name in ("add", "sub", "mul", "truediv", "pow"):
    def _make_r(n):
        return lambda self, other: getattr(np.asarray(self._v()), f"__r{n}__")(other)
    setattr(_Lazy, f"__r{_name}__", _make_r(_name))
_Lazy.__neg__ = lambda self: -np.asarray(self._v())
_Lazy.__abs__ = lambda self: abs(np.asarray(self._v()))
_Lazy.__hash__ = lambda self: id(self)
import numbers as _numbers

_numbers.Real.register(_LazyScalar)


class _HostCopy:
    """One device tensor, copied to the host (float64 numpy) at most once per content.  The copy is ASYNCHRONOUS into a cached pinned
    buffer (round 6; one buffer per thread and shape): start() enqueues it behind the launch that produced the tensor and records an
    event, get() waits for that event only -- the blocking pageable-memory .cpu() of round 5 cost 0.3-4 ms per step depending on the
    box's host (BENCH_r05: 5.9 ms eager against 1.95 with lazy metrics), this costs the wait for the GPU to reach the event."""

    def __init__(self, t):
        self.t = t
        self.cache = None
        self._means = None
        self._pending = None  # (pinned buffer, event) of a copy in flight

    def start(self):
        """Enqueue the device-to-host copy now (no-op while the stream is being captured, or when a copy is cached / in flight)."""
        if self.cache is not None or self._pending is not None or torch.cuda.is_current_stream_capturing():
            return
        t = self.t.detach()
        pool = _state.__dict__.setdefault("pinned", {})
        key = (tuple(t.shape), t.dtype)
        slot = pool.get(key)
        if slot is None:
            if len(pool) > 8:
                pool.clear()
            slot = pool[key] = (torch.empty(t.shape, dtype=t.dtype, pin_memory=True), torch.cuda.Event())
        buf, ev = slot
        with ops._on(t.device):
            buf.copy_(t, non_blocking=True)
            ev.record(torch.cuda.current_stream(t.device))
        self._pending = slot

    def get(self):
        if torch.cuda.is_current_stream_capturing():
            raise _lib.DfepeError("a host-side metric of get_Rt_loss was read while its stream is being captured into a graph")
        if self.cache is None:
            self.start()
            buf, ev = self._pending
            ev.synchronize()
            self.cache = buf.numpy().astype(np.float64)  # a copy: the pinned buffer serves the next call
            self._pending = None
        return self.cache

    def means(self):
        """[2, L]: the batch mean of every (kind, layer) row, one vectorised reduction per host copy."""
        if self._means is None or self.cache is None:
            self._means = self.get().mean(axis=2)
        return self._means

    def refresh(self):
        """Forget the host copy (after a graph replay rewrote the device buffer in place)."""
        self.cache = None
        self._means = None
        self._pending = None


class _GeoErrors(dict):
    """get_Rt_loss's dict (exactly the reference's twelve keys) with one attribute: host_metrics, the pending host copy of the
    angular errors (None when they were copied at once)."""

    host_metrics = None
    _LAZY_KEYS = ("R_angle_error_mean", "t_angle_error_mean", "R_angle_error_list", "t_angle_error_list",
                  "R_angle_error_layers_list", "t_angle_error_layers_list")

    def realise(self):
        """Replace every lazy value by the reference's own type (python float / numpy array / list of numpy arrays), read from
        the device buffer as it is now: one device-to-host copy.  Returns self.  A capturing caller runs this after a replay
        (host_metrics.refresh() first) when it wants to log or pickle the dict."""
        self.update(self._realised_items())
        return self

    def _realised_items(self):
        out = {}
        for k in self._LAZY_KEYS:
            v = self.get(k)
            if isinstance(v, list):
                out[k] = [np.asarray(x._v()) if isinstance(x, _Lazy) else x for x in v]
            elif isinstance(v, _LazyScalar):
                out[k] = float(v._v())
            elif isinstance(v, _Lazy):
                out[k] = np.asarray(v._v())
        return out

    def realised(self):
        """A NEW dict with the reference's host types, this one left lazy: what a replayed graph's (static) dict needs -- it is
        rewritten by the next replay and must keep following the device buffer."""
        new = _GeoErrors(self)
        new.update(self._realised_items())
        new.host_metrics = None
        return new


_dense_T_cache = {}


def _dense_T(T, B):
    """[B,3,3] contiguous copy of an expanded (stride-0) image-size transform, made once per (buffer, B): DeepFNet hands out
    views of one cached constant, and a copy kernel per step for it would be a launch spent on a constant."""
    if T.stride(0) != 0:
        return T if T.is_contiguous() else T.contiguous()
    key = (T.data_ptr(), int(B), str(T.device))
    hit = _dense_T_cache.get(key)
    if hit is None or hit[0]._version != T._version:
        if len(_dense_T_cache) > 16:
            _dense_T_cache.clear()
        hit = (T, T.expand(B, 3, 3).contiguous())  # T itself is kept: its memory cannot be recycled under the key while the entry lives
        _dense_T_cache[key] = hit
    return hit[1]


class _PerThread(threading.local):
    """The fused tail of this thread's latest get_all_loss_DeepF call that was given the ground truth (loss_params["pose_gt"]);
    per thread because nn.DataParallel runs one python thread per GPU through these functions (train_good.py:311-312)."""

    def __init__(self):
        self.tail = {}


_state = _PerThread()


def get_all_loss_DeepF(outs, pts1_virt_ori, pts2_virt_ori, Ks, loss_params, get_residual_summaries=True):
    """F-loss on the virtual correspondences for every layer + E-from-F.  Returns, like the reference (:511-519),
    (losses_dict, E_ests, F_ests, logits_softmax, residual_norm_layers, residual_norm_max_layers, E_ests_layers).

    ONE kernel launch for all layers plus ONE for every batch mean / minimum of the returned dict (dfepe_loss_stats); per-layer
    lists are rows of one buffer (ops.stack_rows / unstack_rows: no torch.stack copies, no SelectBackward fills in the backward).  Two keys beyond the
    reference's loss_params, both optional:
      "pose_gt": (qs_cam, ts_cam, delta_Rtijs_4_4) -- the ground truth that the get_Rt_loss call right after this one will be
                 given.  With it the pose errors are formed in the SAME launch (dfepe_loss_tail_jac) and get_Rt_loss finds
                 them ready; without it get_Rt_loss launches the pose kernel itself.  Same numbers either way.
      "floss_grad": False -- promise that no gradient will be asked of loss_F / loss_layers (an objective of the pose terms
                 only, the reference's if_qt_loss, Train_model_pipeline.py:580-587): skips the F-loss adjoint work of the launch."""
    if loss_params.get("if_tri_depth", False) or loss_params.get("if_sample_loss", False):
        raise NotImplementedError("if_tri_depth / if_sample_loss are outside the built hot path (all shipped configs disable them)")
    logits_softmax = outs["weights"]
    F_est_normalized, T1, T2 = outs["F_est"], outs["T1"], outs["T2"]
    out_layers, residual_layers, weights_layers = outs["out_layers"], outs["residual_layers"], outs["weights_layers"]
    depth = loss_params["depth"]
    F_layers = ops.stack_rows(out_layers[:depth])  # [L,B,3,3]; no copy when DeepFNet.forward wrote the layers into one buffer
    M = pts1_virt_ori.shape[1]
    B = F_layers.shape[1]
    gt = loss_params.get("pose_gt")
    _last_tail = _state.tail = {}
    # loss_epi_res (:429-438, logged only): sum_n epi_res * weights of every (layer, pair) in one launch; its means ride in the
    # statistics launch below
    epidot, n_epi, N_pts = None, 0, 1
    if depth > 1:
        n_epi = min(len(outs["epi_res_layers"]), len(outs["weights_layers"]))  # zip() of the reference
        if n_epi > 0:
            epi = ops.stack_rows(outs["epi_res_layers"][:n_epi])  # [n,B,1,N], strided views when the layers share a buffer
            w = ops.stack_rows(outs["weights_layers"][:n_epi])
            N_pts = epi.shape[-1]
            epidot = ops.row_dot(epi.squeeze(2) if epi.dim() == 4 else epi, w.squeeze(2) if w.dim() == 4 else w)  # [n,B]
    if gt is not None and M <= 112:
        q_gt, t_gt, delta = (torch.as_tensor(x).to(F_layers.device) for x in gt)
        R_gt = ops.camera_rotation(delta)
        r = ops.loss_tail_jac(F_layers, T1, T2, Ks, pts1_virt_ori, pts2_virt_ori, loss_params["clamp_at"], q_gt, t_gt, R_gt,
                              floss_grad=loss_params.get("floss_grad", True), extra=epidot, extra_scale=1.0 / N_pts)
        E_layers, m_loss, o_loss, row_min, col_min, m_epi, o_epi = (r[k] for k in ("E_layers", "m_loss", "o_loss", "row_min", "col_min", "m_extra", "o_extra"))
        # the caller's own objects and their device copies are HELD until get_Rt_loss consumes the entry: an address or an id()
        # recycled for other values in between can then not pass for "the same ground truth"
        _last_tail.update(E=E_layers, gt_src=tuple(gt), gt_dev=(q_gt, t_gt, delta), **{k: r[k] for k in ("q_l2", "t_l2", "ang", "m_q", "o_q", "m_t", "o_t")})
    else:
        # without the ground truth the pose part cannot ride along: the stand-alone F-loss kernel, whose adjoint takes the
        # gradient w.r.t. the E matrices that get_Rt_loss's pose kernel sends back (any number M of virtual points)
        loss_sum, E_layers = ops.floss(F_layers, T1, T2, Ks, pts1_virt_ori, pts2_virt_ori, loss_params["clamp_at"])
        ((m_loss, o_loss), _, _, (m_epi, o_epi)), row_min, col_min = ops.loss_stats(
            [(loss_sum, 1.0 / M), None, None, None if epidot is None else (epidot, 1.0 / N_pts)], want_min=True)
    loss_layers = list(m_loss.unbind(0))  # losses.mean() per layer (:343-354)
    loss_F_all = o_loss                   # sum(loss_layers) / len(loss_layers) (:364)
    E_ests_layers = list(ops.unstack_rows(E_layers))
    if _last_tail:
        # get_Rt_loss takes the fused launch's errors only for THESE row objects (not for detached copies or rows rebuilt over the
        # same memory: those must not inherit a graph to F); and when the rows die unconsumed the entry -- which holds the step's
        # autograd graph -- goes with them instead of waiting for the next call
        _last_tail["rows"] = [weakref.ref(r) for r in E_ests_layers]
        weakref.finalize(E_ests_layers[0], _last_tail.clear)
    same_T = T1 is T2 or (T1.data_ptr() == T2.data_ptr() and T1.stride() == T2.stride() and T1.shape == T2.shape)
    if same_T and T1.dim() == 3:
        F_ests = ops.congruence_diff(F_est_normalized, _dense_T(T1, B))
    else:
        F_ests = T2.permute(0, 2, 1) @ F_est_normalized @ T1
    if len(out_layers) >= depth and F_est_normalized is out_layers[depth - 1]:
        E_ests = E_ests_layers[-1]  # K^T T2^T F_est T1 K is exactly the last layer's E (:356-358 vs :366-369)
    else:
        E_ests = ops.congruence_diff(F_ests, Ks)
    losses_dict = {
        "loss_layers": loss_layers,
        "loss_F": loss_F_all,
        "loss_min_layers": row_min,  # per_pair.min(dim=1)[0] (:375); these two carry no gradient here (logged only)
        "loss_min_batch": col_min,   # per_pair.min(dim=0)[0] (:376)
    }
    loss_epi_res_all = 0.0
    loss_epi_res_layers = []
    if m_epi is not None:
        loss_epi_res_layers = list(m_epi.unbind(0))
        loss_epi_res_all = o_epi
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


def _same_rows(entry, given):
    refs = entry.get("rows", ())
    return len(refs) == len(given) and all(ref() is g for ref, g in zip(refs, given))


def _same_gt(entry, given, on_device):
    """Is the ground truth handed to get_Rt_loss the one get_all_loss_DeepF's fused launch was given?  Yes when they are the very
    same objects, or device tensors over the same memory with the same layout as the (still held) ones of that launch."""
    if all(a is b for a, b in zip(entry["gt_src"], given)):
        return True
    return all(torch.is_tensor(g) and g.is_cuda and d.data_ptr() == k.data_ptr() and d.shape == k.shape and d.stride() == k.stride() and d.dtype == k.dtype
               for g, d, k in zip(given, on_device, entry["gt_dev"]))


def get_Rt_loss(E_ests_layers, Ks_cpu, x1_cpu, x2_cpu, delta_Rtijs_4_4_cpu, qs_cam, ts_cam, device="cpu", lazy_host_metrics=None):
    """Pose loss from the per-layer essential matrices and the ground-truth camera motion.  Same 12-key dict as the
    reference (:272-293).  NB the reference stacks the *translation* list under "q_l2_error_list" (:276); that slip
    is reproduced so downstream logging sees identical values.  Ks/x1/x2 are unused, as in the reference.

    One launch for all layers and pairs (none when get_all_loss_DeepF was given "pose_gt" and these are its E matrices), one
    reduction for all means; the per-layer lists are rows of one buffer each, so the caller's
    clamp(stack(q_l2_error_layers_list)).mean() (Train_model_pipeline.py:580-586) costs its own three kernels and nothing
    more, forward or backward.  The angular errors (host-side numpy / floats in the reference) are copied to the host when first
    read when LAZY_HOST_METRICS (or the keyword ``lazy_host_metrics``, which overrides the module default for this call: what
    compat.CapturedStep passes instead of flipping the global under other threads' feet) is set; by default they are the reference's
    numpy arrays / floats on return, copied through a pinned buffer while this function builds its lists."""
    E_ests_layers = list(E_ests_layers)
    E_layers = ops.stack_rows(E_ests_layers)  # [L,B,3,3]; the very buffer get_all_loss_DeepF filled when the rows are its own
    if not E_layers.is_cuda:
        raise _lib.DfepeError("get_Rt_loss: E_ests_layers must live on the GPU")
    dev = E_layers.device
    q_gt, t_gt = torch.as_tensor(qs_cam).to(dev), torch.as_tensor(ts_cam).to(dev)
    delta = torch.as_tensor(delta_Rtijs_4_4_cpu).to(dev)
    lt = _state.tail
    L = E_layers.shape[0]
    if lt and _same_rows(lt, E_ests_layers) and _same_gt(lt, (qs_cam, ts_cam, delta_Rtijs_4_4_cpu), (q_gt, t_gt, delta)):
        q_l2, t_l2, ang, m_q, o_q, m_t, o_t = (lt[k] for k in ("q_l2", "t_l2", "ang", "m_q", "o_q", "m_t", "o_t"))  # get_all_loss_DeepF's launches
        _state.tail = {}  # consumed: do not keep the step's graph alive until the next call
    else:
        R_gt = ops.camera_rotation(delta)
        _, q_l2, t_l2, ang, _ = ops.pose_errors_packed(E_layers, q_gt, t_gt, R_gt)
        ((m_q, o_q), (m_t, o_t), _, _), _, _ = ops.loss_stats([(q_l2, 1.0), (t_l2, 1.0)])
    host = _HostCopy(ang)
    lazy = LAZY_HOST_METRICS if lazy_host_metrics is None else bool(lazy_host_metrics)
    if not lazy:
        host.start()  # the copy travels while the lists below are built; realise() at the end waits for its event only
    t_l2_layers = list(ops.unstack_rows(t_l2))
    q_l2_layers = list(ops.unstack_rows(q_l2))
    R_layers = [_Lazy(lambda i=i: host.get()[0, i]) for i in range(L)]
    t_layers = [_Lazy(lambda i=i: host.get()[1, i]) for i in range(L)]
    R_list = _Lazy(lambda: host.means()[0].copy())   # np.array([layer.mean() for layer in ...]) (:288-291)
    tA_list = _Lazy(lambda: host.means()[1].copy())
    R_mean = _LazyScalar(lambda: mean_list([float(v) for v in host.means()[0]]))
    tA_mean = _LazyScalar(lambda: mean_list([float(v) for v in host.means()[1]]))
    out = _GeoErrors({
        "t_l2_error_mean": o_t,      # mean_list of the per-layer means (:272-273)
        "q_l2_error_mean": o_q,
        "t_l2_error_list": m_t,
        "q_l2_error_list": m_t,      # sic: reference train_good_utils.py:276
        "R_angle_error_mean": R_mean,
        "R_angle_error_list": R_list,
        "t_angle_error_mean": tA_mean,
        "t_angle_error_list": tA_list,
        "R_angle_error_layers_list": R_layers,
        "t_angle_error_layers_list": t_layers,
        "t_l2_error_layers_list": t_l2_layers,
        "q_l2_error_layers_list": q_l2_layers,
    })
    out.host_metrics = host  # .refresh() after a graph replay rewrote the device buffer
    if not lazy and not torch.cuda.is_current_stream_capturing():
        out.realise()
    return out


_warned_opencv = False


def val_rt(idx, K_np, x1_single_np, x2_single_np, E_est_np, E_gt_np, F_est_np, F_gt_np, delta_Rtijs_4_4_cpu_np, five_point,
           if_opencv=True):
    """Same call and same 9-tuple as the reference's per-pair validation worker (train_good_utils.py:553-646):
    (error_Rt_estW, epi_dist_mean_estW, error_Rt_opencv, epi_dist_mean_opencv, error_Rt_gt, epi_dist_mean_gt, idx, M_estW, M_opencv).
    The estimated-E and ground-truth-E legs go through utils_F.goodCorr_eval_nondecompose (the cheirality kernel in place of
    cv2.recoverPose) and utils_F.epi_distance_np.  The OpenCV five-point / eight-point RANSAC baseline (``if_opencv``,
    utils_opencv.recover_camera_opencv) is outside this build (OpenCV baselines): its three slots are None.
    ``if_opencv`` keeps the reference's default (True); a caller that asks for the OpenCV leg -- explicitly or by that default --
    gets one warning saying that its slots stay None, instead of an unrelated TypeError when it indexes them later.
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
