"""Mirror of deepFEPE/models/DeepFNet.py for the default training path: NormalizeAndExpand_HW (:93-120),
Fit (:123-295) and DeepFNet (:299-554).  The solver arithmetic runs in libdfepe_hip.so; the Python here only
orchestrates the recurrent loop exactly like DeepFNet.forward (:429-554)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib, ops
from .ErrorEstimators import ErrorEstimator, FusedErrorEstimator


def _require_gpu(t, what):
    if not t.is_cuda:
        raise _lib.DfepeError(f"{what}: tensors must live on the GPU (there is no CPU implementation in this package)")


class NormalizeAndExpand_HW(nn.Module):
    """pixel (x,y) -> [-1,1]^2 homogeneous; returns pts1, pts2 [B,3,N] and T1, T2 [B,3,3] (DeepFNet.py:108-120).
    Inside DeepFNet.forward this map is fused into the solver kernel; the module exists for API parity."""

    def __init__(self, image_size, is_cuda=True, is_test=False):
        super().__init__()
        self.H, self.W = image_size[0], image_size[1]

    def normalize(self, pts):
        _require_gpu(pts, "NormalizeAndExpand_HW")
        T = torch.tensor([[2.0 / self.W, 0.0, -1.0], [0.0, 2.0 / self.H, -1.0], [0.0, 0.0, 1.0]], device=pts.device,
                         dtype=pts.dtype).unsqueeze(0).expand(pts.size(0), -1, -1)
        ones = torch.ones(pts.size(0), pts.size(1), 1, device=pts.device, dtype=pts.dtype)
        return T @ torch.cat((pts, ones), 2).permute(0, 2, 1), T

    def forward(self, pts):
        pts1, T1 = self.normalize(pts[:, :, :2])
        pts2, T2 = self.normalize(pts[:, :, 2:])
        return pts1, pts2, T1, T2


class Fit(nn.Module):
    """Weighted normalised 8-point fit.  forward(pts1[B,N,3], pts2[B,N,3], weights[B,1,N]) -> (out[B,3,3], residual[B,N]).

    ``if_cpu_svd`` is accepted and ignored (it selected a per-sample CPU LAPACK round trip in the reference,
    DeepFNet.py:219-230; both of its branches compute the same thing).  Gradients flow to all three tensor inputs
    (the point gradients are only computed when pts1/pts2 require grad)."""

    def __init__(self, is_cuda=True, is_test=False, if_cpu_svd=False, normalize_SVD=True):
        super().__init__()
        self.normalize_SVD = normalize_SVD  # False: rows of X are w_i p_i instead of w_i p_i / |p_i| (DeepFNet.py:211-212)
        self.if_cpu_svd = if_cpu_svd
        self.is_cuda = is_cuda

    def normalize(self, pts, weights):
        """Weighted Hartley normalisation, pts [B,N,3], weights [B,N,1] -> (T pts^T [B,3,N], T [B,3,3]) with the literal scale
        1.4142 (DeepFNet.py:148-179).  weighted_svd calls it with unit weights (:198-199), which is what the fit kernel
        fuses; this method serves callers of the reference API and is plain elementwise torch."""
        denom = weights.sum(1)
        c = (pts * weights).sum(1) / denom
        d = pts - c.unsqueeze(1)
        meandist = ((weights * d[:, :, :2].pow(2).sum(2).sqrt().unsqueeze(2)).sum(1) / denom).squeeze(1)
        scale = 1.4142 / meandist
        T = torch.zeros(pts.shape[0], 3, 3, dtype=pts.dtype, device=pts.device)
        T[:, 0, 0] = scale
        T[:, 1, 1] = scale
        T[:, 2, 2] = 1
        T[:, 0, 2] = -c[:, 0] * scale
        T[:, 1, 2] = -c[:, 1] * scale
        return torch.bmm(T, pts.permute(0, 2, 1)), T

    def weighted_svd(self, pts1, pts2, weights, if_print=False):
        _require_gpu(weights, "Fit")
        out, residual = ops.w8pt(pts1, pts2, weights, normalize_rows=self.normalize_SVD)
        return out, residual

    def forward(self, pts1, pts2, weights, if_print=False, matches_good_unique_num=None):
        return self.weighted_svd(pts1, pts2, weights, if_print=if_print)


class DeepFNet(nn.Module):
    """Recurrent weight-estimation / fit loop.  Same constructor and ``forward(data_batch) -> dict`` contract as the
    reference (DeepFNet.py:300,429-554; dict keys :534-548).  Built: the default branch every shipped config uses plus
    if_quality, if_img_w and if_learn_offsets (the gradient reaches the offsets through the solver's d/d(matches));
    descriptors, triangulated depth and the deprecated GoodCorresNet architecture raise."""

    def __init__(self, depth, image_size, if_quality, if_img_w=False, if_goodCorresArch=False, if_tri_depth=False,
                 if_learn_offsets=False, if_des=False, des_size=None, quality_size=0, is_cuda=True, is_test=False,
                 if_cpu_svd=False, **params):
        super().__init__()
        for flag, name in ((if_goodCorresArch, "if_goodCorresArch"), (if_tri_depth, "if_tri_depth"), (if_des, "if_des")):
            if flag:
                raise NotImplementedError(f"DeepFNet({name}=True) is outside the built hot path (SURVEY.md §8)")
        if not if_quality:
            quality_size = 0
        self.if_quality = if_quality
        self.if_img_w = if_img_w
        self.image_size = image_size
        self.depth = depth
        # same parameters / state_dict as the reference's ErrorEstimator; evaluated on the bf16 matrix cores with fp32-accurate
        # split operands, InstanceNorm + LeakyReLU in the GEMM epilogue (csrc/est_gemm.hip; other variants: fp32 library GEMMs)
        Est = FusedErrorEstimator if params.get("fused_estimator", True) else ErrorEstimator
        self.input_weights = Est(4 + quality_size)
        self.update_weights = Est(4 + quality_size + 3)  # + weights, epi_res, residual (DeepFNet.py:340)
        self.if_learn_offsets = if_learn_offsets
        if if_learn_offsets:  # (DeepFNet.py:341-342): per-correspondence pixel offsets, no batch norm
            self.update_offsets = Est(4 + quality_size + 3, output_size=4, if_bn=False)
        if is_test:
            self.input_weights.eval()
            self.update_weights.eval()
            if if_learn_offsets:
                self.update_offsets.eval()
        self.norm_HW = NormalizeAndExpand_HW(image_size, is_cuda, is_test)
        self.fit = Fit(is_cuda, is_test, if_cpu_svd)

    def _T_hw(self, B, dev):
        """The image-size transform as [B,3,3] (an expanded view of one cached device constant: no per-step host copy, so the
        forward can be captured in a hipGraph)."""
        key = (str(dev), float(self.image_size[0]), float(self.image_size[1]))
        cache = self.__dict__.setdefault("_t_hw_cache", {})
        if key not in cache:
            H, W = float(self.image_size[0]), float(self.image_size[1])
            cache[key] = torch.tensor([[2.0 / W, 0.0, -1.0], [0.0, 2.0 / H, -1.0], [0.0, 0.0, 1.0]], device=dev)
        return cache[key].unsqueeze(0).expand(B, -1, -1)

    def _inputs(self, data_batch, offsets=None, recurrent_copies=0):
        """get_input plus, on the default path, the channel-major buffers of the later estimator calls (ops.deepf_input)."""
        pts = data_batch["matches_xy_ori"]
        if offsets is None and not (torch.is_grad_enabled() and pts.requires_grad):
            # the default path: ONE launch instead of ~15 elementwise / bmm ones (the matches are data, nothing to differentiate)
            quality = data_batch["quality"] if self.if_quality else None
            weight_in, pts1, pts2, stores = ops.deepf_input(pts, float(self.image_size[1]), float(self.image_size[0]), quality,
                                                            recurrent_copies=recurrent_copies)
            T = self._T_hw(pts.shape[0], pts.device)
            return weight_in, pts1, pts2, T, T, pts, stores
        if offsets is not None:  # (DeepFNet.py:369-373)
            pts = pts + offsets.permute(0, 2, 1)
        pts1, pts2, T1, T2 = self.norm_HW(pts)
        pts1 = pts1.permute(0, 2, 1)
        pts2 = pts2.permute(0, 2, 1)
        parts = [(pts1[:, :, :2] + 1) / 2, (pts2[:, :, :2] + 1) / 2]
        if self.if_quality:
            parts.append(data_batch["quality"])
        weight_in = torch.cat(parts, 2).permute(0, 2, 1)
        return weight_in, pts1, pts2, T1, T2, pts, None

    def get_input(self, data_batch, offsets=None, iter=None):
        return self._inputs(data_batch, offsets)[:6]

    def _fit(self, matches, logits, data_batch, want_epi, dst=None):
        """logits [B,1,N] -> (out, residual[, epi], weights_prod [B,1,N]).  The softmax over N is fused into the solver
        kernel unless per-correspondence image weights multiply it afterwards (if_img_w).  ``dst``: rows of the per-layer
        stacks the outputs are written into (ops.w8pt_forward), so that the loss functions find every layer's F, epipolar
        residual and weights in one buffer each (no torch.stack copies, train_good_utils.get_all_loss_DeepF)."""
        H, W = float(self.image_size[0]), float(self.image_size[1])
        if self.if_img_w:
            weights_prod = F.softmax(logits, dim=2) * data_batch["weights_im"]
            return ops.w8pt_raw(matches, weights_prod, W, H, clamp_at=0.5, want_epi=want_epi) + (weights_prod,)
        outs = ops.w8pt_raw_logits(matches, logits, W, H, clamp_at=0.5, want_epi=want_epi, dst=dst)
        return outs[:-1] + (outs[-1].unsqueeze(1),)

    def forward(self, data_batch):
        # update_weights is evaluated depth - 1 times on the same parameters (:510): they are packed and split into planes once
        shared = getattr(self.update_weights, "shared_parameters", None)
        if shared is None or self.depth < 3:
            return self._forward(data_batch)
        with shared():
            return self._forward(data_batch)

    def _forward(self, data_batch):
        matches = data_batch["matches_xy_ori"]
        _require_gpu(matches, "DeepFNet")
        plain = not self.if_learn_offsets and not self.if_img_w
        pts_normalized_in, pts1, pts2, T1, T2, _, stores = self._inputs(data_batch, recurrent_copies=(self.depth - 1) if plain else 0)
        logits = self.input_weights(pts_normalized_in)
        _ = data_batch["matches_good_unique_nums"]  # read like the reference does (DeepFNet.py:449,453)
        _ = data_batch["t_scene_scale"]

        out_layers, epi_res_layers, residual_layers = [], [], []
        weights_layers, logits_layers = [], [logits]
        offsets_accu = None
        # Where the per-layer outputs live.  F of every layer: one [L,B,3,3] buffer (the loss functions take it as it is).  With the
        # channel-major estimator inputs of ops.deepf_input (``stores`` [L-1,C,B,N]) the weights, epipolar residual and residual
        # of layer l are written straight into channels c0.. of stores[l]: the next estimator call reads its input without a
        # torch.cat, and the same channel of consecutive layers is a strided stack for the loss functions.
        B, N, dev = matches.shape[0], matches.shape[1], matches.device
        F_stack = torch.empty(self.depth, B, 3, 3, device=dev)
        c0 = pts_normalized_in.shape[1]

        def dst(l):
            d = {"F": (F_stack, l)}
            if stores is not None and l < self.depth - 1:
                d.update({"weights": (stores[l], c0), "epi": (stores[l], c0 + 1), "residual": (stores[l], c0 + 2)})
            return d

        for it in range(self.depth - 1):
            out, residual, epi, weights_prod = self._fit(matches, logits, data_batch, True, dst(it))
            weights_layers.append(weights_prod)
            out_layers.append(out)
            residual_layers.append(residual)
            epi_res = epi.unsqueeze(1)
            epi_res_layers.append(epi_res)
            rec = [weights_prod.squeeze(1), epi, residual]
            if stores is not None and all(r.data_ptr() == stores[it][c0 + k].data_ptr() and r.is_contiguous() for k, r in enumerate(rec)):
                net_in = ops.estimator_input(stores[it], c0, rec)  # the three rows are where the fit kernel left them
            else:
                net_in = torch.cat((pts_normalized_in, weights_prod, epi_res, residual.unsqueeze(1)), 1)
            if self.if_learn_offsets:  # (DeepFNet.py:490-507): the later fits see the corrected matches
                offsets_accu = self.update_offsets(net_in)
                pts_normalized_in, pts1, pts2, T1, T2, matches = self.get_input(data_batch, offsets_accu, it)
                net_in = torch.cat((pts_normalized_in, weights_prod, epi_res, residual.unsqueeze(1)), 1)
            logits = self.update_weights(net_in)
            logits_layers.append(logits)
        out, residual, weights_prod = self._fit(matches, logits, data_batch, False, dst(self.depth - 1))
        weights_layers.append(weights_prod)
        residual_layers.append(residual)
        out_layers.append(out)
        preds = {
            "logits": logits.squeeze(1),
            "logits_layers": logits_layers,
            "F_est": out,
            "epi_res_layers": epi_res_layers,
            "T1": T1,
            "T2": T2,
            "out_layers": out_layers,
            "pts1": pts1,
            "pts2": pts2,
            "weights": weights_prod,
            "residual_layers": residual_layers,
            "weights_layers": weights_layers,
        }
        if self.if_learn_offsets:
            preds["offsets"] = offsets_accu
        return preds
