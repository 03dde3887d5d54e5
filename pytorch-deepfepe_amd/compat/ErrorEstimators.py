"""Per-correspondence weight estimator (mirror of deepFEPE/models/ErrorEstimators.py:14-68).

Stock PyTorch-ROCm (MIOpen / rocBLAS): SURVEY.md §8 row a18 keeps it outside the hand-written hot path.
The layer stack is restated so that ``state_dict`` keys (``fw.<idx>.weight`` ...) match the reference and its
checkpoints load unchanged."""
import contextlib

import torch
import torch.nn as nn


class ErrorEstimator(nn.Module):
    def __init__(self, input_size, output_size=1, if_bn=False):
        super().__init__()
        widths = [input_size, 64, 128, 1024, 512, 256]
        layers = []
        for cin, cout in zip(widths[:-1], widths[1:]):
            layers.append(nn.Conv1d(cin, cout, kernel_size=1, bias=True))
            if if_bn:
                layers.append(nn.BatchNorm1d(cout))
            layers.append(nn.InstanceNorm1d(cout, affine=True))
            layers.append(nn.LeakyReLU(inplace=True))
        # the reference's batch-norm variant drops the bias of the last convolution (ErrorEstimators.py:47)
        layers.append(nn.Conv1d(256, output_size, kernel_size=1, bias=not if_bn))
        self.fw = nn.Sequential(*layers)

    # MIOpen's backward for this stack fails (miopenStatusUnknownError) at 4096 pairs x 100 points on ROCm 7.2; every
    # layer acts per pair (k=1 convolutions, InstanceNorm1d), so the batch can be processed in independent chunks.
    max_chunk = 2048

    def forward(self, data):
        # BatchNorm statistics are over the batch: chunking would change them (and the running averages) in training mode
        batch_stats = self.training and any(isinstance(m, nn.BatchNorm1d) for m in self.fw)
        if data.shape[0] <= self.max_chunk or batch_stats:
            return self.fw(data)
        return torch.cat([self.fw(c) for c in data.split(self.max_chunk, dim=0)], dim=0)


class FusedErrorEstimator(ErrorEstimator):
    """Same parameters / state_dict as ErrorEstimator, MI355X-shaped evaluation ("next" row f-1 of SURVEY.md §8).

    Default (``split_bf16 = True``): the whole stack runs on the bf16 matrix cores with fp32-accurate split operands --
    csrc/est_gemm.hip through ``estimator.estimator_forward``: per layer ONE kernel (GEMM + InstanceNorm + LeakyReLU in its
    epilogue) forward at N = 100 points per pair, two (plain product, then normalisation + activation + split) at any other
    N >= 2 -- the SIFT configurations' 1000-2000 --; three backward (normalisation adjoint, weight-gradient GEMM, data-gradient
    GEMM).  ``split_bf16 = False`` keeps the native-fp32 evaluation described next:
    activations live channel-major as [C, B*N], every 1x1 convolution is ONE large GEMM W[C_out,C_in] @ X[C_in, B*N]
    (rocBLAS / hipBLASLt through torch.mm) instead of B small ones, and InstanceNorm + LeakyReLU is one fused HIP pass
    (ops.inorm_lrelu).  The biases of the convolutions that feed an InstanceNorm cancel in the normalisation and are
    skipped in the arithmetic; they stay in the autograd graph with the exact zero gradient the reference computes for them
    (so DistributedDataParallel sees every parameter used and the optimizer state matches).  The split-bf16 path serves any head width (1, or the 4 of ``update_offsets``); the batch-norm variant
    always takes the stock path, and so does the native-fp32 evaluation for N not a multiple of 4 or above 512."""

    split_bf16 = True
    _prepared = None  # estimator.Prepared of the current shared_parameters() scope

    def _stack(self):
        """(hidden, head, eps, slope) of the Conv1d -> InstanceNorm1d(affine) -> LeakyReLU stack, or None for another architecture."""
        mods = list(self.fw)
        if any(isinstance(m, nn.BatchNorm1d) for m in mods):
            return None
        hidden, i = [], 0
        while i + 2 < len(mods) and isinstance(mods[i + 1], nn.InstanceNorm1d):
            hidden.append((mods[i].weight, mods[i].bias, mods[i + 1].weight, mods[i + 1].bias))
            i += 3
        if not hidden or i != len(mods) - 1:
            return None
        head, inorm, act = mods[i], mods[1], mods[2]
        if not (act.negative_slope > 0 and all(m.affine for m in mods if isinstance(m, nn.InstanceNorm1d))):
            return None
        return hidden, (head.weight, head.bias), inorm.eps, act.negative_slope

    @contextlib.contextmanager
    def shared_parameters(self):
        """Scope in which every forward() of this module uses ONE preparation of its parameters (estimator.prepare: the packed
        parameter vector + the weights' 16-bit planes): for a caller that evaluates the estimator several times on unchanged
        parameters -- DeepFNet.forward's recurrent loop (deepFEPE/models/DeepFNet.py:510).  The parameters must not be modified
        inside the scope.  Per module OBJECT, so the replicas nn.DataParallel runs in threads each have their own."""
        from .. import estimator

        st = self._stack() if self.split_bf16 else None
        self._prepared = estimator.prepare(st[0], st[1]) if (st is not None and st[0][0][0].is_cuda) else None
        try:
            yield self
        finally:
            self._prepared = None

    def forward(self, data):
        from .. import estimator, ops

        B, C0, N = data.shape
        mods = list(self.fw)
        has_bn = any(isinstance(m, nn.BatchNorm1d) for m in mods)
        if self.split_bf16 and not has_bn and estimator.supported(data):
            st = self._stack()
            if st is not None:
                hidden, head, eps, slope = st
                return estimator.estimator_forward(data, hidden, head, eps=eps, slope=slope, prepared=self._prepared)
        if has_bn or (N % 4) or N > 512 or not data.is_cuda:
            return super().forward(data)
        x = data.permute(1, 0, 2).reshape(C0, B * N)  # channel-major
        i = 0
        while i < len(mods):
            conv = mods[i]
            W = conv.weight[:, :, 0]
            if i + 2 < len(mods) and isinstance(mods[i + 1], nn.InstanceNorm1d):
                inorm, act = mods[i + 1], mods[i + 2]
                y = torch.mm(W, x)  # bias cancels in the instance normalisation
                a = ops.inorm_lrelu(y.view(W.shape[0], B, N), inorm.weight, inorm.bias, inorm.eps, act.negative_slope,
                                    skipped_bias=conv.bias)
                x = a.view(W.shape[0], B * N)
                i += 3
            else:  # last convolution
                y = torch.mm(W, x)
                if conv.bias is not None:
                    y = y + conv.bias[:, None]
                x = y
                i += 1
        return x.view(-1, B, N).permute(1, 0, 2)
