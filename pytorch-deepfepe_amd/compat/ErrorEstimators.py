"""Per-correspondence weight estimator (mirror of deepFEPE/models/ErrorEstimators.py:14-68).

Stock PyTorch-ROCm (MIOpen / rocBLAS): SURVEY.md §8 row a18 keeps it outside the hand-written hot path.
The layer stack is restated so that ``state_dict`` keys (``fw.<idx>.weight`` ...) match the reference and its
checkpoints load unchanged."""
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
        if data.shape[0] <= self.max_chunk:
            return self.fw(data)
        return torch.cat([self.fw(c) for c in data.split(self.max_chunk, dim=0)], dim=0)
