"""Drop-in for the one class the hot path's callers take from the un-vendored ``superpoint`` package
(``superpoint.models.model_wrap.PointTracker``; deepFEPE/train_good.py:74,222, utils/eval_tools.py:664-672): only the
matcher is provided, which is all get_matches_from_SP and the evaluation scripts call on it."""
import numpy as np
import torch

from .. import _lib, ops


class PointTracker(object):
    def __init__(self, max_length=2, nn_thresh=0.7):
        if max_length < 2:
            raise ValueError("max_length must be greater than or equal to 2.")
        self.maxl = max_length
        self.nn_thresh = nn_thresh

    def nn_match_two_way(self, desc1, desc2, nn_thresh):
        """desc1 [D,N1], desc2 [D,N2] (numpy or torch, unit-norm columns) -> numpy float64 [3,n]: index in 1, index in 2,
        L2 distance of every mutual nearest-neighbour pair closer than ``nn_thresh``.  The matrices are staged on the GPU
        (the reference hands this method host arrays, train_good_utils.py:687-691); batches should call
        ``compat.train_good_utils.get_matches_from_SP`` / ``ops.nn_match_two_way`` instead, which never leave the device."""
        d1, d2 = torch.as_tensor(desc1), torch.as_tensor(desc2)
        assert d1.shape[0] == d2.shape[0]
        if d1.shape[1] == 0 or d2.shape[1] == 0:
            return np.zeros((3, 0))
        if nn_thresh < 0.0:
            raise ValueError("'nn_thresh' should be non-negative")
        if not torch.cuda.is_available():
            raise _lib.DfepeError("PointTracker.nn_match_two_way needs a GPU (no CPU implementation)")
        dev = d1.device if d1.is_cuda else torch.device("cuda", torch.cuda.current_device())
        a = d1.to(dev, torch.float32).t().contiguous().unsqueeze(0)
        b = d2.to(dev, torch.float32).t().contiguous().unsqueeze(0)
        m1, m2, sc, cnt = ops.nn_match_two_way(a, b, float(nn_thresh))
        n = int(cnt[0].item())
        out = np.zeros((3, n))
        out[0] = m1[0, :n].cpu().numpy()
        out[1] = m2[0, :n].cpu().numpy()
        out[2] = sc[0, :n].cpu().numpy()
        return out
