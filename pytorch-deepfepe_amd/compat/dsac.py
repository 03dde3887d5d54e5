"""Mirror of deepFEPE/dsac_tools/dsac.py (class DSAC, :13-200): the RANSAC-style hypothesis loop over essential matrices.

The reference runs the loop on the CPU, one hypothesis at a time ("working on CPU because of many, small matrices",
dsac.py:110): sample 10 correspondences -> _E_from_XY -> soft inlier count from Sampson distances -> refinement by a
weighted _E_from_XY over all correspondences -> loss, and returns the per-correspondence average score of the hypotheses
each correspondence was sampled into (:197).  Here the `hyps` hypotheses are independent problems of the same kernels the
solver uses, so each stage is ONE launch over all hypotheses: the minimal-sample fits (ops.eight_point, B = hyps, N = 10),
the Sampson distances (ops.epi_metrics, hyps x N lanes), the weighted refinement fits (ops.eight_point, B = hyps, N = N).
Sampling uses Python's `random.sample` with the reference's call sequence (one call per hypothesis), so a seeded run draws
the same minimal sets.  The reference calls an undefined `utils_F.E_to_F` at :68 (only `_E_to_F` exists); its evident intent,
F = K^-T E K^-1, is what is computed.
"""
import random

import torch

from .. import _lib, ops


class DSAC:
    def __init__(self, hyps, inlier_thresh, inlier_beta, inlier_alpha, K, loss_function):
        self.hyps = hyps
        self.inlier_thresh = inlier_thresh
        self.inlier_beta = inlier_beta
        self.inlier_alpha = inlier_alpha
        self.loss_function = loss_function
        self.K = K

    def _normalized(self, P, Ki):
        """K^-1 applied to inhomogeneous points, _de_homo's 1e-10 guard included (utils_F.py:111-112)."""
        Ph = torch.cat((P, torch.ones(*P.shape[:-1], 1, device=P.device)), -1) @ Ki.t()
        return Ph[..., :2] / (Ph[..., 2:3] + 1e-10)

    def __call__(self, X, Y, H_gt=None):
        """X, Y [N,2] pixel correspondences -> [N,1] average soft-inlier score of the hypotheses each correspondence was
        drawn into (dsac.py:197).  Also sets best_H, best_H_idx, best_dists, best_corres_idx, max_score, inlier_scores
        [hyps,N], sampson_dists [hyps,N] like the reference."""
        if not (torch.is_tensor(X) and X.is_cuda):
            raise _lib.DfepeError("compat.dsac.DSAC: X, Y must live on the GPU (this mirror has no CPU loop)")
        X, Y = X.float(), Y.float()
        dev = X.device
        N, H = X.shape[0], self.hyps
        assert Y.shape[0] == N, "N mismatch between X and Y!"
        self.N = N
        K = torch.as_tensor(self.K, dtype=torch.float32, device=dev)
        Ki = torch.linalg.inv(K)
        idx = torch.tensor([random.sample(range(N), 10) for _ in range(H)], device=dev)  # one draw per hypothesis (:45)
        Xn, Yn = self._normalized(X, Ki), self._normalized(Y, Ki)
        # step 1: minimal-sample hypotheses, all at once
        E = ops.eight_point(Xn[idx].contiguous(), Yn[idx].contiguous(), None, essential=True)          # [H,3,3]
        # step 2: soft inlier count from the Sampson distances in pixel space
        F = Ki.t() @ E @ Ki
        Xe, Ye = X.unsqueeze(0).expand(H, N, 2).contiguous(), Y.unsqueeze(0).expand(H, N, 2).contiguous()
        d_s = ops.epi_metrics(1, F.contiguous(), Xe, Ye)                                                # [H,N]
        dists = 1.0 - torch.sigmoid(self.inlier_beta * (d_s - self.inlier_thresh))
        score = dists.sum(1)
        # step 3: refinement, weighted fit over all correspondences with W = diag(sqrt(dists)) (:92)
        Href = ops.eight_point(Xn.unsqueeze(0).expand(H, N, 2).contiguous(), Yn.unsqueeze(0).expand(H, N, 2).contiguous(),
                               torch.sqrt(dists).contiguous(), essential=True)
        self.inlier_scores, self.sampson_dists = dists, d_s
        # step 4: loss of every hypothesis (a user callable, like the reference's)
        self.hyp_losses = None
        if self.loss_function is not None:
            self.hyp_losses = torch.stack([torch.as_tensor(self.loss_function(Href[h], X, Y), dtype=torch.float32, device=dev).reshape(())
                                           for h in range(H)])
        self.hyp_scores = score
        best = int(torch.argmax(score).item())  # first maximum = the reference's strict '>' update (:168)
        self.max_score = float(score[best].item())
        self.best_H, self.best_H_idx, self.best_dists, self.best_corres_idx = Href[best], best, dists[best], idx[best].tolist()
        n_scores = torch.zeros(N, device=dev).index_add_(0, idx.reshape(-1), score.repeat_interleave(10))
        n_counts = torch.zeros(N, device=dev).index_add_(0, idx.reshape(-1), torch.ones(H * 10, device=dev))
        return (n_scores / (n_counts + 1e-10)).unsqueeze(1)
