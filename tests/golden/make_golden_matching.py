#!/usr/bin/env python3
"""Golden vectors for the match-construction row: tests/golden/matching.npz.

Runs only in the build container.  The reference's own ``get_matches_from_SP`` (train_good_utils.py:649-724) and
``crop_or_pad_choice`` (dsac_tools/utils_misc.py:139-161) are imported unmodified and executed; the two objects the
function receives from its caller are stand-ins, because their code is not in /root/reference:
  * ``net_SP`` / ``process_SP_output``: the SuperPoint front-end (out of scope) -> a fake that hands back the
    pre-made keypoints / descriptors / offsets stored in the fixture;
  * ``SP_tracker``: PointTracker of the un-vendored ``superpoint`` package -> an object whose ``nn_match_two_way``
    is the oracle's restatement of that published routine (so that function itself stays "parity unpinned";
    what IS pinned here is everything the reference does with its [3,n] result).
Only input/output arrays are written.

    python tests/golden/make_golden_matching.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import make_golden as mg  # noqa: E402  (stubs for cv2 / pebble / superpoint, sys.path of the reference)
from oracle import deepf_oracle as oracle  # noqa: E402


def make_descriptors(B, N, D, seed, frac_common=0.6, noise=0.25):
    g = torch.Generator().manual_seed(seed)
    d1 = torch.nn.functional.normalize(torch.randn(B, N, D, generator=g), dim=2)
    d2 = torch.nn.functional.normalize(torch.randn(B, N, D, generator=g), dim=2)
    n_common = int(frac_common * N)
    for b in range(B):
        perm = torch.randperm(N, generator=g)[:n_common]
        slots = torch.randperm(N, generator=g)[:n_common]
        scale = noise * (0.2 + 3.8 * torch.rand(n_common, 1, generator=g))  # distances from ~0.05 up to ~1.1
        d2[b, slots] = torch.nn.functional.normalize(d1[b, perm] + scale * torch.randn(n_common, D, generator=g) / D ** 0.5, dim=1)
    xs = [torch.randint(0, 1200, (B, N, 2), generator=g).float() for _ in range(2)]
    res = [torch.rand(B, N, 2, generator=g) - 0.5 for _ in range(2)]
    return xs, [d1, d2], res


class Tracker:
    def __init__(self, nn_thresh):
        self.nn_thresh = nn_thresh

    def nn_match_two_way(self, desc1, desc2, nn_thresh):
        return oracle.nn_match_two_way(desc1, desc2, nn_thresh)


def main():
    mg.install_stubs()
    import train_good_utils as tgu
    from dsac_tools import utils_misc

    out = {}
    # crop_or_pad_choice alone
    cases = [(50, 20, True), (20, 50, True), (37, 37, True), (10, 25, False), (30, 8, False)]
    out["cp_cases"] = np.array([(a, b, int(s)) for a, b, s in cases])
    for k, (a, b, s) in enumerate(cases):
        np.random.seed(100 + k)
        out[f"cp_choice_{k}"] = utils_misc.crop_or_pad_choice(a, b, shuffle=s)

    # the whole second half of get_matches_from_SP, two regimes: crop (many matches) and pad (few matches)
    for tag, (N, out_n, thr, seed) in {"crop": (96, 32, 1.0, 1), "pad": (64, 80, 0.7, 2)}.items():
        B, D = 2, 64
        xs, des, res = make_descriptors(B, N, D, seed)
        fake_outs = [{"pts_int": xs[i], "pts_desc": des[i], "pts_offset": res[i]} for i in range(2)]
        calls = iter(fake_outs)
        tgu.process_SP_output = lambda outs, proc: outs
        net = lambda img: next(calls)
        imgs = [torch.zeros(B, 8, 8), torch.zeros(B, 8, 8)]
        np.random.seed(7 + seed)
        with mg.quiet():
            r = tgu.get_matches_from_SP(imgs, net, None, Tracker(thr), out_num_points=out_n)
        for i in range(2):
            out[f"{tag}_pts{i}"] = xs[i].numpy(); out[f"{tag}_des{i}"] = des[i].numpy(); out[f"{tag}_res{i}"] = res[i].numpy()
        out[f"{tag}_cfg"] = np.array([N, out_n, thr, 7 + seed])
        out[f"{tag}_xs"] = r["xs"].numpy(); out[f"{tag}_offsets"] = r["offsets"].numpy()
        out[f"{tag}_quality"] = r["quality"].numpy(); out[f"{tag}_num_matches"] = r["num_matches"].numpy()
        out[f"{tag}_xs_SP0"] = r["xs_SP"][0].numpy()
        print(tag, "num_matches", r["num_matches"].tolist())
    np.savez_compressed(os.path.join(HERE, "matching.npz"), **out)
    print("matching.npz:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "matching.npz")), "bytes")


if __name__ == "__main__":
    main()
