#!/usr/bin/env python3
"""Golden vectors for the validation summary: the reference's own write_metrics_summary (deepFEPE/train_good_utils.py:758-856,
imported unmodified, see make_golden.py) run with a recording writer on seeded inputs; every scalar it logs is stored by tag.

    python tests/golden/make_golden_metrics.py      # rewrites tests/golden/metrics.npz (build container only)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


class Recorder:
    def __init__(self):
        self.scalars = {}

    def add_scalar(self, tag, value, n_iter):
        self.scalars[tag] = float(value)

    def add_histogram(self, tag, values, n_iter):
        pass


def main():
    mg.install_stubs()
    with mg.quiet():
        import train_good_utils as tgu
    import logging

    # `logging` reaches the reference through `from superpoint.utils.logging import *` (train_good_utils.py:40); the stubbed
    # superpoint package does not provide it
    tgu.logging = logging
    rng = np.random.default_rng(3)
    B, N, nb = 37, 60, 3  # three "batches" per list, like the validation loop appends them
    d = {"epi_dists": {}, "err_q": {}, "err_t": {}, "relative_scale": {}}
    for tag, scale in (("gt", 0.3), ("ours", 0.8), ("opencv", 2.0)):
        d["epi_dists"][tag] = [np.abs(rng.standard_normal((B, N))).astype(np.float32) * scale for _ in range(nb)]
        d["err_q"][tag] = [np.abs(rng.standard_normal(B)).astype(np.float32) * scale * 3 for _ in range(nb)]
        d["err_t"][tag] = [np.concatenate((np.abs(rng.standard_normal(B - 2)).astype(np.float32) * scale * 20, [90.0, 180.0])).astype(np.float32) for _ in range(nb)]
        d["relative_scale"][tag] = [rng.random(B).astype(np.float32) for _ in range(nb)]
    rec = Recorder()
    with mg.quiet():
        tgu.write_metrics_summary(rec, d, "val", 7)
    out = {}
    for m, per in d.items():
        for tag, lst in per.items():
            out[f"in_{m}_{tag}"] = np.stack(lst)
    tags = sorted(rec.scalars)
    out["tags"] = np.array(tags)
    out["values"] = np.array([rec.scalars[t] for t in tags])
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **out)
    with open(os.path.join(HERE, "MANIFEST.txt"), "a") as f:
        f.write(f"metrics.npz: {len(out)} arrays, {os.path.getsize(os.path.join(HERE, 'metrics.npz'))} bytes "
                f"(make_golden_metrics.py: the reference's write_metrics_summary with a recording writer, {len(tags)} scalars)\n")
    print(len(tags), "scalars recorded")


if __name__ == "__main__":
    main()
