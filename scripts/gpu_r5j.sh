#!/bin/bash
cd "$(dirname "$0")/.."
timeout 200 python - <<'PY'
import importlib, torch
d = importlib.import_module("pytorch-deepfepe_amd")
DEV="cuda:0"
for N in (100, 100, 128):
    B = 16384
    sc = d.synth.make_scene(B, N, seed=5, outlier_ratio=0.2, noise_px=0.5)
    m = sc["matches_xy_ori"].to(DEV).contiguous(); lg = sc["logits_layers"][0].to(DEV).contiguous()
    junk = torch.full((B, 128), 7.0, device=DEV); del junk
    big = d.ops.w8pt_forward(m, None, lg, True, 1241.0, 376.0, 0.5, True, True, logits=True)
    for c in range(0, B, 4096):
        part = d.ops.w8pt_forward(m[c:c+4096].contiguous(), None, lg[c:c+4096].contiguous(), True, 1241.0, 376.0, 0.5, True, True, logits=True)
        for k, (x, y) in enumerate(zip(big, part)):
            a, b = x[c:c+4096], y
            neq = (a != b) & ~(torch.isnan(a) & torch.isnan(b))
            if neq.any():
                idx = neq.nonzero()
                print(N, "chunk", c, "output", k, "differs at", int(neq.sum()), "entries; columns", sorted(set(idx[:, -1].tolist()))[:20], "rows", idx[:3, 0].tolist(), "max abs", float((a - b).abs().nan_to_num().max()))
    print(N, "done")
PY
