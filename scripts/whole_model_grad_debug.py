"""Distances between the parameter gradients of compat.DeepFNet on (a) the one-call-per-pass estimator path, (b) the per-launch host code,
(c) the stock PyTorch estimators -- same parameters, same batch, smooth objective.   python scripts/whole_model_grad_debug.py [B [N [depth]]]"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 3
DEV = "cuda:0"
D = d.compat.DeepFNet
net = D.DeepFNet(depth=depth, image_size=[376, 1241, 3], if_quality=False).to(DEV)
d.synth.fill_params_deterministic(net, seed=5)
ref = D.DeepFNet(depth=depth, image_size=[376, 1241, 3], if_quality=False, fused_estimator=False).to(DEV)
ref.load_state_dict(net.state_dict())
sc = d.synth.make_scene(B, N, seed=3, outlier_ratio=0.2, noise_px=0.5)
b = {k: sc[k].to(DEV) for k in ("matches_xy_ori",)}
g = torch.Generator().manual_seed(8)
RF = [torch.randn(B, 3, 3, generator=g).to(DEV) for _ in range(depth)]
RG = torch.randn(B, N, generator=g).to(DEV)


def step(model, which):
    model.zero_grad(set_to_none=True)
    outs = model({"matches_xy_ori": b["matches_xy_ori"], "matches_good_unique_nums": None, "t_scene_scale": None})
    if which == "F":
        sgn = [torch.sign((o.detach() * r).flatten(1).sum(1))[:, None, None] for o, r in zip(outs["out_layers"], RF)]
        obj = sum((o * r * s_).sum() for o, r, s_ in zip(outs["out_layers"], RF, sgn))
    elif which == "logits0":
        obj = (outs["logits_layers"][0].squeeze(1) * RG).sum()
    elif which == "logits_last":
        obj = (outs["logits_layers"][-1].squeeze(1) * RG).sum()
    else:
        obj = (outs["residual_layers"][-1].square() * RG).sum()
    obj.backward()
    return {n: (torch.zeros_like(p) if p.grad is None else p.grad.clone()) for n, p in model.named_parameters()}, [l.detach().clone() for l in outs["logits_layers"]]


for which in ("logits0", "logits_last", "res", "F"):
    d.estimator.USE_PASS = True
    ga, la = step(net, which)
    d.estimator.USE_PASS = False
    gb, lb = step(net, which)
    gc, lc = step(ref, which)
    rel = lambda x, y: float((x - y).norm() / y.norm().clamp_min(1e-30))
    names = [n for n in ga if float(gc[n].abs().max()) > 0 and not (n.endswith(".bias") and "fw." in n and float(ga[n].abs().max()) == 0)]
    wab = max(rel(ga[n], gb[n]) for n in names)
    wac = max(rel(ga[n], gc[n]) for n in names)
    wbc = max(rel(gb[n], gc[n]) for n in names)
    print(f"B={B} N={N} depth={depth} objective {which:12s}: worst parameter-gradient distance pass/per-launch {wab:.1e}, pass/stock {wac:.1e}, per-launch/stock {wbc:.1e};"
          f" logits pass/stock {max(float((x - y).abs().max()) for x, y in zip(la, lc)):.1e}, pass/per-launch {max(float((x - y).abs().max()) for x, y in zip(la, lb)):.1e}")
