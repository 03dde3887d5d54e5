"""Central differences of a smooth objective of compat.DeepFNet along random parameter directions against <gradient, direction>, for the fused and the
stock estimators and several step sizes.   python scripts/whole_model_fd_check.py [B [N [depth]]]"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 3
DEV = "cuda:0"
D = d.compat.DeepFNet
sc = d.synth.make_scene(B, N, seed=3, outlier_ratio=0.2, noise_px=0.5)
m = sc["matches_xy_ori"].to(DEV)
g = torch.Generator().manual_seed(8)
RF = [torch.randn(B, 3, 3, generator=g).to(DEV) for _ in range(depth)]
for fused in (True, False):
    net = D.DeepFNet(depth=depth, image_size=[376, 1241, 3], if_quality=False, fused_estimator=fused).to(DEV)
    d.synth.fill_params_deterministic(net, seed=5)
    with torch.no_grad():
        o0 = net({"matches_xy_ori": m, "matches_good_unique_nums": None, "t_scene_scale": None})
    sgn = [torch.sign((o * r).flatten(1).sum(1))[:, None, None] for o, r in zip(o0["out_layers"], RF)]

    def objective():
        outs = net({"matches_xy_ori": m, "matches_good_unique_nums": None, "t_scene_scale": None})
        return sum(((o / o.flatten(1).norm(dim=1)[:, None, None]) * r * s_).double().sum() for o, r, s_ in zip(outs["out_layers"], RF, sgn))

    net.zero_grad(set_to_none=True)
    objective().backward()
    params = list(net.parameters())
    grads = [p.grad.detach().clone() for p in params]
    for trial in range(2):
        gd = torch.Generator().manual_seed(100 + trial)
        dirs = [torch.randn(p.shape, generator=gd).to(DEV) * p.detach().abs().mean().clamp_min(1e-3) for p in params]
        analytic = sum(float((g_.double() * v.double()).sum()) for g_, v in zip(grads, dirs))
        line = f"B={B} N={N} depth={depth} {'fused' if fused else 'stock'} direction {trial}: <grad, v> = {analytic:.5e}; central differences"
        for h in (4e-3, 2e-3, 1e-3, 5e-4, 2.5e-4, 1e-4):
            vals = []
            for sg in (1.0, -1.0):
                with torch.no_grad():
                    for p, v in zip(params, dirs):
                        p.add_(v, alpha=sg * h)
                    vals.append(float(objective()))
                    for p, v in zip(params, dirs):
                        p.sub_(v, alpha=sg * h)
            line += f" h={h:g}: {(vals[0] - vals[1]) / (2 * h):.5e}"
        print(line)
