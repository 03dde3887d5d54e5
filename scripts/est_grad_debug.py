import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
EE = d.compat.ErrorEstimators
for cin, B, seed in ((7, 5, 5), (7, 5, 6), (7, 5, 7), (7, 6, 8), (4, 6, 9)):
    stock = EE.ErrorEstimator(cin); d.synth.fill_params_deterministic(stock, seed=seed)
    fused = EE.FusedErrorEstimator(cin).cuda(); fused.load_state_dict(stock.state_dict())
    f32 = EE.FusedErrorEstimator(cin).cuda(); f32.load_state_dict(stock.state_dict()); f32.split_bf16 = False
    s32 = EE.ErrorEstimator(cin).cuda(); s32.load_state_dict(stock.state_dict())
    stock = stock.double()
    g = torch.Generator().manual_seed(1)
    x = torch.rand(B, cin, 100, generator=g); G = torch.randn(B, 1, 100, generator=g)
    xa = x.double().requires_grad_(True); (stock(xa) * G.double()).sum().backward()
    out = []
    for m in (fused, f32, s32):
        xb = x.cuda().requires_grad_(True); (m(xb) * G.cuda()).sum().backward()
        pa, pb = dict(stock.named_parameters()), dict(m.named_parameters())
        errs = {"x": float((xb.grad.cpu().double() - xa.grad).abs().max() / xa.grad.abs().max())}
        for n in pa:
            if pb[n].grad.abs().max() > 0:
                errs[n] = (float((pb[n].grad.cpu().double() - pa[n].grad).abs().max() / pa[n].grad.abs().max()),
                           float((pb[n].grad.cpu().double() - pa[n].grad).norm() / pa[n].grad.norm()))
        out.append(errs)
    print(f"cin={cin} B={B} seed={seed}")
    for n in out[0]:
        f = lambda v: v if isinstance(v, float) else v[0]
        print(f"   {n:14s} split {f(out[0][n]):.1e}   fused-fp32 {f(out[1][n]):.1e}   stock-fp32 {f(out[2][n]):.1e}")
