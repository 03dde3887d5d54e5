"""One estimator call against the stock module in float64, through the one-call-per-pass path and through the per-launch host code: logits and
every parameter gradient (relative 2-norm).   python scripts/est_path_accuracy.py [cin [B [N [seed]]]]"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
cin = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 5
DEV = "cuda:0"
EE = d.compat.ErrorEstimators
stock = EE.ErrorEstimator(cin)
d.synth.fill_params_deterministic(stock, seed=seed)
fused = EE.FusedErrorEstimator(cin).to(DEV)
fused.load_state_dict(stock.state_dict())
stock32 = EE.ErrorEstimator(cin).to(DEV)
stock32.load_state_dict(stock.state_dict())
stock = stock.double()
g = torch.Generator().manual_seed(N + B)
x = torch.rand(B, cin, N, generator=g)
G = torch.randn(B, 1, N, generator=g)
margin = [float("inf")]
hooks = [m.register_forward_hook(lambda _m, _i, o: margin.__setitem__(0, min(margin[0], float(o.detach().abs().min())))) for m in stock.fw if isinstance(m, torch.nn.InstanceNorm1d)]
ya = stock(x.double())
(ya * G.double()).sum().backward()
ref = {n: p.grad for n, p in stock.named_parameters()}
print(f"cin={cin} B={B} N={N}: float64 kink margin {margin[0]:.1e}")
rel = lambda a, b: float((a.cpu().double() - b).norm() / b.norm().clamp_min(1e-300))
for label, model, use in (("pass", fused, True), ("per-launch", fused, False), ("stock fp32", stock32, None)):
    if use is not None:
        d.estimator.USE_PASS = use
    model.zero_grad(set_to_none=True)
    y = model(x.to(DEV))
    (y * G.to(DEV)).sum().backward()
    errs = {n: rel(p.grad, ref[n]) for n, p in model.named_parameters() if float(ref[n].abs().max()) > 1e-9}
    worst = max(errs, key=errs.get)
    print(f"  {label:10s}: logits {float((y.detach().cpu().double() - ya.detach()).abs().max()):.1e}; worst gradient {worst} {errs[worst]:.1e}; " +
          " ".join(f"{n.replace('fw.', '')}:{e:.0e}" for n, e in errs.items()))
