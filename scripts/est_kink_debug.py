import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
EE = d.compat.ErrorEstimators
for cin, B, seed in ((7, 5, 5), (7, 5, 6), (7, 5, 7)):
    stock = EE.ErrorEstimator(cin); d.synth.fill_params_deterministic(stock, seed=seed); stock = stock.double()
    g = torch.Generator().manual_seed(1)
    x = torch.rand(B, cin, 100, generator=g).double()
    h = x
    mods = list(stock.fw)
    i = 0
    print("seed", seed)
    while i + 2 < len(mods):
        y = mods[i](h); z = mods[i + 1](y)
        var = y.var(2, unbiased=False)
        srt = z.abs().flatten().sort()[0]
        print(f"  layer {i//3}: min|z| {srt[0]:.2e} {srt[1]:.2e} {srt[2]:.2e}; min var {var.min():.2e}; #|z|<1e-6: {(z.abs()<1e-6).sum().item()}")
        h = mods[i + 2](z.clone()); i += 3
