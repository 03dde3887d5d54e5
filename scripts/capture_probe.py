"""Which part of the full DeepFNet training step survives hipGraph capture on this stack?  Every variant runs in a subprocess of its
own (a failing capture can take the process down):  python scripts/capture_probe.py            (the parent)
                                                     python scripts/capture_probe.py <variant>  (one variant)"""
import importlib
import os
import subprocess
import sys

VARIANTS = ["fwd_only", "fwd_loss", "grad_fn", "backward", "backward_default_stream", "fixed_logits_backward", "stock_estimator_backward",
            "estimator_only_backward", "estimator_only_grad", "helper"]

if len(sys.argv) == 1:
    for v in VARIANTS:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), v], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        tail = [l for l in r.stdout.strip().splitlines() if l.strip()][-2:]
        print(f"{v:28s} rc {r.returncode:4d}  {' | '.join(tail)[:230]}", flush=True)
    sys.exit(0)

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
v = sys.argv[1]
DEV, depth, B, N = "cuda:0", 3, 48, 100
IMAGE_SIZE = [376, 1241, 3]
tgu = d.compat.train_good_utils
tgu.LAZY_HOST_METRICS = True
sc = d.synth.make_scene(B, N, seed=5, outlier_ratio=0.2, noise_px=0.5, depth_layers=depth)
b = {k: t.to(DEV) for k, t in sc.items()}
if v == "fixed_logits_backward":
    rows = [b["logits_layers"][l].detach().clone().unsqueeze(1).requires_grad_(True) for l in range(depth)]
    net = d.pipeline.make_api_net(depth, IMAGE_SIZE, rows)
    params = rows
else:
    net = d.compat.DeepFNet.DeepFNet(depth=depth, image_size=IMAGE_SIZE, if_quality=False, fused_estimator=(v != "stock_estimator_backward")).to(DEV)
    d.synth.fill_params_deterministic(net, 3)
    params = list(net.parameters())
lp = {"depth": depth, "clamp_at": 0.02, "if_tri_depth": False, "if_sample_loss": False, "topK": 8, "matches_good_unique_nums": None}


def fwd():
    for est in (net.input_weights, net.update_weights):
        if hasattr(est, "k"):
            est.k = 0
    return net({"matches_xy_ori": b["matches_xy_ori"], "matches_good_unique_nums": None, "t_scene_scale": None})


def loss_of(outs):
    losses, _, _, _, _, _, E_layers = tgu.get_all_loss_DeepF(outs, b["pts1_virt_ori"], b["pts2_virt_ori"], b["Ks"], lp, get_residual_summaries=False)
    geo = tgu.get_Rt_loss(E_layers, b["Ks"], None, None, b["delta_Rtijs_4_4"], b["qs_cam"], b["ts_cam"], device=DEV)
    lq = torch.clamp(torch.stack(geo["q_l2_error_layers_list"]), 0.0, 0.1).mean()
    lt = torch.clamp(torch.stack(geo["t_l2_error_layers_list"]), 0.0, 0.5).mean()
    return losses["loss_F"] + lq + 0.1 * lt


st = {}
x_est = torch.randn(B, 4, N, device=DEV)


def body():
    if v == "fwd_only":
        st["o"] = fwd()["F_est"]
    elif v == "fwd_loss":
        st["o"] = loss_of(fwd())
    elif v == "grad_fn":
        st["o"] = torch.autograd.grad(loss_of(fwd()), params, allow_unused=True)
    elif v in ("estimator_only_backward", "estimator_only_grad"):
        y = net.input_weights(x_est).square().mean()
        if v.endswith("grad"):
            st["o"] = torch.autograd.grad(y, list(net.input_weights.parameters()))
        else:
            y.backward()
    else:
        loss_of(fwd()).backward()


if v == "helper":
    step = d.compat.CapturedStep(lambda bb: (loss_of(fwd()), None), params, warmup=2)
    for _ in range(5):
        loss, _ = step(b)
    torch.cuda.synchronize()
    print("OK helper, loss", float(loss), "replays", step.n_replays)
    sys.exit(0)

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        for p in params:
            p.grad = None
        body()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
for p in params:
    p.grad = None
g = torch.cuda.CUDAGraph()
kw = {} if v == "backward_default_stream" else {"stream": side}
print("capturing", v, flush=True)
with torch.cuda.graph(g, **kw):
    body()
print("captured", flush=True)
g.replay()
torch.cuda.synchronize()
print("OK", v, flush=True)
