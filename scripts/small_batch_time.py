"""The full compat.DeepFNet training step at the reference's own batch sizes (its configs train with 4-32 pairs per batch): eager and
through compat.CapturedStep, and -- under `rocprofv3 --kernel-trace --stats` -- how many launches a step is.
   python scripts/small_batch_time.py [B [N]]"""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
depth, dev = 5, "cuda:0"
net = d.compat.DeepFNet.DeepFNet(depth=depth, image_size=[376, 1241, 3], if_quality=False).to(dev)
d.synth.fill_params_deterministic(net, 1)
tg = d.compat.train_good_utils
keys = ("matches_xy_ori", "pts1_virt_ori", "pts2_virt_ori", "Ks", "delta_Rtijs_4_4", "qs_cam", "ts_cam")
batches = []
for s in range(3):
    sc = d.synth.make_scene(B, N, seed=s + 1, outlier_ratio=0.2, noise_px=0.5)
    batches.append({k: sc[k].to(dev) for k in keys})


def forward_and_loss(b):
    lp = {"depth": depth, "clamp_at": 0.02, "if_tri_depth": False, "if_sample_loss": False, "topK": 8, "matches_good_unique_nums": None}
    outs = net({"matches_xy_ori": b["matches_xy_ori"], "matches_good_unique_nums": None, "t_scene_scale": None})
    losses, _, _, _, _, _, E_layers = tg.get_all_loss_DeepF(outs, b["pts1_virt_ori"], b["pts2_virt_ori"], b["Ks"], lp, get_residual_summaries=False)
    geo = tg.get_Rt_loss(E_layers, b["Ks"], None, None, b["delta_Rtijs_4_4"], b["qs_cam"], b["ts_cam"], device=dev)
    lq = torch.clamp(torch.stack(geo["q_l2_error_layers_list"]), 0.0, 0.1).mean()
    lt = torch.clamp(torch.stack(geo["t_l2_error_layers_list"]), 0.0, 0.5).mean()
    return losses["loss_F"] + lq + 0.1 * lt, {"geo": geo}


opt = torch.optim.Adam(net.parameters(), lr=1e-4)


def eager(b):
    opt.zero_grad(set_to_none=True)
    loss, _ = forward_and_loss(b)
    loss.backward()
    opt.step()


def timed(fn, n=20):
    for i in range(4): fn(batches[i % 3])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(batches[i % 3])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


te = timed(eager)
step = d.compat.CapturedStep(forward_and_loss, net, warmup=1)


def captured(b):
    opt.zero_grad(set_to_none=True)
    step(b)
    opt.step()


tc = timed(captured)
print(f"full DeepFNet step (+ Adam) B={B} N={N}: eager {te:.2f} ms, CapturedStep {tc:.2f} ms (captures {step.n_captures}, replays {step.n_replays})")
