"""Where the HOST time of the eager full-model training step goes at the reference's batch (8 pairs; estimators included): cProfile over the steps of
scripts/small_batch_time.py's eager loop (forward, both losses, backward, Adam).   python scripts/eager_profile_full_model.py [B [N [steps]]]"""
import cProfile, importlib, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
depth, dev = 5, "cuda:0"
net = d.compat.DeepFNet.DeepFNet(depth=depth, image_size=[376, 1241, 3], if_quality=False).to(dev)
d.synth.fill_params_deterministic(net, 1)
tg = d.compat.train_good_utils
keys = ("matches_xy_ori", "pts1_virt_ori", "pts2_virt_ori", "Ks", "delta_Rtijs_4_4", "qs_cam", "ts_cam")
sc = d.synth.make_scene(B, N, seed=1, outlier_ratio=0.2, noise_px=0.5)
b = {k: sc[k].to(dev) for k in keys}
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
lp = {"depth": depth, "clamp_at": 0.02, "if_tri_depth": False, "if_sample_loss": False, "topK": 8, "matches_good_unique_nums": None}


def step():
    opt.zero_grad(set_to_none=True)
    outs = net({"matches_xy_ori": b["matches_xy_ori"], "matches_good_unique_nums": None, "t_scene_scale": None})
    losses, _, _, _, _, _, E_layers = tg.get_all_loss_DeepF(outs, b["pts1_virt_ori"], b["pts2_virt_ori"], b["Ks"], lp, get_residual_summaries=False)
    geo = tg.get_Rt_loss(E_layers, b["Ks"], None, None, b["delta_Rtijs_4_4"], b["qs_cam"], b["ts_cam"], device=dev)
    lq = torch.clamp(torch.stack(geo["q_l2_error_layers_list"]), 0.0, 0.1).mean()
    lt = torch.clamp(torch.stack(geo["t_l2_error_layers_list"]), 0.0, 0.5).mean()
    (losses["loss_F"] + lq + 0.1 * lt).backward()
    opt.step()


for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print(f"eager full model B={B} N={N}: {(time.perf_counter() - t0) * 1e3 / steps:.3f} ms per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(40)
