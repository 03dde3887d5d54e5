"""Time the full compat DeepFNet (stock-PyTorch ErrorEstimator + HIP solver) forward + F-loss + qt loss + backward."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
B, N, depth = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 100, 5
dev = "cuda:0"
sc = d.pipeline.scene_to_device(d.synth.make_scene(B, N, seed=1, outlier_ratio=0.2), dev)
net = d.compat.DeepFNet.DeepFNet(depth=depth, image_size=[376, 1241, 3], if_quality=False).to(dev)
d.synth.fill_params_deterministic(net, 1)
tg = d.compat.train_good_utils
lp = {"depth": depth, "clamp_at": 0.02, "if_tri_depth": False, "if_sample_loss": False, "topK": 8, "matches_good_unique_nums": None}
batch = {"matches_xy_ori": sc["matches_xy_ori"], "matches_good_unique_nums": None, "t_scene_scale": None}
def step():
    net.zero_grad(set_to_none=True)
    outs = net(batch)
    losses, E, F, _, _, _, E_layers = tg.get_all_loss_DeepF(outs, sc["pts1_virt_ori"], sc["pts2_virt_ori"], sc["Ks"], lp, get_residual_summaries=False)
    rt = tg.get_Rt_loss(E_layers, None, None, None, sc["delta_Rtijs_4_4"], sc["qs_cam"], sc["ts_cam"], device=dev)
    loss = torch.clamp(torch.stack(rt["q_l2_error_layers_list"]), 0, 0.1).mean() + 0.1 * torch.clamp(torch.stack(rt["t_l2_error_layers_list"]), 0, 0.5).mean() + losses["loss_F"]
    loss.backward()
    return loss
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 5
for _ in range(n): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"full DeepFNet step B={B} N={N} depth={depth}: {dt*1e3:.1f} ms/step = {B/dt:.0f} pairs/s; peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
