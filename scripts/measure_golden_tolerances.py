import importlib, sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_compat_gpu as tc
d = importlib.import_module("pytorch-deepfepe_amd")
g = np.load("/root/repo/tests/golden/pipeline.npz")
depth=3; DEV="cuda:0"
net = d.compat.DeepFNet.DeepFNet(depth=depth, image_size=tc.IMAGE_SIZE, if_quality=False, if_cpu_svd=True)
d.synth.fill_params_deterministic(net, seed=5); net=net.to(DEV)
layer={"i":0}; orig=net._fit
def fit(matches, logits, data_batch, want_epi):
    o=orig(matches, logits, data_batch, want_epi)
    ref=torch.from_numpy(g["net_out_layers"][layer["i"]]).to(DEV)
    s=torch.sign((o[0].detach()*ref).flatten(1).sum(1)); layer["i"]+=1
    return (o[0]*s[:,None,None], o[1]*s[:,None])+tuple(o[2:])
net._fit=fit
batch={"matches_xy_ori": torch.from_numpy(g["net_matches_xy_ori"]).to(DEV), "matches_good_unique_nums": None, "t_scene_scale": None}
outs=net(batch)
def md(a,b): 
    a=np.asarray(a,np.float64); b=np.asarray(b,np.float64); return np.abs(a-b).max(), np.abs(a-b).max()/(np.abs(b).max()+1e-300)
for l in range(depth):
    a,r,_=tc.unit_align(outs["out_layers"][l].detach().cpu().numpy(), g["net_out_layers"][l])
    print(l,"F unit err", np.linalg.norm(a-r,axis=1).max(), "logits", md(outs["logits_layers"][l].detach().cpu().numpy(), g["net_logits_layers"][l]),
          "weights", md(outs["weights_layers"][l].detach().cpu().numpy(), g["net_weights_layers"][l]), "residual", md(outs["residual_layers"][l].detach().cpu().numpy(), g["net_residual_layers"][l]))
for l in range(depth-1): print("epi", md(outs["epi_res_layers"][l].detach().cpu().numpy(), g["net_epi_res_layers"][l]))
loss_params={"depth":depth,"clamp_at":0.02,"if_tri_depth":False,"if_sample_loss":False,"topK":8,"matches_good_unique_nums":None}
T=lambda x: torch.from_numpy(x)
losses,*_=d.compat.train_good_utils.get_all_loss_DeepF(outs, T(g["net_pts1_virt_ori"]).to(DEV), T(g["net_pts2_virt_ori"]).to(DEV), T(g["net_Ks"]).to(DEV), loss_params, get_residual_summaries=False)
print("loss_F", losses["loss_F"].item(), g["net_loss_F"], abs(losses["loss_F"].item()-g["net_loss_F"])/g["net_loss_F"])
losses["loss_F"].backward()
gn={n:(0.0 if p.grad is None else float(p.grad.double().norm())) for n,p in net.named_parameters()}
ours=np.array([gn[n] for n in sorted(gn)]); ref=g["net_grad_norms"]
print("grad norms rel", np.abs(ours-ref).max()/ref.max(), "per-entry rel max", (np.abs(ours-ref)/np.maximum(ref,1e-4*ref.max())).max())
ga=net.input_weights.fw[0].weight.grad.cpu().numpy().ravel(); gr=g["net_grad_first_conv"].ravel()
print("first conv cos", (ga*gr).sum()/(np.linalg.norm(ga)*np.linalg.norm(gr)), "relerr", np.abs(ga-gr).max()/np.abs(gr).max())
