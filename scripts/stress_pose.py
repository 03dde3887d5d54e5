"""One-off randomized sweep of the pose loss (forward + adjoint) against the fp64 oracle for hard inputs: exact essential
matrices (s1 = s2), noisy ones, arbitrary 3x3 matrices, tiny and huge scales."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
oracle = importlib.import_module("oracle.deepf_oracle")
B = 256
sc = d.synth.make_scene(B, 50, seed=5, outlier_ratio=0.0, noise_px=0.0)
g = torch.Generator().manual_seed(1)
E0 = sc["E_gt"].double()
cases = {
    "exact essential": E0,
    "essential + 1e-3 noise": E0 + 1e-3 * torch.randn(B, 3, 3, generator=g).double() * E0.flatten(1).norm(dim=1)[:, None, None],
    "essential + 1e-1 noise": E0 + 1e-1 * torch.randn(B, 3, 3, generator=g).double() * E0.flatten(1).norm(dim=1)[:, None, None],
    "random 3x3": torch.randn(B, 3, 3, generator=g).double(),
    "scaled 1e-6": E0 * 1e-6,
    "scaled 1e+6": E0 * 1e6,
    "transposed (wrong convention)": E0.transpose(1, 2).contiguous(),
}
for name, E in cases.items():
    Ef = E.float()
    Ed = Ef.double().clone().requires_grad_(True)           # identical fp32-representable inputs on both sides
    ref = oracle.rt_loss([Ed], sc["delta_Rtijs_4_4"].double(), sc["qs_cam"].double(), sc["ts_cam"].double())
    lref = oracle.qt_training_loss(ref["q_l2"], ref["t_l2"], 0.1, 0.5, 1.0, 0.1)
    lref.backward()
    Eg = Ef.unsqueeze(0).cuda().requires_grad_(True)
    R_gt = sc["delta_Rtijs_4_4"][:, :3, :3].transpose(1, 2).contiguous().cuda()
    q, t, Rd, td, sel = d.ops.pose_errors(Eg, sc["qs_cam"].cuda(), sc["ts_cam"].cuda(), R_gt)
    loss = torch.clamp(q, 0, 0.1).mean() * 1.0 + torch.clamp(t, 0, 0.5).mean() * 0.1
    loss.backward()
    dq = (q[0].detach().cpu().double() - ref["q_l2"][0].detach()).abs().max().item()
    dt = (t[0].detach().cpu().double() - ref["t_l2"][0].detach()).abs().max().item()
    dR = np.abs(Rd[0].cpu().numpy() - ref["R_deg"][0]).max()
    dT = np.abs(td[0].cpu().numpy() - ref["t_deg"][0]).max()
    gg, gr = Eg.grad[0].cpu().double(), Ed.grad
    rel = ((gg - gr).flatten(1).norm(dim=1) / gr.flatten(1).norm(dim=1).clamp_min(1e-300))
    live = gr.flatten(1).norm(dim=1) > 0
    print(f"{name:32s}: |dq| {dq:.1e} |dt| {dt:.1e} |dR deg| {dR:.1e} |dt deg| {dT:.1e}  finite {bool(torch.isfinite(Eg.grad).all())}  grad rel err median {rel[live].median().item():.1e} max {rel[live].max().item():.1e}  (live {int(live.sum())})")
