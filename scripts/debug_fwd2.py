import importlib, sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
o = importlib.import_module("oracle.deepf_oracle")
np.set_printoptions(precision=4, suppress=False, linewidth=220)
B, N = 2, 100
sc = d.synth.make_scene(B, N, seed=1, outlier_ratio=0.2)
m = sc["matches_xy_ori"]; w = torch.softmax(sc["logits_layers"][0], 1)
p1, p2, _ = o.normalize_hw(m, [376, 1241, 3])
L = d._lib.lib()
F = torch.empty(B, 9, device="cuda"); res = torch.empty(B, N, device="cuda"); save = torch.empty(B, 128, device="cuda")
p1c, p2c, wc = p1.cuda().contiguous(), p2.cuda().contiguous(), w.cuda().contiguous()
rc = L.dfepe_w8pt_fwd(p1c.data_ptr(), p2c.data_ptr(), wc.data_ptr(), B, N, 0, 0.0, 0.0, -1.0, F.data_ptr(), res.data_ptr(), None, save.data_ptr(), None)
torch.cuda.synchronize(); print("rc", rc)
A = save[0, 15:96].cpu().double().numpy().reshape(9, 9)
out, r, aux = o.fit_forward(p1.double(), p2.double(), w.double().unsqueeze(1))
X = aux["X"][0].numpy(); M = X.T @ X
print("A ours\n", A); print("M oracle\n", M); print("ratio\n", A / M)
