"""Forward / forward+backward time of ONE ErrorEstimator call: stock PyTorch, the native-fp32 fused evaluation (library GEMMs +
inorm kernel) and the split-bf16 matrix-core chain (csrc/est_gemm.hip), plus the accuracy of each against float64.
   python scripts/estimator_time.py [B [N]]      (N = 100: the fused epilogue; any other N: plain product + norm kernel)"""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
EE = d.compat.ErrorEstimators
if os.environ.get("TN_BLOCKS"):  # A/B of the weight-gradient GEMM's split-K granularity
    d.estimator.TN_BLOCKS = int(os.environ["TN_BLOCKS"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
FL = 2.0 * B * N * (7 * 64 + 64 * 128 + 128 * 1024 + 1024 * 512 + 512 * 256 + 256)
variants = [("stock", EE.ErrorEstimator, None), ("fused fp32", EE.FusedErrorEstimator, False), ("split-bf16 MFMA", EE.FusedErrorEstimator, True)]
x0 = torch.rand(B, 7, N, device="cuda")
G = torch.randn(B, 1, N, device="cuda")
# float64 truth on a slice
ref = EE.ErrorEstimator(7); d.synth.fill_params_deterministic(ref, 1); ref = ref.double()
xs = x0[:8].cpu().double().requires_grad_(True)
ys = ref(xs); (ys * G[:8].cpu().double()).sum().backward()
for name, cls, split in variants:
    m = cls(7).cuda(); d.synth.fill_params_deterministic(m, 1)
    if split is not None: m.split_bf16 = split
    x = x0.clone().requires_grad_(True)
    def step():
        m.zero_grad(set_to_none=True); x.grad = None; y = m(x); (y * G).sum().backward()
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    with torch.no_grad():
        for _ in range(2): m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): m(x)
        torch.cuda.synchronize(); df = (time.perf_counter() - t0) / 5
    # accuracy on the 8-pair slice (instances are per pair: the slice is self-contained)
    x8 = x0[:8].clone().requires_grad_(True)
    m.zero_grad(set_to_none=True)
    y8 = m(x8); (y8 * G[:8]).sum().backward()
    el = float((y8.detach().cpu().double() - ys.detach()).abs().max())
    ex = float((x8.grad.cpu().double() - xs.grad).abs().max() / xs.grad.abs().max())
    pw = dict(m.named_parameters())["fw.9.weight"].grad.cpu().double(); rw = dict(ref.named_parameters())["fw.9.weight"].grad
    ew = float((pw - rw).norm() / rw.norm())
    print(f"{name:18s} B={B} N={N}: fwd {df*1e3:7.2f} ms ({FL/df/1e12:6.1f} TF/s-equiv), fwd+bwd {dt*1e3:7.2f} ms ({3*FL/dt/1e12:6.1f}); "
          f"vs fp64: logits {el:.1e}, d/dx {ex:.1e}, dW(1024x512) rel-norm {ew:.1e}; peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
    del m, x
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
