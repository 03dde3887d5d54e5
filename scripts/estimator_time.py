import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = importlib.import_module("pytorch-deepfepe_amd")
EE = d.compat.ErrorEstimators
B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 100
for cls in (EE.ErrorEstimator, EE.FusedErrorEstimator):
    m = cls(7).cuda(); d.synth.fill_params_deterministic(m, 1)
    x = torch.rand(B, 7, N, device="cuda", requires_grad=True)
    def step():
        m.zero_grad(set_to_none=True); y = m(x); y.sum().backward()
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    with torch.no_grad():
        for _ in range(2): m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): m(x)
        torch.cuda.synchronize(); df = (time.perf_counter() - t0) / 5
    print(f"{cls.__name__:22s} B={B}: forward {df*1e3:7.2f} ms, forward+backward {dt*1e3:7.2f} ms  (peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB)")
    torch.cuda.reset_peak_memory_stats()
