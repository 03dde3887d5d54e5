"""Is torch's reduction over a long outer dimension replay-safe on this stack?"""
import torch
DEV = "cuda:0"
for shape, dim in (((512, 64, 32), 0), ((512, 1024), 0), ((48, 64), 0), ((4096, 1024), 0), ((1, 512, 256), 1), ((96, 1024, 128), 0)):
    xs = [torch.randn(*shape, device=DEV) for _ in range(3)]
    st = torch.empty(*shape, device=DEV)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        st.copy_(xs[0]); st.sum(dim)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        tmp = torch.empty(*shape, device=DEV)   # like the partial-sum buffers of the estimator's backward: allocated in the capture
        tmp.copy_(st)
        y = tmp.sum(dim)
        del tmp
        z = y * 1.0
    res = []
    for it in range(5):
        k = it % 3
        st.copy_(xs[k]); g.replay(); torch.cuda.synchronize()
        ref = xs[k].sum(dim)
        res.append(f"{float((y - ref).abs().max() / ref.abs().max()):.1e}/{float((z - ref).abs().max() / ref.abs().max()):.1e}")
    print(shape, dim, res, flush=True)
