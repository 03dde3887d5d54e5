"""Host logic of the reference-API path that needs no GPU: the zero-copy per-layer stacks (ops.stack_rows / unstack_rows: the
reference's API speaks in python lists of per-layer tensors, the kernels in one [L, ...] buffer), the lazily copied host metrics
of get_Rt_loss, and the code-object check that no kernel of the library uses scratch memory (two instantiations
of the head-carrying backward spilled unnoticed)."""
import os
import sys

import numpy as np
import pytest
import torch


def test_stack_rows_aliases_a_buffer_and_copies_otherwise(dfepe):
    ops = dfepe.ops
    buf = torch.arange(24.0).reshape(3, 2, 4).clone()
    rows = [ops.row_of(buf, l) for l in range(3)]
    assert all(r._base is None and r.data_ptr() == buf[l].data_ptr() and torch.equal(r, buf[l]) for l, r in enumerate(rows))
    a = ops.alias_rows(rows)
    assert a is not None and a.data_ptr() == buf.data_ptr() and a.shape == buf.shape and torch.equal(a, buf)
    assert ops.alias_rows(rows[::-1]) is None  # order
    big = torch.arange(48.0).reshape(6, 2, 4).clone()
    every_other = ops.alias_rows([ops.row_of(big, l) for l in (0, 2, 4)])  # a uniform distance: a strided stack (the same channel
    assert every_other is not None and not every_other.is_contiguous() and torch.equal(every_other, big[::2])  # of consecutive buffers)
    assert ops.alias_rows([ops.row_of(big, l) for l in (0, 2, 5)]) is None  # uneven distances
    assert ops.alias_rows([rows[0], torch.zeros(2, 4)]) is None and ops.alias_rows([rows[0], rows[1].double()]) is None
    assert ops.alias_rows([buf[:, :, :2][0], buf[:, :, :2][1]]) is None  # non-contiguous rows
    assert ops.alias_rows([]) is None and ops.alias_rows([None]) is None
    v = [buf[l].unsqueeze(0) for l in range(3)]  # views of the rows (DeepFNet's epi_res.unsqueeze(1)) still alias
    assert ops.alias_rows(v).data_ptr() == buf.data_ptr()
    s = ops.stack_rows(rows)
    assert s.data_ptr() == buf.data_ptr()
    s2 = ops.stack_rows([torch.ones(2), torch.zeros(2)])
    assert torch.equal(s2, torch.tensor([[1.0, 1.0], [0.0, 0.0]]))


def test_rows_carry_gradients_like_stack_and_unbind(dfepe):
    ops = dfepe.ops

    class Fill(torch.autograd.Function):  # a producer that writes its result into row l of the caller's stack (like the fit kernels)
        @staticmethod
        def forward(ctx, w, dst):
            out = ops.row_of(*dst)
            out.copy_(w * 2)
            return out

        @staticmethod
        def backward(ctx, g):
            return g * 2, None

    g = torch.Generator().manual_seed(0)
    ws = [torch.randn(4, generator=g, requires_grad=True) for _ in range(3)]
    buf = torch.empty(3, 4)
    rows = [Fill.apply(ws[l], (buf, l)) for l in range(3)]
    S = ops.stack_rows(rows)
    assert S.data_ptr() == buf.data_ptr() and S.requires_grad
    E = S * 3
    rows2 = ops.unstack_rows(E)
    assert rows2[1].data_ptr() == E[1].data_ptr() and all(r.grad_fn is not None for r in rows2)
    assert ops.stack_rows(rows2).data_ptr() == E.data_ptr()  # what get_Rt_loss does with get_all_loss_DeepF's E_ests_layers
    loss = torch.clamp(torch.stack(rows2), 0, 0.5).mean() + 0.1 * ops.stack_rows(rows2).sum() + rows2[2].sum()
    loss.backward()
    ref = [w.detach().clone().requires_grad_() for w in ws]
    st = torch.stack([w * 6 for w in ref])
    (torch.clamp(st, 0, 0.5).mean() + 0.1 * st.sum() + st[2].sum()).backward()
    for a, b in zip(ws, ref):
        assert torch.allclose(a.grad, b.grad)
    # a single row used: the others get exact zeros; nothing used: no gradient at all
    ws3 = [w.detach().clone().requires_grad_() for w in ws]
    r3 = ops.unstack_rows(ops.stack_rows([w * 2 for w in ws3]) * 3)
    r3[1].sum().backward()
    assert torch.equal(ws3[0].grad, torch.zeros(4)) and torch.equal(ws3[1].grad, torch.full((4,), 6.0))
    # the gradient of torch.stack(rows) arrives as the rows of one buffer and is taken without a copy
    seen = {}
    x = torch.randn(3, 5, requires_grad=True)
    y = x * 1.0
    y.register_hook(lambda gr: seen.__setitem__("ptr", gr.data_ptr()))
    up = torch.randn(3, 5)
    stacked = torch.stack(ops.unstack_rows(y))
    stacked.register_hook(lambda gr: seen.__setitem__("up", gr.data_ptr()))
    (stacked * up).sum().backward()
    assert seen["ptr"] == seen["up"] and torch.allclose(x.grad, up)


def test_lazy_host_metrics_behave_like_the_references_numpy_values(dfepe):
    tgu = dfepe.compat.train_good_utils
    calls = {"n": 0}

    def fetch():
        calls["n"] += 1
        return np.array([0.02, 3.0, 0.7])

    x = tgu._Lazy(fetch)
    assert calls["n"] == 0  # nothing is read until somebody looks
    # the uses Train_model_pipeline.py:823-885 makes of the per-layer error arrays and their means
    assert np.amax(x) == 3.0 and np.max(x) == 3.0 and x.flatten().shape == (3,)
    np.testing.assert_array_equal(np.clip(x, 0.0, 1.0), [0.02, 1.0, 0.7])
    np.testing.assert_array_equal(np.hstack((x[x < 0.05], np.zeros(1))), [0.02, 0.0])
    assert np.stack([x, x]).shape == (2, 3) and len(x) == 3 and list(x) == [0.02, 3.0, 0.7] and x.mean() == pytest.approx(1.24)
    np.testing.assert_array_equal(x * 2, [0.04, 6.0, 1.4])
    np.testing.assert_array_equal(2 * x, [0.04, 6.0, 1.4])
    m = tgu._LazyScalar(lambda: 1.5)
    assert float(m) == 1.5 and np.isscalar(m) and "%.2f" % m == "1.50" and f"{m:.1f}" == "1.5" and m + 1 == 2.5 and m < 2
    np.testing.assert_array_equal(np.array([m]), [1.5])
    assert calls["n"] > 0
    assert not np.isscalar(x)  # only the scalar lazies count as numbers.Real
    # pickle / torch.save / deepcopy store the realised value, not the closure
    import copy
    import io
    import pickle

    back = pickle.loads(pickle.dumps({"a": x, "m": m}))
    assert isinstance(back["a"], np.ndarray) and isinstance(back["m"], float) and back["m"] == 1.5
    np.testing.assert_array_equal(back["a"], [0.02, 3.0, 0.7])
    bio = io.BytesIO()
    torch.save({"a": x}, bio)
    assert isinstance(copy.deepcopy(m), float)
    # the dict of get_Rt_loss: reference types by default (LAZY_HOST_METRICS is off), realise() converts a lazy dict in place
    assert tgu.LAZY_HOST_METRICS is False
    geo = tgu._GeoErrors({"R_angle_error_mean": m, "t_angle_error_mean": m, "R_angle_error_list": x, "t_angle_error_list": x,
                          "R_angle_error_layers_list": [x, x], "t_angle_error_layers_list": [x], "q_l2_error_mean": torch.zeros(())})
    geo.realise()
    assert type(geo["R_angle_error_mean"]) is float and type(geo["R_angle_error_list"]) is np.ndarray
    assert all(type(a) is np.ndarray for a in geo["R_angle_error_layers_list"]) and torch.is_tensor(geo["q_l2_error_mean"])
    pickle.dumps(dict(geo))


def test_in_place_writes_to_aliased_rows_raise_instead_of_corrupting_saved_stacks(dfepe):
    """the per-layer rows and the [L, ...] stacks share memory.  A caller's in-place write to a row must not
    silently change what a kernel saved of the stack: rows handed out by unstack_rows are real autograd views (the write itself
    raises, like on the outputs of torch.unbind), and a stack re-assembled from rows checks the rows' version counters in its
    backward."""
    ops = dfepe.ops
    x = torch.randn(3, 4, requires_grad=True)
    rows = ops.unstack_rows(x * 2.0)
    assert all(r._is_view() for r in rows)
    with pytest.raises(RuntimeError, match="modified inplace|in-place"):
        rows[1].mul_(2.0)
    # without gradient tracking the rows are plain views and writable (a no-grad evaluation loop may normalise in place)
    with torch.no_grad():
        free = ops.unstack_rows(torch.ones(2, 3))
        free[0].mul_(3.0)
    # rows written by producers into one buffer, re-assembled without a copy, saved by a consumer, then overwritten by the caller
    buf = torch.zeros(3, 4)
    prods = []
    for l in range(3):
        r = ops.row_of(buf, l)
        r.copy_(torch.full((4,), float(l)))
        prods.append(r.requires_grad_())
    S = ops.stack_rows(prods)
    assert S.data_ptr() == buf.data_ptr()
    loss = (S * S).sum()  # MulBackward saves S, the alias
    with torch.no_grad():
        prods[2].add_(1.0)  # the caller "post-processes" a layer's output in place
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        loss.backward()


def test_no_kernel_of_the_library_uses_scratch_memory():
    """Code-object metadata of every object the library is linked from (scripts/kernel_resources.py): private segment and
    spill counts must be zero -- a spilling instantiation is a performance bug that no parity test sees."""
    repo = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    sys.path.insert(0, os.path.join(repo, "scripts"))
    try:
        import kernel_resources
    finally:
        sys.path.pop(0)
    objdir = os.path.join(repo, "pytorch-deepfepe_amd", "csrc", "build")
    if not os.path.isdir(objdir) or not any(f.endswith(".o") for f in os.listdir(objdir)):
        pytest.skip("no objects under csrc/build (the library was built elsewhere)")
    ks = kernel_resources.kernels(objdir)
    assert len(ks) > 100
    bad = [(k["name"][:90], k["scratch"], k["spill"]) for k in ks if k["scratch"] or k["spill"]]
    assert not bad, bad


def test_estimator_launch_geometry_helpers(dfepe):
    """Host-side choices of the estimator: the weight-gradient GEMM's split-K slices stay a multiple of eight whenever a launch
    has more than one slice per XCD to place (est_gemm_tn's XCD-aware order needs it and falls back to launch order otherwise) and
    fill TN_BLOCKS workgroups; the N-generic normalisation splits a pair's rows only when (pair, channel-block) workgroups do not
    fill the chip, never below one unrolled trip (128 rows) per split, never beyond the kernel's 64."""
    est = dfepe.estimator
    for cout, cin in [(64, 32), (128, 64), (1024, 128), (512, 1024), (256, 512), (1024, 1024), (256, 32)]:
        tiles = ((cout + 127) // 128) * ((cin + 127) // 128)
        s = est._slices_for(cout, cin)
        assert 1 <= s <= 512 and s * tiles <= max(est.TN_BLOCKS, tiles)
        assert s % 8 == 0 or s < 8, (cout, cin, s)
    assert est._row_splits(4096, 1024, 100) == 1      # the benchmark's shape never splits
    assert est._row_splits(128, 1024, 1000) == 1      # 2048 workgroups already
    assert est._row_splits(12, 64, 2000) == 15        # 12 workgroups -> 15 splits of >= 128 rows
    assert est._row_splits(12, 1024, 2000) == 6       # 192 workgroups -> ~1024
    assert est._row_splits(1, 64, 100000) == 64       # the kernel's limit
    assert est._row_splits(2, 64, 200) == 1           # too few rows to split
    for pairs, c, n in [(1, 32, 256), (3, 64, 300), (8, 256, 1000), (12, 512, 2000)]:
        s = est._row_splits(pairs, c, n)
        assert 1 <= s <= 64 and (s == 1 or n // s >= 128)


def test_estimator_small_batch_slices_and_the_one_call_per_pass_decision(dfepe):
    """Host-side choices added in round 5: over few columns a split-K slice keeps >= 256 of them (the reference's batch sizes: the
    partial sums would otherwise outweigh the product); and which estimators take ONE library call per pass -- a one-channel head,
    <= 8 hidden layers of widths on the 32 grid, fp32 parameters -- everything else the per-launch host code."""
    est = dfepe.estimator
    assert est._slices_for(512, 1024, 800) == 3 and est._slices_for(512, 1024, 100) == 1
    assert est._slices_for(512, 1024, 409600) == est._slices_for(512, 1024)
    for cols in (200, 800, 3200, 6400, 409600):
        for cout, cin in [(64, 32), (1024, 128), (512, 1024)]:
            s = est._slices_for(cout, cin, cols)
            assert s == 1 or cols // s >= 256
    EE = dfepe.compat.ErrorEstimators

    def flat_of(net):
        mods, out, i = list(net.fw), [], 0
        while i + 2 < len(mods):
            out += [mods[i].weight, mods[i].bias, mods[i + 1].weight, mods[i + 1].bias]
            i += 3
        return out + [mods[i].weight, mods[i].bias], (len(mods) - 1) // 3

    x = torch.zeros(2, 7, 100)
    flat, n = flat_of(EE.ErrorEstimator(7))
    assert est._pass_ok(x, flat, n)
    flat4, n4 = flat_of(EE.ErrorEstimator(7, output_size=4))
    assert not est._pass_ok(x, flat4, n4)                                    # four head channels: the per-launch host code
    half, nh = flat_of(EE.ErrorEstimator(7).half())
    assert not est._pass_ok(x, half, nh)                                     # not fp32
    assert not est._pass_ok(torch.zeros(2, 4, 100), flat, n)                 # the first layer does not take these inputs
    odd = [p for p in flat]
    odd[0] = torch.nn.Parameter(torch.zeros(64, 14, 1)[:, ::2])              # non-contiguous weight: fine since round 6 (the
    assert est._pass_ok(x, odd, n)                                           # parameters are packed into one dense vector first)
    broken = [p for p in flat]
    broken[4] = torch.nn.Parameter(torch.zeros(128, 32, 1))                  # a chain of widths that does not close
    assert not est._pass_ok(x, broken, n)
    net = EE.ErrorEstimator(7)
    net.half()
    net.float()                                                              # same Parameter objects, dtype back
    flat2, n2 = flat_of(net)
    assert est._pass_ok(x, flat2, n2)
    net.half()
    assert not est._pass_ok(x, flat_of(net)[0], n2)                          # every tensor is looked at on every call: no stale answer
    flat3, n3 = flat_of(EE.ErrorEstimator(7))
    flat3[6].data = flat3[6].data.half()                                     # ONE parameter's data swapped under the same object
    assert not est._pass_ok(x, flat3, n3)
    assert est.prepare([tuple(flat[4 * l:4 * l + 4]) for l in range(n)], (flat[-2], flat[-1])) is None  # CPU parameters: nothing to prepare


def test_fused_tail_is_reused_only_for_the_announced_ground_truth(dfepe):
    """get_Rt_loss takes the pose errors of get_all_loss_DeepF's fused launch only when it is handed the very objects that launch
    was given (or device tensors over the same memory): equal VALUES in other objects recompute -- slower, never wrong."""
    tgu = dfepe.compat.train_good_utils
    q, t, d = torch.zeros(4, 4), torch.zeros(4, 3), torch.eye(4).repeat(4, 1, 1)
    entry = {"gt_src": (q, t, d), "gt_dev": (q, t, d)}
    assert tgu._same_gt(entry, (q, t, d), (q, t, d))
    assert not tgu._same_gt(entry, (q.clone(), t, d), (q.clone(), t, d))          # another object, on the host: no identity, no device memory
    assert not tgu._same_gt(entry, (q.numpy(), t, d), (q, t, d))                  # numpy view of the same memory: still not the object
    assert not tgu._same_gt(entry, (t, q, d), (t, q, d))                          # order
