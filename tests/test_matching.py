"""Match construction: oracle vs the reference's own get_matches_from_SP / crop_or_pad_choice run
(tests/golden/matching.npz, CPU), and the HIP path vs the oracle / the golden vectors (GPU, through the C ABI)."""
import numpy as np
import pytest
import torch

DEV = "cuda:0"


def T(x):
    return torch.from_numpy(np.asarray(x))


def _sp_inputs(g, tag):
    xs = [T(g[f"{tag}_pts{i}"]) for i in range(2)]
    des = [T(g[f"{tag}_des{i}"]) for i in range(2)]
    res = [T(g[f"{tag}_res{i}"]) for i in range(2)]
    N, out_n, thr, seed = g[f"{tag}_cfg"]
    return xs, des, res, int(out_n), float(thr), int(seed)


# ---------------------------------------------------------------- CPU: oracle pinned by the reference's run
def test_crop_or_pad_choice_matches_reference(oracle, dfepe, golden):
    g = golden("matching")
    for k, (a, b, s) in enumerate(g["cp_cases"]):
        for fn in (oracle.crop_or_pad_choice, dfepe.compat.utils_misc.crop_or_pad_choice):
            np.random.seed(100 + k)
            np.testing.assert_array_equal(fn(int(a), int(b), shuffle=bool(s)), g[f"cp_choice_{k}"])


@pytest.mark.parametrize("tag", ["crop", "pad"])
def test_oracle_match_construction_matches_reference(oracle, golden, tag):
    g = golden("matching")
    xs, des, res, out_n, thr, seed = _sp_inputs(g, tag)
    np.random.seed(seed)
    r = oracle.matches_from_sp_outputs(xs, des, res, thr, out_n)
    np.testing.assert_array_equal(r["num_matches"].numpy(), g[f"{tag}_num_matches"])
    np.testing.assert_array_equal(r["xs"].numpy(), g[f"{tag}_xs"])
    np.testing.assert_array_equal(r["offsets"].numpy(), g[f"{tag}_offsets"])
    np.testing.assert_allclose(r["quality"].numpy(), g[f"{tag}_quality"], rtol=0, atol=1e-7)
    np.testing.assert_array_equal(r["xs_SP"][0].numpy(), g[f"{tag}_xs_SP0"])


def test_oracle_nn_match_properties(oracle):
    """Known answers of the published routine: a permuted copy matches exactly (distance 0), in increasing index order;
    duplicates resolve to the first index; the threshold is strict; empty inputs give a [3,0] array."""
    rng = np.random.default_rng(0)
    d1 = rng.standard_normal((32, 20)); d1 /= np.linalg.norm(d1, axis=0)
    perm = rng.permutation(20)
    m = oracle.nn_match_two_way(d1, d1[:, perm], 0.5)
    assert m.shape == (3, 20) and (m[0] == np.arange(20)).all() and (perm[m[1].astype(int)] == np.arange(20)).all()
    assert np.abs(m[2]).max() < 1e-3
    d2 = np.concatenate((d1[:, :1], d1), axis=1)            # column 0 duplicated in front: ties go to index 0
    m = oracle.nn_match_two_way(d1, d2, 0.5)
    assert m[1, 0] == 0 and (m[1, 1:] == np.arange(2, 21)).all()
    assert oracle.nn_match_two_way(d1, d1, 0.0).shape == (3, 0)   # scores < 0 never holds
    assert oracle.nn_match_two_way(d1[:, :0], d1, 0.5).shape == (3, 0)
    with pytest.raises(ValueError):
        oracle.nn_match_two_way(d1, d1, -1.0)


def test_matching_abi_validation(dfepe):
    L = dfepe._lib.lib()
    assert L.dfepe_nn_match_workspace_bytes(2, 10, 20) == 2 * 30 * 8
    assert L.dfepe_nn_match_workspace_bytes(0, 10, 20) == 0
    assert L.dfepe_nn_match_two_way(None, None, 0, 10, 10, 256, 0.7, None, None, None, None, None, None) == 0   # empty batch
    assert L.dfepe_nn_match_two_way(None, None, 2, 10, 10, 256, -0.1, None, None, None, None, None, None) == -1  # ValueError in the reference
    assert L.dfepe_nn_match_two_way(None, None, 2, 10, 10, 256, 0.7, None, None, None, None, None, None) == -1
    assert L.dfepe_gather_matches(None, None, None, None, 2, 10, 10, None, None, None, None, 8, None, None, None, None) == -1
    assert L.dfepe_gather_matches(None, None, None, None, 0, 10, 10, None, None, None, None, 8, None, None, None, None) == 0
    with pytest.raises(dfepe.DfepeError):
        dfepe.ops.nn_match_two_way(torch.zeros(1, 8, 32), torch.zeros(1, 8, 32), 0.7)  # CPU tensors: no fallback


# ---------------------------------------------------------------- GPU: HIP path vs oracle / golden
def _rand_desc(B, N1, N2, D, seed, common=0.5):
    g = torch.Generator().manual_seed(seed)
    d1 = torch.nn.functional.normalize(torch.randn(B, N1, D, generator=g), dim=2)
    d2 = torch.nn.functional.normalize(torch.randn(B, N2, D, generator=g), dim=2)
    n = int(common * min(N1, N2))
    for b in range(B):
        src = torch.randperm(N1, generator=g)[:n]
        dst = torch.randperm(N2, generator=g)[:n]
        scale = 0.05 + 1.2 * torch.rand(n, 1, generator=g)
        d2[b, dst] = torch.nn.functional.normalize(d1[b, src] + scale * torch.randn(n, D, generator=g) / D ** 0.5, dim=1)
    return d1, d2


@pytest.mark.gpu
@pytest.mark.parametrize("B,N1,N2,D,thr", [(3, 200, 180, 256, 0.7), (2, 1000, 1100, 256, 1.0), (1, 5, 300, 64, 0.9),
                                           (2, 129, 127, 32, 1.2), (4, 128, 256, 256, 0.5)])
def test_nn_match_two_way_vs_oracle(dfepe, oracle, B, N1, N2, D, thr):
    d1, d2 = _rand_desc(B, N1, N2, D, seed=N1 + N2)
    m1, m2, sc, cnt = dfepe.ops.nn_match_two_way(d1.to(DEV), d2.to(DEV), thr)
    for b in range(B):
        ref = oracle.nn_match_two_way(d1[b].numpy().T, d2[b].numpy().T, thr)
        n = int(cnt[b].item())
        assert n == ref.shape[1], (b, n, ref.shape[1])
        np.testing.assert_array_equal(m1[b, :n].cpu().numpy(), ref[0].astype(np.int64))
        np.testing.assert_array_equal(m2[b, :n].cpu().numpy(), ref[1].astype(np.int64))
        np.testing.assert_allclose(sc[b, :n].cpu().numpy(), ref[2], rtol=0, atol=2e-6)  # fp32 dot of 256 terms, then sqrt


@pytest.mark.gpu
def test_nn_match_ties_and_edges(dfepe, oracle):
    g = torch.Generator().manual_seed(3)
    d1 = torch.nn.functional.normalize(torch.randn(1, 40, 64, generator=g), dim=2)
    d2 = torch.cat((d1[:, :1], d1, d1[:, 5:6]), dim=1)        # duplicates: numpy's argmin keeps the first occurrence
    m1, m2, sc, cnt = dfepe.ops.nn_match_two_way(d1.to(DEV), d2.to(DEV), 0.5)
    ref = oracle.nn_match_two_way(d1[0].numpy().T, d2[0].numpy().T, 0.5)
    n = int(cnt[0].item())
    assert n == ref.shape[1]
    np.testing.assert_array_equal(m2[0, :n].cpu().numpy(), ref[1].astype(np.int64))
    # all descriptors identical: every distance is 0, first-occurrence arg-min leaves exactly the match (0, 0)
    same = d1[:, :1].expand(1, 37, 64).contiguous().to(DEV)
    s1, s2, ssc, scnt = dfepe.ops.nn_match_two_way(same, same[:, :29].contiguous(), 0.5)
    refs = oracle.nn_match_two_way(same[0].cpu().numpy().T, same[0, :29].cpu().numpy().T, 0.5)
    assert int(scnt[0].item()) == refs.shape[1] == 1 and int(s1[0, 0]) == 0 and int(s2[0, 0]) == 0
    # a single keypoint on one side
    o1, o2, osc, ocnt = dfepe.ops.nn_match_two_way(d1[:, 7:8].contiguous().to(DEV), d1.to(DEV), 0.5)
    assert int(ocnt[0].item()) == 1 and int(o2[0, 0]) == 7 and float(osc[0, 0]) < 1e-3
    # threshold 0: nothing is < 0
    assert int(dfepe.ops.nn_match_two_way(d1.to(DEV), d1.to(DEV), 0.0)[3].item()) == 0
    # empty sides
    e = torch.zeros(2, 0, 64, device=DEV)
    assert dfepe.ops.nn_match_two_way(e, d1.expand(2, 40, 64).contiguous().to(DEV), 0.7)[3].tolist() == [0, 0]
    assert dfepe.ops.nn_match_two_way(d1.expand(2, 40, 64).contiguous().to(DEV), e, 0.7)[3].tolist() == [0, 0]
    with pytest.raises(ValueError):
        dfepe.ops.nn_match_two_way(d1.to(DEV), d1.to(DEV), -1.0)
    with pytest.raises(dfepe.DfepeError):  # D not a multiple of 32
        dfepe.ops.nn_match_two_way(torch.zeros(1, 8, 48, device=DEV), torch.zeros(1, 8, 48, device=DEV), 0.7)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["crop", "pad"])
def test_match_construction_matches_reference_golden(dfepe, golden, tag):
    """compat.train_good_utils.matches_from_SP_outputs == what the reference's get_matches_from_SP returned."""
    g = golden("matching")
    xs, des, res, out_n, thr, seed = _sp_inputs(g, tag)
    np.random.seed(seed)
    r = dfepe.compat.train_good_utils.matches_from_SP_outputs([x.to(DEV) for x in xs], [d.to(DEV) for d in des],
                                                               [x.to(DEV) for x in res], thr, out_n)
    np.testing.assert_array_equal(r["num_matches"].numpy(), g[f"{tag}_num_matches"])
    np.testing.assert_array_equal(r["xs"].cpu().numpy(), g[f"{tag}_xs"])
    np.testing.assert_array_equal(r["offsets"].cpu().numpy(), g[f"{tag}_offsets"])
    np.testing.assert_allclose(r["quality"].cpu().numpy(), g[f"{tag}_quality"], rtol=0, atol=2e-6)
    np.testing.assert_array_equal(r["xs_SP"][0].cpu().numpy(), g[f"{tag}_xs_SP0"])


@pytest.mark.gpu
def test_get_matches_from_SP_call_surface_and_tracker(dfepe, oracle, golden):
    """Same call as the reference (fake front-end objects) and the PointTracker drop-in on host arrays."""
    g = golden("matching")
    xs, des, res, out_n, thr, seed = _sp_inputs(g, "crop")
    outs = iter([{"pts_int": xs[i].to(DEV), "pts_desc": des[i].to(DEV), "pts_offset": res[i].to(DEV)} for i in range(2)])
    tracker = dfepe.compat.model_wrap.PointTracker(max_length=2, nn_thresh=thr)
    np.random.seed(seed)
    r = dfepe.compat.train_good_utils.get_matches_from_SP([torch.zeros(2, 8, 8), torch.zeros(2, 8, 8)], lambda img: next(outs), None,
                                                           tracker, out_num_points=out_n, process_SP_output=lambda o, p: o)
    np.testing.assert_array_equal(r["xs"].cpu().numpy(), g["crop_xs"])
    assert set(r) == {"xs", "offsets", "quality", "num_matches", "xs_SP"} and r["quality"].shape == (2, out_n, 1)
    m = tracker.nn_match_two_way(des[0][0].numpy().T, des[1][0].numpy().T, thr)
    ref = oracle.nn_match_two_way(des[0][0].numpy().T, des[1][0].numpy().T, thr)
    assert m.dtype == np.float64 and m.shape == ref.shape
    np.testing.assert_array_equal(m[:2], ref[:2])
    np.testing.assert_allclose(m[2], ref[2], atol=2e-6)


@pytest.mark.gpu
def test_nn_match_full_size_properties(dfepe):
    """SuperPoint-sized batch (1024 keypoints, D=256): matches are mutual, unique, sorted and below the threshold."""
    B, N, D = 16, 1024, 256
    d1, d2 = _rand_desc(B, N, N, D, seed=11)
    d1, d2 = d1.to(DEV), d2.to(DEV)
    m1, m2, sc, cnt = dfepe.ops.nn_match_two_way(d1, d2, 0.8)
    dm = torch.sqrt((2 - 2 * torch.clamp(torch.bmm(d1, d2.transpose(1, 2)), -1, 1)).clamp_min(0))
    for b in range(B):
        n = int(cnt[b].item())
        assert 0.3 * N < n <= N
        i, j, s = m1[b, :n].long(), m2[b, :n].long(), sc[b, :n]
        assert (i[1:] > i[:-1]).all() and j.unique().numel() == n and (s < 0.8).all()
        assert (dm[b, i].argmin(dim=1) == j).all() and (dm[b][:, j].argmin(dim=0) == i).all()
        assert (dm[b, i, j] - s).abs().max().item() < 5e-6
