"""The DPP row primitives of csrc/rowgroup.h on the hardware against their definition (what tests/emu/ assumes)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_rowgroup_primitives_match_their_definition(dfepe):
    rng = np.random.default_rng(0)
    x = rng.standard_normal(64)
    y = rng.standard_normal(64)
    xd, yd = torch.from_numpy(x).to(DEV), torch.from_numpy(y).to(DEV)
    out = torch.zeros(36, 64, dtype=torch.float64, device=DEV)
    rc = dfepe._lib.lib().dfepe_selftest_rowgroup(xd.data_ptr(), yd.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    o = out.cpu().numpy()
    X = x.reshape(4, 16)
    lane = np.arange(16)
    rows = lambda v: np.repeat(np.asarray(v)[:, None], 16, 1).reshape(64)
    np.testing.assert_array_equal(o[0], rows(X[:, 0]))
    np.testing.assert_array_equal(o[1], rows(X[:, 5]))
    np.testing.assert_array_equal(o[2], rows(X[:, 15]))
    np.testing.assert_allclose(o[3], rows(X.sum(1)), rtol=1e-14, atol=1e-14)
    np.testing.assert_allclose(o[4], rows(X[:, 2:9].sum(1)), rtol=1e-14, atol=1e-14)
    np.testing.assert_array_equal(o[5], X[:, 15 - lane].reshape(64))
    np.testing.assert_array_equal(o[6], X[:, lane ^ 7].reshape(64))
    np.testing.assert_array_equal(o[7], X[:, lane ^ 2].reshape(64))
    np.testing.assert_array_equal(o[8], X[:, lane ^ 1].reshape(64))
    Xf = X.astype(np.float32)
    np.testing.assert_array_equal(o[9], rows(Xf.max(1)).astype(np.float64))
    Xi = (X * 16.0).astype(np.int32)
    np.testing.assert_array_equal(o[10], rows(Xi.sum(1)).astype(np.float64))
    np.testing.assert_allclose(o[11], y + rows(X[:, 3]) * y, rtol=1e-15, atol=1e-15)
    np.testing.assert_allclose(o[12], rows(Xf.sum(1)), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(o[13], rows(Xf[:, 9].astype(np.float64) + Xi[:, 12]) + np.tile(lane, 4), rtol=0, atol=1e-12)
    # fused broadcast-FMA chains (v_fmac_f64_dpp inside asm statements): dot, axpy, axpy2, sum over lanes J0..8
    Y = y.reshape(4, 16)
    j9 = np.arange(9, dtype=np.float64)
    m = x[:, None] + j9[None, :] * y[:, None]          # [64, 9], per-lane arrays
    n = y[:, None] - j9[None, :]
    bc = lambda v, j: rows(np.asarray(v).reshape(4, 16)[:, j])   # lane j of the row, in every lane
    xa, xb, xc, xd_, xe, xf = x * y + 1.0, x - y, x * 3.0, y * y, x + 2.0, x * x
    np.testing.assert_allclose(o[14], sum(bc(xa, j) * m[:, j] for j in range(1, 9)), rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(o[15], sum(bc(xb, j) * m[:, j] for j in range(6, 9)), rtol=1e-13, atol=1e-13)
    m2 = m.copy()
    for j in range(3, 9):
        m2[:, j] += bc(xc, j) * y
    np.testing.assert_allclose(o[16:25].T, m2, rtol=1e-13, atol=1e-13)
    n2 = n.copy()
    for j in range(1, 9):
        n2[:, j] += bc(xd_, j) * x + bc(xe, j) * y
    np.testing.assert_allclose(o[25:34].T, n2, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(o[34], sum(bc(xf, j) for j in range(0, 9)), rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(o[35], sum(bc(xf + 1.0, j) for j in range(5, 9)), rtol=1e-13, atol=1e-13)
