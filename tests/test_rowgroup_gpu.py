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
    out = torch.zeros(14, 64, dtype=torch.float64, device=DEV)
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
