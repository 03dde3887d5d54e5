"""The small host-side helpers of the mirrored dsac_tools modules against the reference's own outputs
(tests/golden/helpers.npz, tests/golden/make_golden_helpers.py).  Plain numpy / elementwise torch like the reference: no GPU."""
import numpy as np
import torch


def test_E_F_from_Rt_and_E_to_F(dfepe, golden):
    g = golden("helpers")
    uF = dfepe.compat.utils_F
    R, t, K = g["R"][:12], g["t"], g["K"]
    for b in range(12):
        E, F = uF.E_F_from_Rt_np(R[b], t[b], K[b])
        np.testing.assert_allclose(E, g["E_np"][b], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(F, g["F_np"][b], rtol=1e-10, atol=1e-18)
        Et, Ft = uF._E_F_from_Rt(R[b], t[b], K[b])
        assert Et.dtype == torch.float64
        np.testing.assert_allclose(Et.numpy(), g["E_th"][b], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(Ft.numpy(), g["F_th"][b], rtol=1e-10, atol=1e-18)
        np.testing.assert_allclose(uF.E_to_F_np(g["E_np"][b], K[b]), g["E_to_F_np"][b], rtol=1e-10, atol=1e-18)
    Eb, Fb = uF._E_F_from_Rt(torch.from_numpy(R), torch.from_numpy(t), torch.from_numpy(K), tensor_input=True)
    np.testing.assert_allclose(Eb.numpy(), g["E_th_batch"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(Fb.numpy(), g["F_th_batch"], rtol=1e-10, atol=1e-18)
    # the batched numpy forms, which raise in the reference (utils_F.py:475,845), agree with the per-sample ones
    En, Fn = uF.E_F_from_Rt_np(R, t, K)
    np.testing.assert_allclose(Fn, g["F_np"], rtol=1e-10, atol=1e-18)
    np.testing.assert_allclose(uF.E_to_F_np(g["E_np"], K), g["E_to_F_np"], rtol=1e-10, atol=1e-18)
    # differentiable like the reference's
    tt = torch.from_numpy(t[0]).requires_grad_(True)
    uF._E_F_from_Rt(torch.from_numpy(R[0]), tt, torch.from_numpy(K[0]), tensor_input=True)[1].sum().backward()
    assert tt.grad is not None and torch.isfinite(tt.grad).all()


def test_quaternion_and_angle_helpers(dfepe, golden):
    g = golden("helpers")
    uG = dfepe.compat.utils_geo
    Rs = g["R"]
    assert set(g["q_branch"].tolist()) == {0, 1}  # the fixture reaches the large-angle branches of the trace method
    for i in range(len(Rs)):
        q = uG.R_to_q_np(Rs[i])
        assert q.dtype == np.float32 and q.shape == (4, 1) and q[0, 0] >= 0
        np.testing.assert_array_equal(q, g["q"][i])
        np.testing.assert_allclose(uG.q_to_R_np(g["q"][i].astype(np.float64) * (1.0 + 0.3 * i)), g["R_from_q"][i], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(uG.q_to_R_np(q.astype(np.float64)), Rs[i], atol=2e-6)  # round trip (q is float32)
        a = uG._rot_angle_error(torch.from_numpy(Rs[i]), torch.from_numpy(Rs[(i + 1) % len(Rs)]))
        assert torch.is_tensor(a) and a.dim() == 0
        np.testing.assert_allclose(float(a), g["rot_angle"][i], rtol=1e-12)
    np.testing.assert_allclose(uG.vectors_angle(g["v1"], g["v2"]), g["vectors_angle"], rtol=1e-13)
    # the same synthetic-scene quaternions the generator builds its ground truth with (synth.rotation_to_quaternion_np)
    np.testing.assert_allclose(uG.R_to_q_np(Rs[0])[:, 0], dfepe.synth.rotation_to_quaternion_np(Rs[0]).reshape(-1), atol=1e-6)
