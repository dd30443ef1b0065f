"""Pin the CPU oracle (oracle/houv_ref_cpu.py) against golden vectors captured from
the real reference by tests/golden/make_golden.py.  CPU only."""
import numpy as np
import torch

from oracle import houv_ref_cpu as orc

T = torch.tensor


def test_g1_chamfer_forward_unit_test_shapes(golden):
    g = golden("g1_chamfer.npz")
    d1, d2, i1, i2 = orc.chamfer_nn(T(g["p1"]), T(g["p2"]))
    # reference's own bar (utils/metrics/CD/unit_test.py:23-33): idx exact, mean sq diff < 1e-8
    assert np.array_equal(i1.numpy(), g["idx1"]) and np.array_equal(i2.numpy(), g["idx2"])
    assert np.array_equal(d1.numpy(), g["dist1"]) and np.array_equal(d2.numpy(), g["dist2"])
    bd1, bd2, bi1, bi2 = orc.chamfer_nn_chunked(T(g["big_a"]), T(g["big_b"]), chunk=1)
    assert np.array_equal(bi1.numpy(), g["big_idx1"]) and np.array_equal(bi2.numpy(), g["big_idx2"])
    assert np.array_equal(bd1.numpy(), g["big_dist1"]) and np.array_equal(bd2.numpy(), g["big_dist2"])


def test_g1_chamfer_backward(golden):
    g = golden("g1_chamfer.npz")
    p2 = T(g["p2"]).requires_grad_(True)
    d1, _, _, _ = orc.chamfer_nn(T(g["p1"]), p2)
    d1.sum().backward()
    np.testing.assert_allclose(p2.grad.numpy(), g["grad_p2"], rtol=0, atol=1e-7)
    # closed form (the CUDA kernel's formula) == autograd through the python op
    q1, q2 = T(g["q1"]), T(g["q2"])
    gx1, gx2 = orc.chamfer_backward_closed_form(q1, q2, T(g["j1"]), T(g["j2"]), T(g["w1"]), T(g["w2"]))
    np.testing.assert_allclose(gx1.numpy(), g["grad_q1"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gx2.numpy(), g["grad_q2"], rtol=1e-5, atol=1e-6)


def test_g2_loss_glue(golden):
    g = golden("g2_loss.npz")
    mv = T(g["moved"]).requires_grad_(True)
    tg = T(g["target"])
    c = orc.calc_cd_percent(mv, tg, percent=0.5)
    np.testing.assert_array_equal(np.stack([x.detach().numpy() for x in c]), g["cd_percent"])
    for d in range(3):
        v = orc.loss_view(mv, tg, dim=d)
        np.testing.assert_array_equal(np.stack([x.detach().numpy() for x in v]), g["views"][d])
    loss, min1 = orc.predict_loss(mv, tg)
    loss.mean().backward()
    np.testing.assert_array_equal(loss.detach().numpy(), g["loss"])
    np.testing.assert_array_equal(min1.detach().numpy(), g["min_1"])
    np.testing.assert_allclose(mv.grad.numpy(), g["grad_moved"], rtol=0, atol=1e-9)


def test_g3_g4_params_and_forward(golden):
    g = golden("g3_g4_params_forward.npz")
    V, a, tc, ts = orc.houv_init_params(32, seed=2021)
    for mine, key in ((V, "V"), (a, "angle"), (tc, "tran_c"), (ts, "tran_s")):
        np.testing.assert_array_equal(mine, g[key])
    # quirk A.5(2)/(3): lattice rows first, shared normal draws
    assert np.array_equal(V[:26], orc.LATTICE_AXES.astype(np.float32))
    assert np.array_equal(a[:, 0], ts[:, 0])
    for base in range(4):
        mv, R, Tt = orc.houv_forward(T(g["src"]), T(V), T(a), T(tc), T(ts), base)
        np.testing.assert_array_equal(mv.numpy(), g[f"moved_b{base}"])
        np.testing.assert_array_equal(R.numpy(), g[f"R_b{base}"])
        np.testing.assert_array_equal(Tt.numpy(), g[f"T_b{base}"])


def test_g5_trajectory(golden):
    g = golden("g5_trajectory.npz")
    s, t = T(g["src"]), T(g["tgt"])
    for base in (0, 2):
        tr = {"steps": (1, 2, 5, 20), "want_grads": True}
        m1, R, Tt = orc.predict_model(s, t, kernel=16, num_epochs=20, angle_base=base, trace=tr)
        np.testing.assert_array_equal(m1.numpy(), g[f"b{base}_n20_min1"])
        np.testing.assert_array_equal(R.numpy(), g[f"b{base}_n20_R"])
        np.testing.assert_array_equal(Tt.numpy(), g[f"b{base}_n20_T"])
        for n in (1, 2, 5, 20):
            for p, key in zip(tr["params"][n], ("V", "angle", "tran_c", "tran_s")):
                np.testing.assert_array_equal(p, g[f"b{base}_n{n}_{key}"])
        for p, key in zip(tr["grads0"], ("V", "angle", "tran_c", "tran_s")):
            np.testing.assert_array_equal(p, g[f"b{base}_grad_{key}"])


def test_g6_solve_model_and_solve(golden):
    g = golden("g6_solve.npz")
    s, t, pose = T(g["src"]), T(g["tgt"]), T(g["pose"])
    r_err, t_err, ans = orc.solve_model(s, t, pose, kernel=16, num_epochs=30)
    np.testing.assert_array_equal(ans.numpy(), g["ans"])
    np.testing.assert_array_equal(r_err.numpy(), g["r_err"])
    np.testing.assert_array_equal(t_err.numpy(), g["t_err"])
    assert np.all(ans.numpy()[:, 3, :] == 0)            # quirk A.5(1): bottom row stays zero
    # the fixture does exercise the retry stage
    assert (g["base0_min1"].min(axis=1) > orc.RETRY_THRESHOLD).any()
    np.testing.assert_array_equal(orc.solve_model(s, t, None, kernel=16, num_epochs=30, prefix="test").numpy(),
                                  g["ans_test"])
    np.random.seed(int(g["gpa_np_seed"]))
    m1, R, Tt = orc.get_predict_angle(T(g["solve_src"]), T(g["solve_tgt"]), kernel=4, num_epochs=20, angle_base=1)
    np.testing.assert_array_equal(m1.numpy(), g["gpa_min1"])
    np.testing.assert_array_equal(R.numpy(), g["gpa_R"])
    np.testing.assert_array_equal(Tt.numpy(), g["gpa_T"])
    np.random.seed(int(g["solve_np_seed"]))
    ans = orc.solve(T(g["solve_src"]), T(g["solve_tgt"]), kernel=4, prefix="test")
    np.testing.assert_array_equal(ans.numpy(), g["solve_ans"])


def test_g7_svdhead(golden):
    g = golden("g7_svdhead.npz")
    R, t = orc.kabsch_svd(T(g["src"]), T(g["corr"]))
    np.testing.assert_allclose(R.numpy(), g["R"], atol=1e-6)
    np.testing.assert_allclose(t.numpy(), g["t"], atol=1e-6)
    Rw, tw = orc.kabsch_svd(T(g["src"]), T(g["corr"]), T(g["w"]))
    np.testing.assert_allclose(Rw.numpy(), g["R_w"], atol=1e-6)
    np.testing.assert_allclose(tw.numpy(), g["t_w"], atol=1e-6)
    assert np.all(g["det"] > 0.99)                       # reflection fix applied by the reference


def test_g8_metrics(golden):
    g = golden("g8_metrics.npz")
    Ta, Tb = T(g["Ta"]), T(g["Tb"])
    np.testing.assert_array_equal(orc.rotation_error(Ta[:, :3, :3], Tb[:, :3, :3]).numpy(), g["rot_err"])
    np.testing.assert_array_equal(orc.translation_error(Ta[:, :3, 3], Tb[:, :3, 3]).numpy(), g["trans_err"])
    np.testing.assert_array_equal(orc.rmse_loss(T(g["pts"]), Ta, Tb).numpy(), g["rmse"])


def test_c_oracle_pinned_to_golden(golden):
    """oracle/chamfer_ref.c (fp32 direct-difference restatement of chamfer3D.cu) against the reference's outputs:
    indices exactly equal (unit_test.py:29-33), distances = fp32 rounding noise of the float64 values."""
    from oracle import c_oracle
    g = golden("g1_chamfer.npz")
    d1, d2, i1, i2 = c_oracle.chamfer_forward(g["p1"], g["p2"])
    assert np.array_equal(i1, g["idx1"]) and np.array_equal(i2, g["idx2"])
    np.testing.assert_allclose(d1, g["dist1"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(d2, g["dist2"], rtol=2e-5, atol=1e-7)
    assert ((d1 - g["dist1"]) ** 2).mean() + ((d2 - g["dist2"]) ** 2).mean() < 1e-8      # unit_test.py:23-27
    b1, b2, j1, j2 = c_oracle.chamfer_forward(g["big_a"], g["big_b"])
    assert (j1 != g["big_idx1"]).sum() + (j2 != g["big_idx2"]).sum() <= 2                 # fp32 near-ties only
    np.testing.assert_allclose(b1, g["big_dist1"], rtol=1e-4, atol=1e-7)
    gx1, gx2 = c_oracle.chamfer_backward(g["q1"], g["q2"], g["w1"], g["w2"], g["j1"], g["j2"])
    np.testing.assert_allclose(gx1, g["grad_q1"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(gx2, g["grad_q2"], rtol=1e-4, atol=2e-6)
