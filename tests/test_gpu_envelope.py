"""GPU parity rungs added in round 2, all against numbers produced by the REAL reference (tests/golden/make_golden_r2.py):

* G15 -- the trajectory ladder of G5 at 2048x2048 points, the cloud size of BASELINE configs[1] and of the kernel
  bench.py times (solve_kernel<512,4,4>);
* G16 / G17 / G18 -- the chaos-envelope rung.  HOUV trajectories are chaotic (registration/README.md:82-91 calls the
  results non-reproducible): the reference itself, re-run on inputs perturbed by a relative 1e-7,
  leaves most hypotheses bit-stable and sends a few far away.  The fixtures hold the reference's (min_1, R, T) of
  every hypothesis at several horizons for the clean inputs and for two perturbed runs; the larger of the two
  divergences is the reference's own ENVELOPE.  The fused kernel must stay inside it: bulk quantiles of the
  GPU-vs-reference divergence within 3x the envelope's, and no more hypotheses beyond each decade threshold than 3x the
  envelope's count (+3).  A kernel with a bias (wrong gradient term, different tie rule, wrong Adam constant) moves
  EVERY hypothesis by far more than 1e-7 and fails the bulk quantiles immediately."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

T = torch.tensor


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch.device("cuda:0")


def test_golden_trajectory_2048_points(golden, dev, solver_mode):
    """G15: houv.py:106-138 on ONE 2048x2048-point pair, K=26, bases 0 and 2.  Forward n+1 of the reference reads the
    parameters after n Adam steps, so forwards 2/3/6/21 pin the 1/2/5/20-step states with G5's tolerances; the step-1
    gradient is compared like G5's."""
    from houv_amd import _lib, solver
    from houv_amd.models.houv import HOUV, predict_model
    g = golden("g15_traj2048.npz")
    K = int(g["kernel"])
    s, t = T(g["src"]).to(dev), T(g["tgt"]).to(dev)
    assert s.shape[1] == 2048 and _lib.solve_variant(2048, 2048) == (512, 4)
    for base in (0, 2):
        p0 = solver.houv_init_params(K, 2021)
        np.testing.assert_array_equal(p0.astype(np.float32), g[f"b{base}_n1_params"])       # identical starting point
        out, _ = solver.run_stage(s, t, p0, K, 1, angle_base=base, trans_mode=0, use_views=True, f64_params=False,
                                  lr=0.01, want_grad=True)
        ref_g = g[f"b{base}_grad"]
        scale = np.abs(ref_g).max(axis=1, keepdims=True)
        err = (np.abs(out["grad"].cpu().numpy() - ref_g) / scale).max(axis=1)
        # fp32 near-tie flips of a nearest neighbour move one point's share (<= ~3/N) of a gradient: tolerate two hypotheses
        assert (err > 2e-4).sum() <= 2 and err.max() < 3.0 / 2048, err
        np.testing.assert_allclose(out["score"].cpu().numpy(), g[f"b{base}_n1_min1"], atol=1e-5)
        np.testing.assert_allclose(out["loss"].cpu().numpy(), g[f"b{base}_n1_loss"], atol=5e-5)
        for n_fwd, ptol, mtol, rtol in ((2, 2e-6, 1e-5, 1e-5), (3, 1e-5, 1e-5, 1e-5), (6, 5e-5, 1e-5, 1e-4), (21, 1e-3, 5e-5, 1e-3)):
            net = HOUV(K, 0)
            m1, R, Tt = predict_model(net, s, t, kernel=K, num_epochs=n_fwd, angle_base=base)
            # the reference's forward n_fwd read the parameters after n_fwd-1 steps; ours ends one step later, so
            # compare the state BEFORE our last step = run n_fwd-1 steps and read the net
            if n_fwd > 1:
                net2 = HOUV(K, 0)
                predict_model(net2, s, t, kernel=K, num_epochs=n_fwd - 1, angle_base=base)
                np.testing.assert_allclose(net2.packed_params().detach().cpu().numpy(), g[f"b{base}_n{n_fwd}_params"], atol=ptol)
            np.testing.assert_allclose(m1.cpu().numpy().reshape(-1), g[f"b{base}_n{n_fwd}_min1"], atol=mtol)
            np.testing.assert_allclose(R.cpu().numpy().reshape(-1, 3, 3), g[f"b{base}_n{n_fwd}_R"], atol=rtol)
            np.testing.assert_allclose(Tt.cpu().numpy().reshape(-1, 3), g[f"b{base}_n{n_fwd}_T"], atol=rtol)


def _divergence(a, ref):
    return np.abs(a.reshape(len(ref), -1) - ref.reshape(len(ref), -1)).max(axis=1)


def _inside_envelope(gpu, env, floor, what):
    """Bulk quantiles within 3x the envelope's (+ an fp32 rounding floor); decade-threshold exceedance counts within 3x (+3).
    With fewer than 64 hypotheses (G20 / G21: 26) the 90th percentile IS the third-largest value, i.e. it is set by the two or
    three hypotheses that happened to diverge first: there the factor is 5 (round 3: three point orders of the same cloud --
    as given, Morton, k-d leaves -- measured 0.5x / 0.5x / 3.3x the envelope's q90 at one horizon and 0.2-0.9x at the others,
    scripts/probe_g20.py); the median keeps the factor 3 and the decade counts bound the tail either way."""
    for qq in (0.5, 0.9):
        factor = 5.0 if (qq > 0.5 and len(gpu) < 64) else 3.0
        assert np.quantile(gpu, qq) <= factor * np.quantile(env, qq) + floor, (what, qq, np.quantile(gpu, qq), np.quantile(env, qq))
    for thr in (1e-5, 1e-4, 1e-3, 1e-2, 1e-1):
        if thr > 10 * floor:
            assert (gpu > thr).sum() <= 3 * (env > thr).sum() + 3, (what, thr, int((gpu > thr).sum()), int((env > thr).sum()))


@pytest.mark.parametrize("fixture", ["g16_envelope128.npz", "g17_envelope512.npz", "g20_envelope2048.npz"])
def test_predict_model_stays_inside_the_references_chaos_envelope(golden, dev, fixture, solver_mode):
    """G16 (16 pairs x 128 points) / G17 (6 pairs x 512 points, BASELINE configs[0]'s cloud size) / G20 (G15's 2048 x 2048-point
    pair: BASELINE configs[1]'s cloud size, the kernel instantiation bench.py times), K=26, base 0: predict_model
    (houv.py:106-138) at 20/50/100/200 iterations, per hypothesis, against the reference and its envelope."""
    from houv_amd.models.houv import HOUV, predict_model
    g = golden(fixture)
    K = int(g["kernel"])
    s, t = T(g["src"]).to(dev), T(g["tgt"]).to(dev)
    for h in (int(x) for x in g["horizons"]):
        m1, R, Tt = predict_model(HOUV(s.shape[0] * K, 0), s, t, kernel=K, num_epochs=h, angle_base=0)
        for key, val, floor in (("R", R, 2.4e-7), ("T", Tt, 6e-8), ("min1", m1, 1.5e-8)):
            ref = g[f"ref_n{h}_{key}"]
            env = np.maximum(_divergence(g[f"pertA_n{h}_{key}"], ref), _divergence(g[f"pertB_n{h}_{key}"], ref))
            _inside_envelope(_divergence(val.cpu().numpy(), ref), env, floor, (fixture, h, key))


@pytest.mark.parametrize("fixture", ["g18_twin_envelope.npz", "g21_twin_envelope2048.npz"])
def test_solve_twin_stays_inside_the_references_chaos_envelope(golden, dev, fixture, solver_mode):
    """G21: the same on G15's 2048 x 2048-point pair (the single-metric kernel instantiation at BASELINE configs[1]'s size).
    G18: train_utils.getPredict_angle (train_utils.py:359-456: float64 leaves from the harness-seeded global numpy RNG,
    lr 0.1, sigma = sin(s pi), loss 6 min_1) on 8 pairs x 128 points, K=26, base 1, at 5/20/50/100 iterations.  At lr 0.1
    a fifth of the hypotheses has left the 1e-3 ball after 20 iterations in the reference's own perturbed runs; the
    kernel's divergence has the same distribution.  Also the 4th return value: tran_s of the LAST forward (:404,456)."""
    from houv_amd.train_utils import getPredict_angle
    g = golden(fixture)
    K = int(g["kernel"])
    s, t = T(g["src"]).to(dev), T(g["tgt"]).to(dev)
    for h in (int(x) for x in g["horizons"]):
        np.random.seed(int(g["np_seed"]))
        m1, R, Tt, ts = getPredict_angle(s, t, kernel=K, num_epochs=h, angle_base=1)
        for key, val, floor in (("R", R, 2.4e-7), ("T", Tt, 2.4e-7), ("min1", m1, 1.5e-8)):
            ref = g[f"ref_n{h}_{key}"]
            env = np.maximum(_divergence(g[f"pertA_n{h}_{key}"], ref), _divergence(g[f"pertB_n{h}_{key}"], ref))
            _inside_envelope(_divergence(val.cpu().numpy(), ref), env, floor, (fixture, h, key))
    # tran_s belongs to the last forward, like the R / T / min_1 it is returned with: |T| == |sigma|
    ref_ts = g["ref_tran_s"]
    env = np.maximum(np.abs(g["pertA_tran_s"] - ref_ts), np.abs(g["pertB_tran_s"] - ref_ts)).reshape(-1)
    _inside_envelope(np.abs(ts.cpu().numpy() - ref_ts).reshape(-1), env, 2.4e-7, (fixture, "tran_s"))
    np.testing.assert_allclose(np.abs(ts.cpu().numpy().reshape(-1)), Tt.reshape(-1, 3).norm(dim=1).cpu().numpy(), atol=1e-5)
