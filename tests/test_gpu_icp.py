"""GPU: houv_icp_refine against oracle/icp_ref.py (numpy restatement of Open3D's published point-to-point ICP).
PARITY UNPINNED w.r.t. Open3D itself (not installed, no fixtures in the reference) -- see oracle/icp_ref.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import icp_ref  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _perturb(T, rng, ang_deg, tr):
    ax = rng.standard_normal(3); ax /= np.linalg.norm(ax)
    a = np.deg2rad(ang_deg)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K
    P = np.eye(4); P[:3, :3] = R; P[:3, 3] = rng.standard_normal(3) * tr
    return (P @ T).astype(np.float32)


@pytest.mark.parametrize("N", [200, 700, 2048])
def test_icp_matches_oracle(dev, N):
    from houv_amd import ops, synthetic
    P = 6
    src, tgt, pose = synthetic.make_pairs(P, N, seed=31)
    rng = np.random.default_rng(N)
    init = np.stack([_perturb(pose[i].numpy(), rng, 4.0, 0.01) for i in range(P)])
    out = ops.icp_refine(src.to(dev), tgt.to(dev), torch.tensor(init).to(dev), 0.04, 60)
    for i in range(P):
        T, fit, rmse, it = icp_ref.icp_point_to_point(src[i].numpy(), tgt[i].numpy(), init[i], 0.04, 60)
        Tg = out["T"][i].cpu().numpy()
        assert np.array_equal(Tg[3], [0, 0, 0, 1])
        # fp32 kernel vs float64 oracle: same fixed point, same correspondences up to threshold near-ties
        np.testing.assert_allclose(Tg, T, atol=2e-3)
        assert abs(float(out["fitness"][i]) - fit) <= 3.0 / N
        assert abs(float(out["inlier_rmse"][i]) - rmse) <= 2e-4
        assert abs(int(out["iterations"][i]) - it) <= max(3, it // 4)


def test_icp_zero_iterations_and_identity(dev):
    from houv_amd import ops, synthetic
    src, tgt, pose = synthetic.make_pairs(3, 300, seed=5)
    out = ops.icp_refine(src.to(dev), tgt.to(dev), pose.to(dev), 0.02, 0)
    np.testing.assert_allclose(out["T"].cpu().numpy(), pose.numpy(), atol=1e-6)       # no update applied
    assert (out["iterations"].cpu() == 0).all()
    # identical clouds, identity init: already converged
    out = ops.icp_refine(src.to(dev), src.to(dev), None, 0.02, 50)
    np.testing.assert_allclose(out["T"].cpu().numpy(), np.broadcast_to(np.eye(4, dtype=np.float32), (3, 4, 4)), atol=1e-6)
    assert float(out["fitness"].min()) == 1.0 and float(out["inlier_rmse"].max()) < 1e-6


def test_houv_plus_icp_improves_or_keeps_alignment(dev):
    """cfg4 shape: HOUV answer -> ICP refine.  On pairs HOUV already solves, ICP must not make them worse."""
    from houv_amd import synthetic
    from houv_amd.icp import solve_model_icp
    from houv_amd.models.houv import HOUV, solve_model
    src, tgt, pose = synthetic.make_pairs(8, 512, seed=123)
    s, t, p = src.to(dev), tgt.to(dev), pose.to(dev)
    r0, t0, _ = solve_model(HOUV(8 * 32, 0), s, t, p, kernel=32, num_epochs=100)
    r1, t1, T = solve_model_icp(HOUV(8 * 32, 0), s, t, p, kernel=32, num_epochs=100)
    good = r0 < 3.0
    assert bool(good.any())
    assert bool((r1[good] <= r0[good] + 1.0).all())
    assert T.shape == (8, 4, 4) and bool((T[:, 3, 3] == 1).all())
