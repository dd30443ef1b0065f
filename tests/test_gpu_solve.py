"""GPU parity of the fused HOUV loop (houv_solve_iterate) and the host mirrors around it, as a ladder
(SURVEY.md section 7 "chaotic trajectories"): per-op -> single step -> short horizon -> end-to-end."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from oracle import houv_ref_cpu as orc  # noqa: E402
from solve_cases import ORACLE_CASES, PRUNED_CASES, oracle_batch  # noqa: E402

T = torch.tensor


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch.device("cuda:0")


def _oracle_terms(src, tgt, params, base, mode="houv"):
    """Loss terms + parameter grads of ONE forward/backward of the oracle, for n hypotheses (K=1 each)."""
    tv = [T(params[:, 0:3].astype(np.float32), requires_grad=True), T(params[:, 3:4].astype(np.float32), requires_grad=True),
          T(params[:, 4:7].astype(np.float32), requires_grad=True), T(params[:, 7:8].astype(np.float32), requires_grad=True)]
    moved, R, Tt = orc.houv_forward(src, *tv, base, mode)
    cds = [orc.calc_cd_percent(moved, tgt, percent=0.5)]
    if mode == "houv":
        cds += [orc.loss_view(moved, tgt, dim=d) for d in range(3)]
    if mode == "houv":
        loss, min1 = orc.predict_loss(moved, tgt)
    else:
        min1 = torch.minimum(*cds[0]); loss = min1 * 6
    loss.mean().backward()
    cd = np.stack([np.stack([c[0].detach().numpy(), c[1].detach().numpy()], 1) for c in cds], 1)   # [n,4|1,2]
    grads = np.concatenate([t.grad.numpy() for t in tv], 1)
    return dict(cd=cd.reshape(len(params), -1), loss=loss.detach().numpy(), min1=min1.detach().numpy(), grads=grads,
                R=R.detach().numpy(), T=Tt.detach().numpy()[:, 0])


@pytest.mark.parametrize("N,M,base,mode", ORACLE_CASES)
def test_single_forward_backward_vs_oracle(dev, N, M, base, mode):
    """Per-op rung: the 8 Chamfer terms (1e-5, north_star's Chamfer bar), loss, min_1, R/T and the parameter
    gradient (1e-4 relative to the largest component) of one forward from identical parameters.  ORACLE_CASES holds a
    view-term and a no-view size for EVERY solve_kernel<BLOCK,Q> variant (tests/test_host_logic.py enforces that),
    including <512,4> at 2048x2048 -- the kernel bench.py times."""
    from houv_amd import _lib, ops, synthetic
    P = oracle_batch(N, M)          # the float64 [P,N,M] oracle temp is 134 MB per hypothesis at 4096^2
    assert _lib.solve_variant(N, M) is not None
    src, tgt, _ = synthetic.make_pairs(P, max(N, M), seed=77)
    src, tgt = src[:, :N].contiguous(), tgt[:, :M].contiguous()
    rng = np.random.default_rng(N + base)
    params = rng.standard_normal((P, 8))
    params = params.astype(np.float32).astype(np.float64)
    want = _oracle_terms(src, tgt, params, base, mode)
    state = torch.zeros((P, 24), dtype=torch.float64, device=dev)
    state[:, :8] = T(params).to(dev)
    out = ops.solve_iterate(src.to(dev), tgt.to(dev), state, 1, steps_done=0, n_iters=1, angle_base=base,
                            trans_mode=0 if mode == "houv" else 1, use_views=(mode == "houv"), f64_params=(mode != "houv"),
                            k_full=int(N * 0.5), k_view=N, lr=0.01, loss_scale=1.0 / P, want_grad=True, want_cd=True)
    ncd = 8 if mode == "houv" else 2
    np.testing.assert_allclose(out["cd"].cpu().numpy()[:, :ncd], want["cd"][:, :ncd], rtol=0, atol=1e-5)
    np.testing.assert_allclose(out["loss"].cpu().numpy(), want["loss"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(out["score"].cpu().numpy(), want["min1"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(out["R"].cpu().numpy(), want["R"], atol=1e-6)
    np.testing.assert_allclose(out["T"].cpu().numpy(), want["T"], atol=1e-6)
    g = out["grad"].cpu().numpy()
    scale = np.abs(want["grads"]).max(axis=1, keepdims=True)
    err = (np.abs(g - want["grads"]) / scale).max(axis=1)
    # The oracle's float64 search is exact; the fp32 direct-difference search (ours, and the reference's CUDA kernel)
    # may take the other point of a pair tied to within fp32 rounding (relative margin ~1e-7).  One such flip moves
    # a hypothesis' gradient by one point's share, <= ~2/N, and leaves the loss unchanged: allow it on at most one
    # hypothesis of the 30; everything else must agree to 2e-4 of the gradient's largest component.
    assert (err > 2e-4).sum() <= (1 if max(N, M) <= 1100 else 2), err
    assert err.max() < 3.0 / min(N, M), err


def test_golden_single_step_and_trajectory(golden, dev, solver_mode):
    """G5: gradients of step 1, parameters after 1,2,5 steps (1e-5) and the 20-step horizon (loss 1e-5 / R,T 1e-4)
    against the reference's own numbers (B=2, N=256, K=16)."""
    from houv_amd import solver
    from houv_amd.models.houv import HOUV, predict_model
    g = golden("g5_trajectory.npz")
    s, t = T(g["src"]).to(dev), T(g["tgt"]).to(dev)
    for base in (0, 2):
        p0 = solver.houv_init_params(32, 2021)
        out, state = solver.run_stage(s, t, p0, 16, 1, angle_base=base, trans_mode=0, use_views=True, f64_params=False,
                                      lr=0.01, want_grad=True)
        ref_g = np.concatenate([g[f"b{base}_grad_{k}"] for k in ("V", "angle", "tran_c", "tran_s")], 1)
        scale = np.abs(ref_g).max(axis=1, keepdims=True)
        np.testing.assert_allclose(out["grad"].cpu().numpy() / scale, ref_g / scale, atol=2e-4)
        for n, tol in ((1, 2e-6), (2, 1e-5), (5, 5e-5), (20, 1e-3)):
            net = HOUV(32, 0)
            m1, R, Tt = predict_model(net, s, t, kernel=16, num_epochs=n, angle_base=base)
            ref_p = np.concatenate([g[f"b{base}_n{n}_{k}"] for k in ("V", "angle", "tran_c", "tran_s")], 1)
            np.testing.assert_allclose(net.packed_params().detach().cpu().numpy(), ref_p, atol=tol)
            ht = 1e-5 if n <= 5 else 1e-4
            np.testing.assert_allclose(m1.cpu().numpy(), g[f"b{base}_n{n}_min1"], atol=1e-5 if n <= 5 else 5e-5)
            np.testing.assert_allclose(R.cpu().numpy(), g[f"b{base}_n{n}_R"], atol=ht * 10 if n == 20 else ht)
            np.testing.assert_allclose(Tt.cpu().numpy(), g[f"b{base}_n{n}_T"], atol=ht * 10 if n == 20 else ht)


def test_chunked_launches_equal_one_launch(dev):
    """State round-trips through HBM between launches: 7 iterations as 7x1, 3+4 and 1x7 must agree bit for bit."""
    from houv_amd import solver, synthetic
    src, tgt, _ = synthetic.make_pairs(2, 192, seed=3)
    src, tgt = src.to(dev), tgt.to(dev)
    p0 = solver.houv_init_params(2 * 13, 2021)
    res = []
    for chunk in (1, 3, 7):
        out, st = solver.run_stage(src, tgt, p0, 13, 7, angle_base=1, trans_mode=0, use_views=True, f64_params=False,
                                   lr=0.01, iters_per_launch=chunk)
        res.append((out["score"].cpu(), out["R"].cpu(), st.cpu()))
    for r in res[1:]:
        assert torch.equal(r[0], res[0][0]) and torch.equal(r[1], res[0][1]) and torch.equal(r[2], res[0][2])


def test_solve_twin_short_horizon(golden, dev, solver_mode):
    """G6: getPredict_angle (float64 leaves, lr 0.1, sigma = sin(s pi), loss 6*min_1), 20 iterations, harness-seeded numpy."""
    from houv_amd.train_utils import getPredict_angle
    g = golden("g6_solve.npz")
    np.random.seed(int(g["gpa_np_seed"]))
    m1, R, Tt, ts = getPredict_angle(T(g["solve_src"]).to(dev), T(g["solve_tgt"]).to(dev), kernel=4, num_epochs=20, angle_base=1)
    # lr = 0.1 is 10x the HOUV module's: 20 steps already amplify fp32 rounding visibly -> looser than G5
    np.testing.assert_allclose(m1.cpu().numpy(), g["gpa_min1"], atol=2e-3)
    np.testing.assert_allclose(R.cpu().numpy(), g["gpa_R"], atol=3e-2)
    np.testing.assert_allclose(Tt.cpu().numpy(), g["gpa_T"], atol=3e-2)


def test_solve_model_end_to_end_golden(golden, dev, solver_mode):
    """G6 end to end (3 pairs x K=16 x 30 iterations, one >=120 degree pair -> retry stage): the retry set and the
    winning transforms of the reference.  30 iterations at lr 0.01 stay within the short-horizon regime."""
    from houv_amd.models.houv import HOUV, solve_model
    g = golden("g6_solve.npz")
    s, t, pose = T(g["src"]).to(dev), T(g["tgt"]).to(dev), T(g["pose"]).to(dev)
    net = HOUV(48, 0)
    r_err, t_err, ans = solve_model(net, s, t, pose, kernel=16, num_epochs=30)
    a = ans.cpu().numpy()
    assert a.shape == (3, 4, 4) and np.all(a[:, 3, :] == 0)          # quirk: bottom row stays zero (houv.py:187-195)
    np.testing.assert_allclose(a, g["ans"], atol=5e-3)
    np.testing.assert_allclose(r_err.cpu().numpy(), g["r_err"], atol=0.3)
    np.testing.assert_allclose(t_err.cpu().numpy(), g["t_err"], atol=5e-3)
    out = solve_model(HOUV(48, 0), s, t, None, kernel=16, num_epochs=30, prefix='test')
    assert not out.is_cuda and out.shape == (3, 4, 4)                # prefix == 'test' returns a host tensor (houv.py:199-200)


def test_end_to_end_statistical_vs_oracle(dev, solver_mode):
    """Top rung: full solve_model (200 iterations, retry stage) on 4 synthetic 128-pt pairs, K=26: trajectories are
    chaotic beyond ~50 iterations, so compare the distribution: the mean final score and mean RotE must agree with
    the CPU oracle's within the oracle's own sensitivity to a 1e-7 input perturbation (x3 margin, floor 0.5 deg)."""
    from houv_amd import synthetic
    from houv_amd.models.houv import HOUV, solve_model
    from houv_amd.train_utils import rotation_error
    src, tgt, pose = synthetic.make_pairs(4, 128, seed=99)
    r_gpu, t_gpu, ans = solve_model(HOUV(4 * 26, 0), src.to(dev), tgt.to(dev), pose.to(dev), kernel=26, num_epochs=200)
    r_ref, t_ref, ans_ref = orc.solve_model(src, tgt, pose, kernel=26, num_epochs=200)
    r_pert, _, _ = orc.solve_model(src * (1 + 1e-7), tgt, pose, kernel=26, num_epochs=200)
    spread = float((r_ref - r_pert).abs().mean())
    assert abs(float(r_gpu.mean()) - float(r_ref.mean())) <= max(3 * spread, 0.5), (r_gpu, r_ref, spread)
    # pairs the oracle solves well must also be solved well on the GPU
    good = r_ref < 5
    assert bool((r_gpu.cpu()[good] < 8).all())


@pytest.mark.parametrize("fixture,frac_1e4,frac_01", [("g12_stat.npz", 0.75, 0.90), ("g13_stat512.npz", 0.30, 0.80)])
def test_end_to_end_64_pairs_vs_reference(golden, dev, fixture, frac_1e4, frac_01, solver_mode):
    """G12 / G13 (last gate; G13 is BASELINE configs[0]'s shape, 64 pairs x 512 points):
    solve_model over 64 synthetic pairs (K=26, 200 iterations, retry stages, 4 batches of 16) against the REAL
    reference's results on the same inputs (float64 Chamfer, autograd, torch.optim.Adam, CPU).  The best-of-K answer is
    stable at these sizes -- at 128 points the reference moves by < 0.2 deg per pair under a 1e-7 input perturbation
    (stored in G12) -- so the comparison is per pair, not only statistical: a share of the pairs must reproduce the
    reference's RotE / transE to north_star's 1e-4 bar (measured 58/64 at 128 points, 25/64 at 512, where fp32
    near-ties of the nearest neighbour are more frequent), most to 0.1 deg / 1e-3 (62/64, 55/64), all but a few to
    5 deg (64/64 both), and the population statistics must agree."""
    from houv_amd.models.houv import HOUV, solve_model
    g = golden(fixture)
    K, epochs, batch = int(g["kernel"]), int(g["num_epochs"]), int(g["batch"])
    src, tgt, pose = T(g["src"]).to(dev), T(g["tgt"]).to(dev), T(g["pose"]).to(dev)
    r_all, t_all = [], []
    for b in range(0, src.shape[0], batch):
        r, t, ans = solve_model(HOUV(batch * K, 0), src[b:b + batch], tgt[b:b + batch], pose[b:b + batch], kernel=K,
                                num_epochs=epochs)
        assert tuple(ans.shape) == (batch, 4, 4) and bool((ans[:, 3] == 0).all())          # houv.py:150: row 3 stays zero
        r_all.append(r.cpu().numpy()); t_all.append(t.cpu().numpy())
    r, t = np.concatenate(r_all), np.concatenate(t_all)
    dr, dt = np.abs(r - g["ref_r_err"]), np.abs(t - g["ref_t_err"])
    n = len(r)
    assert ((dr <= 1e-4 * 180 / np.pi) & (dt <= 1e-4)).sum() >= frac_1e4 * n, (np.sort(dr)[-16:], np.sort(dt)[-16:])
    assert ((dr <= 0.1) & (dt <= 1e-3)).sum() >= frac_01 * n, (np.sort(dr)[-8:], np.sort(dt)[-8:])
    assert (dr <= 5.0).sum() >= 0.95 * n, np.sort(dr)[-8:]
    if "pert_r_err" in g.files:
        assert np.abs(g["ref_r_err"] - g["pert_r_err"]).max() < 0.5      # the premise of a per-pair comparison
    assert abs(r.mean() - g["ref_r_err"].mean()) <= 1.0 and abs(np.median(r) - np.median(g["ref_r_err"])) <= 0.3
    assert abs((r < 5).mean() - (g["ref_r_err"] < 5).mean()) <= 0.05 and abs(t.mean() - g["ref_t_err"].mean()) <= 3e-3


@pytest.mark.parametrize("mode", ["houv", "solve"])
def test_large_cloud_path_matches_fused_kernel(dev, mode, monkeypatch):
    """Clouds above 4096 points do not fit the fused kernel's LDS and take the un-fused GPU path (stand-alone HIP
    Chamfer op + autograd + torch Adam).  Forced onto 600-point clouds it must track the fused kernel: same last
    forward and same parameters / Adam moments after 4 iterations; then a 4500-point problem runs end to end."""
    from houv_amd import solver, synthetic
    P, K, N = 2, 26, 600
    src, tgt, _ = synthetic.make_pairs(P, N, seed=5)
    src, tgt = src.to(dev), tgt.to(dev)
    p0 = solver.houv_init_params(P * K)
    kw = dict(angle_base=1, trans_mode=0 if mode == "houv" else 1, use_views=(mode == "houv"),
              f64_params=(mode != "houv"), lr=0.01 if mode == "houv" else 0.1, want_grad=True, want_cd=True)
    fused, st_f = solver.run_stage(src, tgt, p0, K, 4, **kw)
    monkeypatch.setattr(solver, "FUSED_MAX_POINTS", 0)
    unf, st_u = solver.run_stage(src, tgt, p0, K, 4, **kw)
    monkeypatch.setattr(solver, "FUSED_MAX_POINTS", 4096)
    tol = 2e-5 if mode == "houv" else 2e-4          # lr 0.1 amplifies fp32 rounding of the first steps
    for key in ("score", "loss", "R", "T", "cd"):
        assert torch.allclose(unf[key], fused[key], rtol=0, atol=tol * (5 if key == "loss" else 1)), key
    g_scale = fused["grad"].abs().max(dim=1, keepdim=True)[0]
    err = ((unf["grad"] - fused["grad"]).abs() / g_scale).max(dim=1)[0]
    # the two paths sum in different orders: after a few steps a hypothesis may sit on the other side of an fp32
    # near-tie of its nearest neighbour (one point's share of the gradient, <= ~3/N each)
    assert int((err > 2e-4).sum()) <= 4 and float(err.max()) < 12.0 / N, err
    assert torch.allclose(st_u[:, :8], st_f[:, :8], rtol=0, atol=20 * tol)
    assert torch.allclose(st_u[:, 8:16], st_f[:, 8:16], rtol=1e-2, atol=1e-7)
    big_s, big_t, _ = synthetic.make_pairs(1, 4500, seed=6)
    out, st = solver.run_stage(big_s.to(dev), big_t.to(dev), solver.houv_init_params(26), 26, 2, angle_base=0,
                               trans_mode=0, use_views=True, f64_params=False, lr=0.01)
    assert bool(torch.isfinite(out["score"]).all()) and tuple(out["R"].shape) == (26, 3, 3)
    eye = torch.eye(3, device=dev).expand(26, 3, 3)
    assert torch.allclose(torch.bmm(out["R"], out["R"].transpose(1, 2)), eye, atol=1e-5)


def test_topk_size_out_of_range_is_an_error(dev):
    """topk(k) with k > number of points raises in the reference (model_utils_completion.py:91): M < N with views."""
    from houv_amd import _lib, solver, synthetic
    src, tgt, _ = synthetic.make_pairs(2, 64, seed=1)
    with pytest.raises(RuntimeError):      # views on and N != M: the reference's mask broadcast fails (loss_view)
        solver.run_stage(src.to(dev), tgt[:, :40].contiguous().to(dev), solver.houv_init_params(26 * 2), 26, 1,
                         angle_base=0, trans_mode=0, use_views=True, f64_params=False, lr=0.01)
    with pytest.raises(_lib.HouvHipError):  # views off, k = N/2 = 32 > M = 20: topk out of range
        solver.run_stage(src.to(dev), tgt[:, :20].contiguous().to(dev), solver.houv_init_params(26 * 2), 26, 1,
                         angle_base=0, trans_mode=1, use_views=False, f64_params=True, lr=0.1)
    with pytest.raises(IndexError):
        solver.houv_init_params(25)                                   # houv.py:47-51 has no bounds check


def test_full_size_properties_cfg2_shape(dev):
    """BASELINE cfg2-shaped problem (2048x2048 points, K=64): properties that need no CPU oracle.
    (i) pairs are independent: permuting the pair list permutes the results BIT-exactly (same batch size, hence the
    same 1/(P*K) loss scale); (ii) the returned R are rotations and T obey the 0..0.25 magnitude window of
    houv.py:99; (iii) scores are finite and positive; (iv) more iterations do not increase the best score of most pairs."""
    from houv_amd import solver, synthetic
    P, K, N = 16, 64, 2048
    src, tgt, _ = synthetic.make_pairs(P, N, seed=404)
    src, tgt = src.to(dev), tgt.to(dev)
    p0 = solver.houv_init_params(P * K)
    kw = dict(angle_base=0, trans_mode=0, use_views=True, f64_params=False, lr=0.01)
    out, st = solver.run_stage(src, tgt, p0, K, 6, **kw)
    perm = torch.randperm(P, generator=torch.Generator().manual_seed(1)).to(dev)
    # hypothesis h of pair p keeps its own initial parameters: permute the parameter blocks along with the pairs
    p0p = torch.as_tensor(p0).reshape(P, K, 8)[perm.cpu()].reshape(P * K, 8)
    outp, stp = solver.run_stage(src[perm].contiguous(), tgt[perm].contiguous(), p0p, K, 6, **kw)
    for key in ("score", "loss", "R", "T"):
        a = out[key].reshape(P, K, -1)[perm].reshape(outp[key].shape)
        assert torch.equal(a, outp[key]), key
    assert torch.equal(st.reshape(P, K, 24)[perm].reshape(P * K, 24), stp)
    R = out["R"]
    eye = torch.eye(3, device=dev).expand_as(R)
    assert float((R @ R.transpose(1, 2) - eye).abs().max()) < 1e-5 and float((torch.det(R) - 1).abs().max()) < 1e-5
    tn = out["T"].norm(dim=1)
    assert float(tn.max()) <= 0.25 + 1e-6 and float(tn.min()) >= 0.0
    assert bool(torch.isfinite(out["score"]).all()) and float(out["score"].min()) > 0
    out2, _ = solver.run_stage(src, tgt, p0, K, 40, **kw)
    better = (out2["score"].reshape(P, K).min(1)[0] <= out["score"].reshape(P, K).min(1)[0]).float().mean()
    assert float(better) >= 0.75


def test_zero_distance_poisons_only_that_hypothesis(dev):
    """a nearest-neighbour distance of exactly 0 makes sqrt's backward NaN (model_utils_completion.py:94-95),
    which poisons that hypothesis only; NaN scores lose the best-of-K (topk sorts NaN last)."""
    from houv_amd import ops, solver, synthetic
    src, _, _ = synthetic.make_pairs(1, 64, seed=2)
    src = src.to(dev)
    K = 26
    p0 = solver.houv_init_params(K)
    p0[3, 3] = -0.5      # theta = sin(-pi/2) pi/8 + pi/8 = 0  -> R = I
    p0[3, 7] = -0.5      # sigma = sin(-pi/2)/8 + 1/8 = 0       -> T = 0: hypothesis 3 moves nothing, every d == 0
    out1, st = solver.run_stage(src, src.clone(), p0, K, 1, angle_base=0, trans_mode=0, use_views=True, f64_params=False,
                                lr=0.01, want_grad=True)
    assert float(out1["score"][3]) == 0.0 and float(out1["loss"][3]) == 0.0
    assert bool(torch.isnan(out1["grad"][3]).all()) and bool(torch.isnan(st[3, :8]).all())
    others = torch.arange(K, device=dev) != 3
    assert bool(torch.isfinite(out1["grad"][others]).all()) and bool(torch.isfinite(st[others]).all())
    out2, _ = solver.run_stage(src, src.clone(), p0, K, 2, angle_base=0, trans_mode=0, use_views=True, f64_params=False, lr=0.01)
    assert bool(torch.isnan(out2["score"][3])) and bool(torch.isfinite(out2["score"][others]).all())
    best, _ = out2["score"].reshape(1, K).topk(1, dim=1, largest=False)
    assert bool(torch.isfinite(best).all())


def test_empty_batches_are_no_ops(dev):
    from houv_amd import ops
    e = torch.zeros((0, 8, 3), device=dev)
    st = torch.zeros((0, 24), dtype=torch.float64, device=dev)
    out = ops.solve_iterate(e, e, st, 4, steps_done=0, n_iters=1, angle_base=0, trans_mode=0, use_views=True,
                            f64_params=False, k_full=4, k_view=8, lr=0.01, loss_scale=1.0)
    assert out["score"].numel() == 0
    d1 = torch.zeros((0, 8), device=dev); i1 = torch.zeros((0, 8), dtype=torch.int32, device=dev)
    assert ops.chamfer_forward(e, e, d1, d1.clone(), i1, i1.clone()) == 1
    R, t = ops.kabsch(torch.zeros((0, 3, 5), device=dev), torch.zeros((0, 3, 5), device=dev))
    assert R.shape == (0, 3, 3)


def test_unfused_loss_glue_vs_golden(golden, dev):
    """a8-a10 through the un-fused mirrors (metrics.cd autograd + torch.topk): calc_cd_percent, loss_view, Predict_loss
    values and the gradient w.r.t. the moved cloud against the reference's own numbers (G2)."""
    from houv_amd.model_utils_completion import calc_cd_percent, loss_view
    from houv_amd.models.houv import Predict_loss
    g = golden("g2_loss.npz")
    mv = T(g["moved"]).to(dev).requires_grad_(True)
    tg = T(g["target"]).to(dev)
    c = calc_cd_percent(mv, tg, percent=0.5)
    np.testing.assert_allclose(np.stack([x.detach().cpu().numpy() for x in c]), g["cd_percent"], atol=1e-6)
    for d in range(3):
        v = loss_view(mv, tg, dim=d)
        np.testing.assert_allclose(np.stack([x.detach().cpu().numpy() for x in v]), g["views"][d], atol=1e-6)
    loss, min1 = Predict_loss(mv, tg)
    loss.mean().backward()
    np.testing.assert_allclose(loss.detach().cpu().numpy(), g["loss"], atol=5e-6)
    np.testing.assert_allclose(min1.detach().cpu().numpy(), g["min_1"], atol=1e-6)
    gm = mv.grad.cpu().numpy()
    np.testing.assert_allclose(gm, g["grad_moved"], atol=1e-6 * max(1.0, np.abs(g["grad_moved"]).max() * 1e3), rtol=2e-3)


def test_solve_twin_end_to_end_vs_oracle(golden, dev, solver_mode):
    """a15 end to end (`solve`: base stage + retry stages, float64 leaves from the harness-seeded global numpy RNG,
    lr 0.1) at a shortened horizon (_iters=12; the reference hard-codes 500): same retry decisions and transforms as the
    oracle, to the looseness lr=0.1 imposes after a dozen steps."""
    from houv_amd.train_utils import solve
    g = golden("g6_solve.npz")
    s, t = T(g["solve_src"]), T(g["solve_tgt"])
    np.random.seed(123)
    ref = orc.solve(s, t, kernel=13, prefix="test", _iters=12)
    np.random.seed(123)
    mine = solve(s.to(dev), t.to(dev), kernel=13, prefix="test", _iters=12)
    assert not mine.is_cuda and mine.shape == ref.shape
    np.testing.assert_allclose(mine.numpy(), ref.numpy(), atol=2e-2)
    assert np.all(mine.numpy()[:, 3, :] == 0)


@pytest.mark.parametrize("N,M,views,f64,tm", PRUNED_CASES)
def test_pruned_search_is_bit_identical_to_brute_force(dev, N, M, views, f64, tm):
    """The pruned search (previous-NN upper bound + sub-tile bounding boxes; the default since round 3) must reproduce the brute-force
    kernel BIT FOR BIT on the same clouds: scores, losses, poses, gradients, the 8 Chamfer terms and the optimiser state,
    across chunked launches (workspace carried over) and bases -- against houv_solve_iterate itself and against the
    pruned entry point's verification mode (ws_valid=-1, pruning switched off)."""
    from houv_amd import solver, synthetic
    P, K = 3, 26
    src, tgt, _ = synthetic.make_pairs(P, max(N, M), seed=31)
    src = solver.spatial_sort(src[:, :N].contiguous().to(dev), solver.sort_leaf(N, M))
    tgt = solver.spatial_sort(tgt[:, :M].contiguous().to(dev), solver.sort_leaf(N, M))
    p0 = solver.houv_init_params(P * K) if not f64 else np.random.default_rng(1).standard_normal((P * K, 8))
    kw = dict(angle_base=1, trans_mode=tm, use_views=views, f64_params=f64, lr=0.1 if f64 else 0.01, want_grad=True,
              want_cd=True)
    ref, st_ref = solver.run_stage(src, tgt, p0, K, 23, iters_per_launch=50, pruned=False, **kw)
    ver, st_ver = solver.run_stage(src, tgt, p0, K, 23, iters_per_launch=50, pruned="verify", **kw)
    for key in ("score", "loss", "R", "T", "grad", "cd"):
        assert torch.equal(ver[key], ref[key]), key
    assert torch.equal(st_ver, st_ref)
    for chunk in (50, 7):
        # run_stage re-sorts sorted clouds (a no-op permutation), so both modes see identical point orders
        out, st = solver.run_stage(src, tgt, p0, K, 23, iters_per_launch=chunk, pruned=True, **kw)
        for key in ("score", "loss", "R", "T", "grad", "cd"):
            assert torch.equal(out[key], ref[key]), (key, chunk)
        assert torch.equal(st, st_ref), chunk


@pytest.mark.parametrize("N,views,f64,tm,pruned", [(2048, True, False, 0, False), (600, True, False, 0, False),
                                                    (1000, False, True, 1, False), (300, True, False, 0, True)])
def test_gradient_direction_prediction_does_not_change_results(dev, monkeypatch, N, views, f64, tm, pruned):
    """Since round 2 the exact-NN rescans and the G / GP sums run only for the direction that wins each metric's min: A's
    for the metrics A won in the PREVIOUS iteration (its sweep state is gone when the winner is known), B's for the
    metrics B wins, and a metric predicted B but won by A is repaired by redoing A for that one metric.  None of this may
    change a bit: the normal run, a run that always predicts B (every A-win takes the repair path) and a run that rescans
    everything (houv_debug_set("solve_predict", ...), a diagnostic switch of the library) must agree exactly -- scores, losses, poses,
    gradients, the 8 Chamfer terms and the optimiser state after 40 iterations in chunks of 50 and of 7."""
    from houv_amd import solver, synthetic
    P, K = 3, 26
    src, tgt, _ = synthetic.make_pairs(P, N, seed=41)
    src, tgt = solver.spatial_sort(src.to(dev)), solver.spatial_sort(tgt.to(dev))
    p0 = solver.houv_init_params(P * K) if not f64 else np.random.default_rng(2).standard_normal((P * K, 8))
    kw = dict(angle_base=0, trans_mode=tm, use_views=views, f64_params=f64, lr=0.1 if f64 else 0.01, want_grad=True,
              want_cd=True, pruned=pruned)
    runs = {}
    from houv_amd import _lib
    try:
        for mode, chunk in ((None, 50), ("b", 50), ("all", 50), (None, 7), ("b", 7)):
            _lib.debug_set("solve_predict", {None: 0, "b": 1, "all": 2}[mode])
            runs[(mode, chunk)] = solver.run_stage(src, tgt, p0, K, 40, iters_per_launch=chunk, **kw)
    finally:
        _lib.debug_set("solve_predict", 0)
    ref, st_ref = runs[(None, 50)]
    for key, (out, st) in runs.items():
        for name in ("score", "loss", "R", "T", "grad", "cd"):
            assert torch.equal(out[name], ref[name]), (key, name)
        assert torch.equal(st, st_ref), key
    # both directions do win somewhere in this workload, i.e. the three modes really exercised different code
    picked_a = (ref["cd"][:, 0::2] > ref["cd"][:, 1::2]) if views else (ref["cd"][:, 0:1] > ref["cd"][:, 1:2])
    assert bool(picked_a.any()) and bool((~picked_a).any())


def test_solve_twin_end_to_end_32_pairs_vs_reference(golden, dev, solver_mode):
    """G14: the REAL reference's ``train_utils.solve`` (test.py:64's path: 500 iterations, lr 0.1, float64 leaves from the
    harness-seeded global numpy RNG, retry stages) on 32 synthetic 128-pt pairs, K=26.  At lr 0.1 the trajectories are
    chaotic (registration/README.md:82-91 calls the results non-reproducible), so per-pair answers agree only loosely
    (measured: 24/32 within 5 deg) and the gate is statistical: mean / median RotE, share solved to < 5 deg, mean transE
    (measured 26.53 / 4.75 / 53.1 % / 0.085 against the reference's 26.41 / 4.32 / 53.1 % / 0.092).  The sample is ONE draw
    of a chaotic system: a kernel that groups its sums differently (tried in round 3: one wave with two points per lane) lands
    32 other trajectories -- mean RotE 19.0, mean transE 0.063, i.e. outside these bars although every per-op rung holds."""
    from houv_amd.train_utils import rotation_error, solve, translation_error
    g = golden("g14_twin.npz")
    K, batch, seed0 = int(g["kernel"]), int(g["batch"]), int(g["seed0"])
    src, tgt, pose = T(g["src"]), T(g["tgt"]), T(g["pose"])
    outs = []
    for i, b in enumerate(range(0, src.shape[0], batch)):
        np.random.seed(seed0 + i)
        out = solve(src[b:b + batch].to(dev), tgt[b:b + batch].to(dev), kernel=K, prefix="test")
        assert not out.is_cuda and tuple(out.shape) == (batch, 4, 4) and bool((out[:, 3] == 0).all())
        outs.append(out)
    mine, ref = torch.cat(outs), T(g["ans"])
    r_m = rotation_error(mine[:, :3, :3], pose[:, :3, :3]).numpy()
    r_r = rotation_error(ref[:, :3, :3], pose[:, :3, :3]).numpy()
    t_m = translation_error(mine[:, :3, 3], pose[:, :3, 3]).numpy()
    t_r = translation_error(ref[:, :3, 3], pose[:, :3, 3]).numpy()
    assert (np.abs(r_m - r_r) <= 5.0).mean() >= 0.6
    assert abs(r_m.mean() - r_r.mean()) <= 3.0 and abs(np.median(r_m) - np.median(r_r)) <= 1.5
    assert abs((r_m < 5).mean() - (r_r < 5).mean()) <= 0.1 and abs(t_m.mean() - t_r.mean()) <= 0.015


def test_concurrent_retry_stages_equal_sequential_ones(dev, monkeypatch):
    """The three retry stages of solve_model / solve are independent solves of the same retried pairs; launched on three side
    streams (solver.CONCURRENT_RETRIES) they must give exactly what the sequential order gives, and `net` must end up with the
    last stage's parameters either way (houv.py:168-180)."""
    from houv_amd import solver, synthetic
    from houv_amd.models.houv import HOUV, solve_model
    from houv_amd.train_utils import solve
    src, tgt, pose = synthetic.make_pairs(12, 256, seed=99)          # several pairs beyond the 0.030 threshold after 25 iterations
    src, tgt, pose = src.to(dev), tgt.to(dev), pose.to(dev)
    outs = {}
    for conc in (True, False):
        monkeypatch.setattr(solver, "CONCURRENT_RETRIES", conc)
        net = HOUV(12 * 26, 0)
        r, t, ans = solve_model(net, src, tgt, pose, kernel=26, num_epochs=25)
        np.random.seed(5)
        twin = solve(src, tgt, kernel=26, prefix='test', _iters=12)
        outs[conc] = (ans.clone(), net.packed_params().detach().clone(), twin.clone())
    monkeypatch.setattr(solver, "CONCURRENT_RETRIES", True)
    for a, b in zip(outs[True], outs[False]):
        assert torch.equal(a, b)
    _, score, retry = solver.best_of_k_with_retry(
        lambda s, t, base: __import__("houv_amd.models.houv", fromlist=["x"]).predict_model(HOUV.blank_like(HOUV(26, 0).to(dev)), s, t, kernel=26, num_epochs=25, angle_base=base), src, tgt)
    assert 0 < retry.numel() < 12            # the workload really has a retry stage


def test_fused_solve_through_torch_custom_ops(golden, dev):
    """SURVEY 8(b) row 2 / VERDICT r2 #6: the fused loop is registered as PyTorch-ROCm custom ops with mutable-argument
    schemas.  Through ``torch.ops.houv.solve_iterate`` and ``solve_iterate_pruned`` (caller-allocated outputs, state and
    workspace updated in place, returns 1 like the C call) the 5-step G5 trajectory of the reference is reproduced, both
    searches agree bit for bit on the same clouds, and ``pose_forward`` / ``icp_refine`` answer like their Python wrappers."""
    from houv_amd import ops, solver
    ops.register_torch_ops()
    g = golden("g5_trajectory.npz")
    s, t = solver.spatial_sort(T(g["src"]).to(dev)), solver.spatial_sort(T(g["tgt"]).to(dev))
    P, N, K, n = s.shape[0], s.shape[1], 16, 32
    res = {}
    for pruned in (False, True):
        state = torch.zeros((n, 24), dtype=torch.float64, device=dev)
        state[:, :8] = T(solver.houv_init_params(n, 2021)).to(dev)
        score, loss = torch.empty(n, device=dev), torch.empty(n, device=dev)
        R, Tt = torch.empty((n, 3, 3), device=dev), torch.empty((n, 3), device=dev)
        grad = torch.empty((n, 8), device=dev)
        args = (s, t, state, K, 0, 5, 2, 0, True, False, N // 2, N, 0.01, 0.9, 0.999, 1e-8, 1.0 / n, score, loss, R, Tt, grad, None)
        if pruned:
            ws = ops.solve_workspace(n, N, N, dev)
            assert torch.ops.houv.solve_iterate_pruned(*args, ws, 0) == 1
        else:
            assert torch.ops.houv.solve_iterate(*args) == 1
        res[pruned] = (score.clone(), R.clone(), Tt.clone(), state.clone(), grad.clone())
    for a, b in zip(res[False], res[True]):
        assert torch.equal(a, b)
    score, R, Tt, state, _ = res[True]
    ref_p = np.concatenate([g[f"b2_n5_{k}"] for k in ("V", "angle", "tran_c", "tran_s")], 1)
    np.testing.assert_allclose(state[:, :8].cpu().numpy(), ref_p, atol=5e-5)          # parameters after 5 Adam steps (G5's bar)
    np.testing.assert_allclose(score.cpu().numpy(), g["b2_n5_min1"].reshape(-1), atol=1e-5)
    np.testing.assert_allclose(R.cpu().numpy(), g["b2_n5_R"].reshape(-1, 3, 3), atol=1e-5)
    # pose_forward / icp_refine
    prm = state[:, :8].float().contiguous()
    R2, T2, moved = torch.ops.houv.pose_forward(prm, 2, 0, None)
    want = ops.pose_forward(prm, 2, 0)
    assert torch.equal(R2, want[0]) and torch.equal(T2, want[1]) and moved.numel() == 0
    src_k = s.repeat_interleave(K, dim=0).contiguous()
    R3, T3, moved = torch.ops.houv.pose_forward(prm, 2, 0, src_k)
    assert torch.allclose(moved, torch.bmm(src_k, R3.transpose(1, 2)) + T3.unsqueeze(1), atol=1e-6)
    Ti, fit, rmse, its = torch.ops.houv.icp_refine(s, t, None, 0.02, 30, 1e-6, 1e-6)
    w = ops.icp_refine(s, t, None, 0.02, 30, 1e-6, 1e-6)
    assert torch.equal(Ti, w["T"]) and torch.equal(fit, w["fitness"]) and torch.equal(its, w["iterations"])


def test_pruned_search_soak_over_edge_and_random_sizes():
    """scripts/soak_pruned.py in a child process: every variant boundary (257, 512/513, ... 2048/2049, 4095/4096), ragged N != M,
    chunked launches and all four angle bases; the pruned kernel's state and outputs must equal the brute-force kernel's bit for bit."""
    import subprocess
    env = dict(os.environ, CASES="64", SEED="11")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "soak_pruned.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "64 cases, 0 mismatches" in r.stdout


def test_calc_cd_f1_branch(dev):
    """calc_cd / calc_cd_percent with calc_f1=True (model_utils_completion.py:69-100, fscore.py:3-16): the harmonic mean of the
    fractions of squared NN distances below 1e-4, per pair; 0 when no point is that close."""
    from houv_amd import model_utils_completion as muc
    gen = torch.Generator().manual_seed(5)
    a = torch.rand(3, 300, 3, generator=gen)
    b = a + 0.004 * torch.randn(3, 300, 3, generator=gen)        # NN distances straddle the 1e-2 threshold on the distance
    b[2] += 5.0                                                  # a pair with nothing close: F-score 0, not NaN
    cd_p, cd_t, f1 = muc.calc_cd(a.to(dev), b.to(dev), calc_f1=True)
    d = ((b.double().unsqueeze(2) - a.double().unsqueeze(1)) ** 2).sum(-1)     # [B, gt=b, out=a]: cd()(gt, output) = (b -> a, a -> b)
    d1, d2 = d.min(2)[0], d.min(1)[0]
    p1, p2 = (d1 < 1e-4).double().mean(1), (d2 < 1e-4).double().mean(1)
    want = torch.where(p1 + p2 > 0, 2 * p1 * p2 / (p1 + p2).clamp_min(1e-300), torch.zeros_like(p1))
    np.testing.assert_allclose(f1.cpu().numpy(), want.numpy(), atol=1e-6)
    assert float(f1[2]) == 0.0 and 0.0 < float(f1[0]) < 1.0
    np.testing.assert_allclose(cd_t.cpu().numpy(), (d1.mean(1) + d2.mean(1)).numpy(), rtol=1e-5)
    pp, pt, pf = muc.calc_cd_percent(a.to(dev), b.to(dev), calc_f1=True, percent=0.5)
    k1, k2 = d1.topk(150, dim=1, largest=False)[0], d2.topk(150, dim=1, largest=False)[0]
    q1, q2 = (k1 < 1e-4).double().mean(1), (k2 < 1e-4).double().mean(1)
    np.testing.assert_allclose(pf.cpu().numpy(), torch.where(q1 + q2 > 0, 2 * q1 * q2 / (q1 + q2).clamp_min(1e-300), torch.zeros_like(q1)).numpy(), atol=1e-6)
