"""GPU parity of the Chamfer op (houv_chamfer_forward/backward through metrics.cd) against the oracle and the
golden vectors captured from the reference.  Calls go through the C ABI (ctypes)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import houv_ref_cpu as orc  # noqa: E402

T = torch.tensor


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch.device("cuda:0")


def _cd(a, b, dev):
    from houv_amd.metrics import cd
    return cd()(a.to(dev), b.to(dev))


def _assert_idx(i_gpu, i_ref, d_gpu, pts_q, pts_r):
    """idx must be exact; where it is not, the two candidates must be tied to fp32 rounding (float64 check)."""
    i_gpu = i_gpu.cpu().long().numpy(); i_ref = i_ref.long().numpy()
    bad = np.argwhere(i_gpu != i_ref)
    for b, i in bad:
        q = pts_q[b, i].double(); r1 = pts_r[b, i_gpu[b, i]].double(); r2 = pts_r[b, i_ref[b, i]].double()
        d1 = float(((q - r1) ** 2).sum()); d2 = float(((q - r2) ** 2).sum())
        assert abs(d1 - d2) <= 2e-7 * max(d1, d2), f"idx mismatch not a tie: {d1} vs {d2}"
    return len(bad)


def test_golden_unit_test_shapes(golden, dev):
    """The reference's own test case and bar (utils/metrics/CD/unit_test.py:15-33): idx exactly equal,
    mean squared dist difference < 1e-8 -- on the reference's outputs for the same inputs."""
    g = golden("g1_chamfer.npz")
    d1, d2, i1, i2 = _cd(T(g["p1"]), T(g["p2"]), dev)
    assert np.array_equal(i1.cpu().numpy(), g["idx1"]) and np.array_equal(i2.cpu().numpy(), g["idx2"])
    assert ((d1.cpu().numpy() - g["dist1"]) ** 2).mean() + ((d2.cpu().numpy() - g["dist2"]) ** 2).mean() < 1e-8
    np.testing.assert_allclose(d1.cpu().numpy(), g["dist1"], rtol=2e-5, atol=1e-7)   # fp32 direct vs f64 expanded form
    np.testing.assert_allclose(d2.cpu().numpy(), g["dist2"], rtol=2e-5, atol=1e-7)
    assert i1.dtype == torch.int32 and d1.dtype == torch.float32


def test_golden_2048(golden, dev):
    g = golden("g1_chamfer.npz")
    a, b = T(g["big_a"]), T(g["big_b"])
    d1, d2, i1, i2 = _cd(a, b, dev)
    n_bad = _assert_idx(i1, T(g["big_idx1"]), d1, a, b) + _assert_idx(i2, T(g["big_idx2"]), d2, b, a)
    assert n_bad <= 2
    np.testing.assert_allclose(d1.cpu().numpy(), g["big_dist1"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(d2.cpu().numpy(), g["big_dist2"], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("B,N,M", [(1, 1, 1), (3, 1, 70), (2, 33, 31), (2, 257, 64), (1, 300, 2049),
                                   (2, 1000, 2000), (1, 4100, 513), (5, 512, 512)])
def test_ragged_shapes_vs_oracle(dev, B, N, M):
    gen = torch.Generator().manual_seed(B * 100003 + N * 17 + M)
    a = torch.rand(B, N, 3, generator=gen) - 0.5
    b = torch.rand(B, M, 3, generator=gen) - 0.5
    d1, d2, i1, i2 = _cd(a, b, dev)
    o1, o2, j1, j2 = orc.chamfer_nn_chunked(a, b, chunk=1)
    assert _assert_idx(i1, j1, d1, a, b) + _assert_idx(i2, j2, d2, b, a) <= 1
    np.testing.assert_allclose(d1.cpu().numpy(), o1.numpy(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(d2.cpu().numpy(), o2.numpy(), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("B,N,M", [(4, 2048, 2048), (3, 777, 1500), (2, 2500, 300), (16, 512, 512),
                                   (2, 900, 1000), (2, 1024, 700)])      # the last two: chamfer_nn_filter_kernel<4>
def test_bit_exact_vs_c_oracle(dev, B, N, M):
    """Integer/index bar: against oracle/chamfer_ref.c (fp32 direct differences, the CUDA kernel's arithmetic with the
    same fma contraction) distances must be BIT-identical and indices exactly equal, ties included."""
    from oracle import c_oracle
    gen = torch.Generator().manual_seed(N * 7 + M)
    a = torch.rand(B, N, 3, generator=gen) - 0.5
    b = torch.rand(B, M, 3, generator=gen) - 0.5
    b[:, M // 2] = b[:, 0]                       # exact duplicates -> genuine ties
    d1, d2, i1, i2 = _cd(a, b, dev)
    o1, o2, j1, j2 = c_oracle.chamfer_forward(a.numpy(), b.numpy())
    assert np.array_equal(d1.cpu().numpy().view(np.uint32), o1.view(np.uint32))
    assert np.array_equal(d2.cpu().numpy().view(np.uint32), o2.view(np.uint32))
    assert np.array_equal(i1.cpu().numpy(), j1) and np.array_equal(i2.cpu().numpy(), j2)


@pytest.mark.parametrize("B,N,M,offset,dups", [(3, 2048, 2048, 3.0, False), (2, 1500, 2048, 50.0, False),
                                                (2, 2048, 2048, 0.0, True), (2, 900, 2600, 7.0, False),
                                                (1, 2600, 5000, 0.0, True), (2, 2048, 2048, 1e4, False),
                                                (2, 1000, 900, 3.0, False), (2, 700, 1024, 0.0, True)])    # Q = 4 loop
def test_filter_uncertainty_paths_bit_exact(dev, B, N, M, offset, dups):
    """The expanded-form filter of chamfer_nn_filter_kernel only SELECTS candidate sub-tiles; whatever it is unsure about
    must be re-decided by the exact arithmetic.  Clouds far from the origin blow the uncertainty tau = 25 u (R+|q|)^2 up
    (offset 3: the second sub-tile is re-evaluated for most queries; 50 / 1e4: every query takes the exact full scan),
    and a reference cloud made of repeated blocks puts exact ties into EVERY sub-tile (lowest index must win) -- single
    LDS pass (M <= 2048, LDS-resident recovery) and multi-pass (recovery from global memory).  Bit-exact vs chamfer_ref.c."""
    from oracle import c_oracle
    gen = torch.Generator().manual_seed(int(N * 31 + M + offset))
    a = torch.rand(B, N, 3, generator=gen) - 0.5 + offset
    b = torch.rand(B, M, 3, generator=gen) - 0.5 + offset
    if dups:
        blk = b[:, :96].clone()                      # 96 points repeated: copies land in every 32-reference sub-tile
        b = blk.repeat(1, M // 96 + 1, 1)[:, :M].contiguous()
        a[:, ::3] = blk[:, torch.arange(0, N, 3)[: a[:, ::3].shape[1]] % 96]      # a third of the queries sit ON references
    d1, d2, i1, i2 = _cd(a, b, dev)
    o1, o2, j1, j2 = c_oracle.chamfer_forward(a.numpy(), b.numpy())
    assert np.array_equal(i1.cpu().numpy(), j1) and np.array_equal(i2.cpu().numpy(), j2)
    assert np.array_equal(d1.cpu().numpy().view(np.uint32), o1.view(np.uint32))
    assert np.array_equal(d2.cpu().numpy().view(np.uint32), o2.view(np.uint32))


@pytest.mark.parametrize("N,M", [(300, 500), (1500, 2300)])     # compiled sub-tile loop (Q = 2) / the assembly loop (Q = 8, two LDS passes)
def test_filter_and_direct_kernels_agree_on_nonfinite_inputs(dev, N, M):
    """NaN / Inf coordinates: such references never win in either kernel (v_min3 drops NaN); a query with no finite
    distance reports reference 0 (chamfer3D.cu:37).  The filtered kernel must behave exactly like the direct sweep it
    replaces -- compared through a second process-free route: the same inputs with the bad points removed."""
    gen = torch.Generator().manual_seed(11)
    a = torch.rand(2, N, 3, generator=gen)
    b = torch.rand(2, M, 3, generator=gen)
    bad = b.clone()
    bad[:, 7] = float("nan"); bad[:, 130, 1] = float("inf"); bad[:, M - 1] = float("-inf")
    d1, _, i1, _ = _cd(a, bad, dev)
    keep = torch.tensor([j for j in range(M) if j not in (7, 130, M - 1)])
    e1, _, k1, _ = _cd(a, b[:, keep].contiguous(), dev)
    assert torch.equal(d1, e1) and torch.equal(keep.to(dev)[k1.long()], i1.long())
    allbad = torch.full((1, 40, 3), float("nan"))
    d, _, i, _ = _cd(a[:1], allbad, dev)
    assert bool(torch.isnan(d).all()) and int(i.abs().max()) == 0


@pytest.mark.parametrize("N,M", [(300, 500), (900, 1000), (1500, 2300)])     # Q = 2 / Q = 4 / Q = 8 with two LDS passes
def test_huge_finite_coordinates_bit_exact(dev, N, M):
    """beyond ~1.3e19 the filter's |r|^2 overflows (e = -inf + inf = NaN for every reference) while the direct
    differences of chamfer3D.cu:31-36 are still finite for nearby points.  Such queries take the exact full scan: distances
    and indices bit-identical to the C oracle, which finds d = 0 for a query sitting on a reference and 2^88-sized minima
    elsewhere.  (2^44 is the grid on which fp32 numbers near 1e20 live.)"""
    from oracle import c_oracle
    gen = torch.Generator().manual_seed(N + M)
    step = float(2 ** 44)
    b = 1e20 + step * torch.randint(0, 40, (2, M, 3), generator=gen).float()
    a = 1e20 + step * torch.randint(0, 40, (2, N, 3), generator=gen).float()
    a[:, ::5] = b[:, torch.arange(0, N, 5) % M]                      # a fifth of the queries coincide with references
    a[1, 3] = 3e38                                                     # every difference^2 overflows: reference 0, d = inf
    d1, d2, i1, i2 = _cd(a, b, dev)
    o1, o2, j1, j2 = c_oracle.chamfer_forward(a.numpy(), b.numpy())
    assert np.array_equal(i1.cpu().numpy(), j1) and np.array_equal(i2.cpu().numpy(), j2)
    assert np.array_equal(d1.cpu().numpy().view(np.uint32), o1.view(np.uint32))
    assert np.array_equal(d2.cpu().numpy().view(np.uint32), o2.view(np.uint32))
    assert float(d1[0, 0]) == 0.0 and bool(torch.isinf(d1[1, 3])) and int(i1[1, 3]) == 0


def test_ties_pick_lowest_index(dev):
    """Duplicated reference points: the lowest index must win, across sub-tile (32) and LDS-tile (2048) borders."""
    gen = torch.Generator().manual_seed(5)
    base = torch.rand(1, 40, 3, generator=gen)
    b = base.repeat(1, 60, 1)                     # 2400 refs: every point appears 60 times, copies 40 apart
    a = base.clone()
    d1, _, i1, _ = _cd(a, b, dev)
    assert torch.equal(i1.cpu()[0].long(), torch.arange(40))
    assert float(d1.abs().max()) == 0.0


def test_backward_golden_and_closed_form(golden, dev):
    g = golden("g1_chamfer.npz")
    from houv_amd.metrics import cd
    p1 = T(g["p1"]).to(dev)
    p2 = T(g["p2"]).to(dev).requires_grad_(True)
    d1, d2, _, _ = cd()(p1, p2)
    torch.sum(d1).backward()                      # the reference's own loss (unit_test.py:19-20)
    np.testing.assert_allclose(p2.grad.cpu().numpy(), g["grad_p2"], rtol=1e-4, atol=2e-6)
    q1 = T(g["q1"]).to(dev).requires_grad_(True)
    q2 = T(g["q2"]).to(dev).requires_grad_(True)
    e1, e2, j1, j2 = cd()(q1, q2)
    assert np.array_equal(j1.cpu().numpy(), g["j1"]) and np.array_equal(j2.cpu().numpy(), g["j2"])
    ((e1 * T(g["w1"]).to(dev)).sum() + (e2 * T(g["w2"]).to(dev)).sum()).backward()
    np.testing.assert_allclose(q1.grad.cpu().numpy(), g["grad_q1"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(q2.grad.cpu().numpy(), g["grad_q2"], rtol=1e-4, atol=2e-6)


def test_pybind_style_module_contract(dev):
    """chamfer_3D.forward/backward keep the reference's positional signature, in-place outputs and return 1
    (chamfer_cuda.cpp:17-33); backward accumulates into pre-zeroed buffers."""
    from houv_amd.metrics import chamfer_3D
    a = torch.rand(2, 50, 3, device=dev); b = torch.rand(2, 60, 3, device=dev)
    d1 = torch.zeros(2, 50, device=dev); d2 = torch.zeros(2, 60, device=dev)
    i1 = torch.zeros(2, 50, dtype=torch.int32, device=dev); i2 = torch.zeros(2, 60, dtype=torch.int32, device=dev)
    assert chamfer_3D.forward(a, b, d1, d2, i1, i2) == 1
    o1, o2, j1, j2 = orc.chamfer_nn(a.cpu(), b.cpu())
    assert torch.equal(i1.cpu(), j1) and torch.equal(i2.cpu(), j2)
    g1 = torch.zeros_like(a); g2 = torch.zeros_like(b)
    gd1 = torch.ones(2, 50, device=dev); gd2 = torch.zeros(2, 60, device=dev)
    assert chamfer_3D.backward(a, b, g1, g2, gd1, gd2, i1, i2) == 1
    first = g1.clone()
    assert chamfer_3D.backward(a, b, g1, g2, gd1, gd2, i1, i2) == 1
    np.testing.assert_allclose(g1.cpu().numpy(), 2 * first.cpu().numpy(), rtol=1e-6)     # accumulation


def test_errors_are_loud(dev):
    from houv_amd import _lib
    from houv_amd.metrics import cd
    with pytest.raises(_lib.HouvHipError):
        cd()(torch.rand(1, 4, 3), torch.rand(1, 4, 3))            # CPU tensors: no fallback
    with pytest.raises(_lib.HouvHipError):
        cd()(torch.rand(1, 0, 3, device=dev), torch.rand(1, 4, 3, device=dev))


def test_full_size_properties(dev):
    """BASELINE cfg2-sized batch slice (2048x2048): size-independent properties instead of a CPU oracle:
    (i) dist equals the distance to the reported index; (ii) swapping the clouds swaps the outputs;
    (iii) a cloud against itself gives dist 0 / idx identity; (iv) rigid motion invariance of idx."""
    gen = torch.Generator().manual_seed(11)
    B = 64
    a = (torch.rand(B, 2048, 3, generator=gen) - 0.5).to(dev)
    b = (torch.rand(B, 2048, 3, generator=gen) - 0.5).to(dev)
    from houv_amd.metrics import cd
    d1, d2, i1, i2 = cd()(a, b)
    nn = torch.gather(b, 1, i1.long().unsqueeze(2).expand(-1, -1, 3))
    np.testing.assert_allclose(((a - nn) ** 2).sum(2).cpu().numpy(), d1.cpu().numpy(), rtol=1e-5, atol=1e-9)
    # brute-force check of a random subset of rows on the GPU with torch (float64)
    sub = torch.randint(0, 2048, (64,), generator=gen).to(dev)
    dm = ((a[:, sub].double().unsqueeze(2) - b.double().unsqueeze(1)) ** 2).sum(3)
    np.testing.assert_allclose(dm.min(2)[0].float().cpu().numpy(), d1[:, sub].cpu().numpy(), rtol=1e-5, atol=1e-9)
    e2, e1, j2, j1 = cd()(b, a)
    assert torch.equal(e1, d1) and torch.equal(j1, i1) and torch.equal(e2, d2) and torch.equal(j2, i2)
    s1, s2, k1, k2 = cd()(a, a)
    assert float(s1.max()) == 0.0 and torch.equal(k1.cpu().long(), torch.arange(2048).expand(B, -1))


def test_torch_ops_registration(dev):
    """The ops are also reachable as PyTorch-ROCm custom ops: torch.ops.houv.* (in-place outputs declared in the schema)."""
    from houv_amd import ops
    ops.register_torch_ops()
    a = torch.rand(2, 40, 3, device=dev); b = torch.rand(2, 30, 3, device=dev)
    d1 = torch.empty(2, 40, device=dev); d2 = torch.empty(2, 30, device=dev)
    i1 = torch.empty(2, 40, dtype=torch.int32, device=dev); i2 = torch.empty(2, 30, dtype=torch.int32, device=dev)
    assert torch.ops.houv.chamfer_forward(a, b, d1, d2, i1, i2) == 1
    o1, o2, j1, j2 = orc.chamfer_nn(a.cpu(), b.cpu())
    assert torch.equal(i1.cpu(), j1) and torch.equal(i2.cpu(), j2)
    R, t = torch.ops.houv.kabsch(torch.randn(2, 3, 50, device=dev), torch.randn(2, 3, 50, device=dev), None)
    assert R.shape == (2, 3, 3) and t.shape == (2, 3)


def test_property_based_bit_exact_vs_c_oracle(dev):
    """Randomised shapes / value ranges / duplicate patterns (hypothesis, 30 examples): HIP op == C oracle bit for bit."""
    from hypothesis import given, settings, strategies as st, HealthCheck
    from oracle import c_oracle
    from houv_amd.metrics import cd

    @settings(max_examples=30, deadline=None, suppress_health_check=list(HealthCheck))
    @given(B=st.integers(1, 4), N=st.integers(1, 700), M=st.integers(1, 2300), seed=st.integers(0, 2 ** 31 - 1),
           scale=st.sampled_from([1e-3, 1.0, 37.5]), quant=st.sampled_from([0, 4, 64]), dup=st.booleans())
    def run(B, N, M, seed, scale, quant, dup):
        rng = np.random.default_rng(seed)
        a = (rng.random((B, N, 3)) - 0.5) * scale
        b = (rng.random((B, M, 3)) - 0.5) * scale
        if quant:                                   # coarse lattice -> many exact ties
            a = np.round(a / scale * quant) / quant * scale
            b = np.round(b / scale * quant) / quant * scale
        if dup and M > 3:
            b[:, rng.integers(0, M, M // 3)] = b[:, rng.integers(0, M, M // 3)]
        a = a.astype(np.float32); b = b.astype(np.float32)
        d1, d2, i1, i2 = cd()(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev))
        o1, o2, j1, j2 = c_oracle.chamfer_forward(a, b)
        assert np.array_equal(d1.cpu().numpy().view(np.uint32), o1.view(np.uint32))
        assert np.array_equal(d2.cpu().numpy().view(np.uint32), o2.view(np.uint32))
        assert np.array_equal(i1.cpu().numpy(), j1) and np.array_equal(i2.cpu().numpy(), j2)

    run()


@pytest.mark.parametrize("N,M", [(700, 900), (7000, 6500)])      # LDS-accumulator path / global-atomics fallback path
def test_backward_both_paths_vs_closed_form(dev, N, M):
    from houv_amd.metrics import cd
    gen = torch.Generator().manual_seed(N)
    a = torch.rand(1, N, 3, generator=gen); b = torch.rand(1, M, 3, generator=gen)
    w1 = torch.rand(1, N, generator=gen); w2 = torch.rand(1, M, generator=gen)
    ag = a.to(dev).requires_grad_(True); bg = b.to(dev).requires_grad_(True)
    d1, d2, i1, i2 = cd()(ag, bg)
    ((d1 * w1.to(dev)).sum() + (d2 * w2.to(dev)).sum()).backward()
    gx1, gx2 = orc.chamfer_backward_closed_form(a, b, i1.cpu(), i2.cpu(), w1, w2)
    np.testing.assert_allclose(ag.grad.cpu().numpy(), gx1.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(bg.grad.cpu().numpy(), gx2.numpy(), rtol=1e-4, atol=1e-5)
