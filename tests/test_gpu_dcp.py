"""GPU parity of the DCP feature head (BASELINE configs[4]): building blocks against torch on the same
inputs, and the whole model against golden vectors from the reference's dcp.py (tests/golden/make_golden_dcp.py;
seeded random weights from tests/golden/dcp_weights.py -- the repository ships no trained DCP checkpoint)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import dcp_weights  # noqa: E402

T = torch.tensor


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("B,N,k", [(2, 48, 20), (3, 300, 20), (1, 2048, 20), (2, 1500, 8), (2, 64, 1)])
def test_knn_matches_reference_formula(dev, B, N, k):
    from houv_amd import ops
    gen = torch.Generator().manual_seed(N + k)
    x = torch.rand(B, N, 3, generator=gen)
    idx = ops.knn(x.to(dev), k).cpu().long()
    # dcp.py:35-42 in float64: -|xi|^2 + 2 xi.xj - |xj|^2, topk largest
    xd = x.double().transpose(1, 2)
    pd = -(xd ** 2).sum(1, keepdim=True).transpose(2, 1) + 2 * xd.transpose(2, 1) @ xd - (xd ** 2).sum(1, keepdim=True)
    ref = pd.topk(k=k, dim=-1)[1]
    same = (idx.sort(-1)[0] == ref.sort(-1)[0]).all(-1)
    assert same.float().mean() > 0.999                      # neighbour SETS agree (fp32 near-ties at the k-th place aside)
    assert (idx[..., 0] == torch.arange(N)).all()           # nearest = the point itself
    d = ((x.unsqueeze(2) - torch.gather(x.unsqueeze(1).expand(B, N, N, 3), 2, idx.unsqueeze(-1).expand(B, N, k, 3))) ** 2).sum(-1)
    assert (d[..., 1:] >= d[..., :-1] - 1e-7).all()        # nearest first


@pytest.mark.parametrize("B,N,k", [(3, 2048, 20), (2, 513, 20), (2, 2047, 16), (1, 4100, 20), (2, 512, 16)])
def test_knn_with_split_references_gives_the_same_lists(dev, B, N, k):
    """houv_knn's default for N >= 512 (four waves per 64 queries, a quarter of the references each, merged in LDS) against the
    single-scan kernel: identical lists.  With exact duplicates of points (distance ties) the single scan's insertion bubble is not
    stable -- a displaced entry leapfrogs its equals -- while the merge is (distance, index)-lexicographic: there the DISTANCES of the
    two lists must agree entry by entry and the split kernel's order must be the lexicographic one."""
    from houv_amd import _lib, ops
    gen = torch.Generator().manual_seed(N * 3 + k)
    x = torch.rand(B, N, 3, generator=gen)

    def both(cloud):
        try:
            _lib.debug_set("knn_split", 0)
            single = ops.knn(cloud.to(dev), k).cpu().long()
        finally:
            _lib.debug_set("knn_split", 1)
        return single, ops.knn(cloud.to(dev), k).cpu().long()

    single, split = both(x)
    np.testing.assert_array_equal(split.numpy(), single.numpy())
    x[:, N // 2:N // 2 + 40] = x[:, :40]                      # exact duplicates in another quarter of the cloud
    single, split = both(x)
    d = lambda idx: ((x.double().unsqueeze(2) - torch.gather(x.double().unsqueeze(1).expand(B, N, N, 3), 2,
                                                              idx.unsqueeze(-1).expand(B, N, k, 3))) ** 2).sum(-1)
    ds, dp = d(single), d(split)
    np.testing.assert_array_equal(dp.numpy(), ds.numpy())      # same neighbours up to exactly tied ones
    tied = dp[..., 1:] == dp[..., :-1]
    assert bool(((dp[..., 1:] >= dp[..., :-1]).all())) and bool((split[..., 1:][tied] > split[..., :-1][tied]).all())


@pytest.mark.parametrize("M,N,K,tb", [(128, 128, 64, True), (300, 70, 130, True), (1000, 512, 512, True),
                                     (257, 64, 64, True), (200, 128, 333, False), (64, 96, 48, False), (5, 3, 7, True)])
def test_gemm_plain(dev, M, N, K, tb):
    from houv_amd import ops
    gen = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=gen).to(dev)
    B = (torch.randn(N, K, generator=gen) if tb else torch.randn(K, N, generator=gen)).to(dev)
    C = ops.gemm(A, B, trans_b=tb)
    ref = (A.double() @ (B.double().t() if tb else B.double())).float()
    np.testing.assert_allclose(C.cpu().numpy(), ref.cpu().numpy(), rtol=2e-5, atol=2e-5 * np.sqrt(K))
    # A = I with an ASYMMETRIC B catches a swapped row/col map in the C write (cdna_hip_programming.md section 3)
    if M == K and tb:
        I = torch.eye(M, device=dev)
        np.testing.assert_array_equal(ops.gemm(I, B, trans_b=True).cpu().numpy(), B.t().cpu().numpy())


@pytest.mark.parametrize("M,N,K", [(1024, 512, 512), (2048, 128, 64), (256, 64, 64), (384, 256, 1024)])
def test_gemm_on_the_bf16_pipe_is_fp32_grade(dev, M, N, K):
    """houv_gemm_f32's default for full tiles: every fp32 operand split into three bf16 parts, six part products per product
    (gemm.hip, gemm_split_kernel).  Its error against a float64 product, in units of sum_k |a||b|, must stay within 1.5x the
    fp32-input MFMA kernel's (measured: 0.85x); the three-product form is 2^-16-grade; an identity operand reproduces B exactly."""
    from houv_amd import _lib, ops
    gen = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=gen) * torch.rand(M, 1, generator=gen).mul(6).exp()).to(dev)    # rows of very different scale
    B = torch.randn(N, K, generator=gen).to(dev)
    ref = A.double() @ B.double().t()
    unit = A.double().abs() @ B.double().abs().t()
    err = {}
    try:
        for mode in (0, 6, 3):
            _lib.debug_set("gemm_split", mode)
            err[mode] = float(((ops.gemm(A, B, trans_b=True).double() - ref).abs() / unit).max())
        _lib.debug_set("gemm_split", 6)
        if M >= K and K % 32 == 0:
            I = torch.eye(M, K, device=dev)
            np.testing.assert_array_equal(ops.gemm(I, B, trans_b=True)[:K].cpu().numpy(), B.t().cpu().numpy())
    finally:
        _lib.debug_set("gemm_split", 6)
    assert err[0] < 8e-7 and err[6] < 1.5 * err[0] and err[3] < 2e-5, err


def test_gemm_epilogue_and_batched_strided(dev):
    from houv_amd import ops
    gen = torch.Generator().manual_seed(5)
    A = torch.randn(333, 64, generator=gen).to(dev); W = torch.randn(128, 64, generator=gen).to(dev)
    sc = torch.rand(128, generator=gen).to(dev) + 0.5; sh = torch.randn(128, generator=gen).to(dev)
    R = torch.randn(333, 128, generator=gen).to(dev)
    C = ops.gemm(A, W, scale=sc, shift=sh, residual=R, relu=True, alpha=0.5)
    ref = torch.relu((0.5 * (A.double() @ W.double().t())) * sc.double() + sh.double() + R.double()).float()
    np.testing.assert_allclose(C.cpu().numpy(), ref.cpu().numpy(), rtol=2e-5, atol=1e-4)
    C = ops.gemm(A, W, shift=sh)                                   # bias only
    np.testing.assert_allclose(C.cpu().numpy(), (A.double() @ W.double().t() + sh.double()).float().cpu().numpy(), rtol=2e-5, atol=1e-4)
    # attention-shaped: per-head strided views, both products
    P, H, Nq, Nk, dk = 2, 4, 150, 170, 128
    Q = torch.randn(P, Nq, H, dk, generator=gen).to(dev); Kt = torch.randn(P, Nk, H, dk, generator=gen).to(dev)
    V = torch.randn(P, Nk, H, dk, generator=gen).to(dev)
    S = ops.gemm(Q.permute(0, 2, 1, 3), Kt.permute(0, 2, 1, 3), trans_b=True, alpha=0.25)
    refS = 0.25 * torch.einsum("pqhd,pkhd->phqk", Q.double(), Kt.double())
    np.testing.assert_allclose(S.cpu().numpy(), refS.float().cpu().numpy(), rtol=2e-5, atol=2e-4)
    ctx = torch.zeros(P, Nq, H, dk, device=dev)
    ops.gemm(S, V.permute(0, 2, 1, 3), ctx.permute(0, 2, 1, 3), trans_b=False)
    refC = torch.einsum("phqk,pkhd->pqhd", S.double(), V.double())
    np.testing.assert_allclose(ctx.cpu().numpy(), refC.float().cpu().numpy(), rtol=2e-5, atol=2e-3)


def test_layernorm_softmax_corr_maxk(dev):
    from houv_amd import ops
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(37, 512, generator=gen).to(dev); a = torch.rand(512, generator=gen).to(dev); b = torch.randn(512, generator=gen).to(dev)
    r = torch.randn(37, 512, generator=gen).to(dev)
    ref = a * (x - x.mean(-1, keepdim=True)) / (x.std(-1, keepdim=True) + 1e-6) + b      # dcp.py:151-154
    np.testing.assert_allclose(ops.layernorm(x, a, b).cpu().numpy(), ref.cpu().numpy(), atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(ops.layernorm(x, a, b, residual=r).cpu().numpy(), (ref + r).cpu().numpy(), atol=3e-6, rtol=1e-5)
    s = (torch.randn(3, 50, 77, generator=gen) * 4).to(dev)
    want = torch.softmax(s, dim=-1)
    pts = torch.randn(3, 77, 3, generator=gen).to(dev)
    corr = ops.softmax_corr(s, pts)
    np.testing.assert_allclose(corr.cpu().numpy(), torch.matmul(pts.transpose(1, 2), want.transpose(2, 1)).cpu().numpy(), atol=2e-6)
    np.testing.assert_allclose(ops.softmax_rows_(s.clone()).cpu().numpy(), want.cpu().numpy(), atol=1e-6)
    act = torch.randn(11 * 20, 64, generator=gen).to(dev)
    out = torch.zeros(11, 512, device=dev)
    ops.max_over_k(act, 20, out, 64)
    np.testing.assert_array_equal(out[:, 64:128].cpu().numpy(), act.view(11, 20, 64).max(1)[0].cpu().numpy())
    assert float(out[:, :64].abs().max()) == 0 and float(out[:, 128:].abs().max()) == 0


def _model(dev):
    from houv_amd.models.dcp import Model
    net = Model(args=None)
    state = {k: T(v) for k, v in dcp_weights.make_state(1234).items()}
    missing, unexpected = net.load_state_dict(state, strict=False)
    assert not unexpected and all(m.endswith("num_batches_tracked") or m == "head.reflect" for m in missing), (missing, unexpected)
    return net.to(dev)


@pytest.mark.parametrize("P,Nq,Nk", [(2, 300, 77), (1, 2048, 2048), (3, 128, 31), (2, 33, 500), (1, 1, 1)])
def test_fused_attention_vs_float64(dev, P, Nq, Nk):
    """houv_attention_f32 (dcp.py:26-32 fused: the scores never reach HBM) against torch in float64 on the same operands,
    ragged sizes (key-block tails, query-block tails) included; operands are the strided per-head views the model passes."""
    from houv_amd import ops
    H, dk = 4, 128
    gen = torch.Generator().manual_seed(P * 1000 + Nq + Nk)
    q = torch.randn(P, Nq, H * dk, generator=gen).to(dev)
    k = torch.randn(P, Nk, H * dk, generator=gen).to(dev)
    v = torch.randn(P, Nk, H * dk, generator=gen).to(dev)
    scale = 1.0 / np.sqrt(dk)
    out = ops.attention(q.view(P, Nq, H, dk), k.view(P, Nk, H, dk), v.view(P, Nk, H, dk), scale)
    qd, kd, vd = (t.double().view(P, -1, H, dk).permute(0, 2, 1, 3) for t in (q, k, v))
    ref = (torch.softmax(qd @ kd.transpose(-1, -2) * scale, dim=-1) @ vd).permute(0, 2, 1, 3)
    np.testing.assert_allclose(out.cpu().numpy(), ref.float().cpu().numpy(), rtol=2e-5, atol=2e-5)
    # large-magnitude logits: the running maximum must keep exp() in range
    out2 = ops.attention((q * 30).view(P, Nq, H, dk), k.view(P, Nk, H, dk), v.view(P, Nk, H, dk), scale)
    ref2 = (torch.softmax((qd * 30) @ kd.transpose(-1, -2) * scale, dim=-1) @ vd).permute(0, 2, 1, 3)
    assert bool(torch.isfinite(out2).all())
    np.testing.assert_allclose(out2.cpu().numpy(), ref2.float().cpu().numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("P,Nq,Nk,spread", [(2, 256, 128, 1.0), (1, 2048, 2048, 1.0), (2, 128, 32, 1.0), (1, 1024, 2048, 4.0)])
def test_attention_on_the_bf16_pipe_is_fp32_grade(dev, P, Nq, Nk, spread):
    """Full tiles run attention_split_kernel (Q, K, V and the probabilities as three bf16 parts, six part products per product): its
    error against float64 must stay within 1.25x the fp32-input MFMA kernel's on the same operands (measured: 0.8x), peaked
    softmaxes (spread 4: logits of +-40) included."""
    from houv_amd import _lib, ops
    H, dk = 4, 128
    gen = torch.Generator().manual_seed(P * 77 + Nq + Nk)
    q = (torch.randn(P, Nq, H * dk, generator=gen) * spread).to(dev)
    k = torch.randn(P, Nk, H * dk, generator=gen).to(dev)
    v = torch.randn(P, Nk, H * dk, generator=gen).to(dev)
    scale = 1.0 / np.sqrt(dk)
    qd, kd, vd = (t.double().view(P, -1, H, dk).permute(0, 2, 1, 3) for t in (q, k, v))
    ref = (torch.softmax(qd @ kd.transpose(-1, -2) * scale, dim=-1) @ vd).permute(0, 2, 1, 3)
    err = {}
    try:
        for mode in (0, 1):
            _lib.debug_set("attn_split", mode)
            out = ops.attention(q.view(P, Nq, H, dk), k.view(P, Nk, H, dk), v.view(P, Nk, H, dk), scale)
            err[mode] = float((out.double() - ref).abs().max())
    finally:
        _lib.debug_set("attn_split", 1)
    assert err[0] < 4e-5 * spread and err[1] <= 1.25 * err[0] + 1e-7, err


def test_model_is_the_same_with_and_without_fused_attention(dev, monkeypatch):
    from houv_amd import ops, synthetic
    net = _model(dev)
    src, tgt, _ = synthetic.make_pairs(2, 300, seed=12)
    a = net(src.to(dev), tgt.to(dev))
    monkeypatch.setattr(ops, "FUSED_ATTENTION", False)
    b = net(src.to(dev), tgt.to(dev))
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=2e-4)


def test_graph_replayed_chunks_equal_eager_ones(dev):
    """Model.registration replays a chunk's launches as one HIP graph from the chunk shape's third occurrence on (first: eager,
    second: capture); answers must equal the eager ones bit for bit, also after the weights change (new capture)."""
    net = _model(dev)
    net.pairs_per_chunk = 2
    gen = torch.Generator().manual_seed(77)
    src = torch.rand(8, 256, 3, generator=gen).to(dev)
    tgt = torch.rand(8, 256, 3, generator=gen).to(dev)
    net.use_graphs = False
    want = net(src, tgt)
    net.use_graphs = True
    got = net(src, tgt)                              # chunks 1 (eager), 2 (capture), 3-4 (replay)
    assert net.use_graphs and any(e["graph"] is not None for e in net._graphs.values()), "capture fell back to eager"
    assert torch.equal(got, want)
    assert torch.equal(net(src.flip(0), tgt.flip(0)), want.flip(0))          # replay on other inputs
    with torch.no_grad():
        net.emb_nn.conv2.weight.mul_(1.01)           # a new version of a weight: the old graph must not be replayed
    net.use_graphs = False
    want2 = net(src, tgt)
    net.use_graphs = True
    assert torch.equal(net(src, tgt), want2) and not torch.equal(want2, want)


def test_state_dict_names_equal_reference():
    from houv_amd.models.dcp import Model
    mine = {k for k in Model(None).state_dict() if not k.endswith("num_batches_tracked") and k != "head.reflect"}
    assert mine == {n for n, _ in dcp_weights.spec()}


def test_model_vs_reference_golden(golden, dev):
    g = golden("g9_dcp.npz")
    net = _model(dev)
    # small case: every intermediate tensor
    s, t = T(g["small_src"]).to(dev), T(g["small_tgt"]).to(dev)
    with torch.no_grad():
        es, et = net.emb_nn(s), net.emb_nn(t)
        np.testing.assert_allclose(es.transpose(1, 2).cpu().numpy(), g["small_emb_src"], atol=2e-4, rtol=1e-4)
        np.testing.assert_allclose(et.transpose(1, 2).cpu().numpy(), g["small_emb_tgt"], atol=2e-4, rtol=1e-4)
        ed = net.pointer.model
        zero_t, zero_s = torch.zeros_like(et), torch.zeros_like(es)
        pt = ed(es, et, add_to=zero_t); ps = ed(et, es, add_to=zero_s)
        np.testing.assert_allclose(pt.transpose(1, 2).cpu().numpy(), g["small_ptr_tgt"], atol=1e-3, rtol=1e-3)
        np.testing.assert_allclose(ps.transpose(1, 2).cpu().numpy(), g["small_ptr_src"], atol=1e-3, rtol=1e-3)
    for name in ("small", "mid"):
        s, t = T(g[f"{name}_src"]).to(dev), T(g[f"{name}_tgt"]).to(dev)
        T12 = net(s, t)
        np.testing.assert_allclose(T12.cpu().numpy(), g[f"{name}_T12"], atol=2e-3)
        assert np.array_equal(T12[:, 3].cpu().numpy(), np.broadcast_to([0, 0, 0, 1], (2, 4)).astype(np.float32))
    with torch.no_grad():
        es = net.emb_nn(T(g["mid_src"]).to(dev))
    np.testing.assert_allclose(es.transpose(1, 2).cpu().numpy()[:, ::8, ::8], g["mid_emb_src_s"], atol=2e-4, rtol=1e-4)


def test_model_vs_reference_golden_2048_points(golden, dev):
    """G19: the reference's dcp.py on ONE 2048x2048-point pair -- BASELINE configs[4]'s cloud size, the
    size bench.py --dcp runs -- with the seeded weights: k-NN sets of the DGCNN graph, a strided sample of the embeddings
    and of the pointer output, the SVD head's (R, t) and the final T_12 (`tests/golden/make_golden_dcp.py 2048`)."""
    from houv_amd import ops
    g = golden("g19_dcp2048.npz")
    net = _model(dev)
    s, t = T(g["full_src"]).to(dev), T(g["full_tgt"]).to(dev)
    assert s.shape == (1, 2048, 3)
    idx = ops.knn(s, 20).cpu().long()
    same = (idx.sort(-1)[0] == T(g["full_knn_src"]).long().sort(-1)[0]).all(-1)
    assert same.float().mean() > 0.995            # neighbour SETS (fp32 near-ties at the 20th place aside)
    with torch.no_grad():
        es, et = net.emb_nn(s), net.emb_nn(t)
        np.testing.assert_allclose(es.transpose(1, 2).cpu().numpy()[:, ::8, ::8], g["full_emb_src_s"], atol=5e-4, rtol=1e-3)
        pt = net.pointer.model(es, et, add_to=torch.zeros_like(et))
        np.testing.assert_allclose(pt.transpose(1, 2).cpu().numpy()[:, ::8, ::8], g["full_ptr_tgt_s"], atol=2e-3, rtol=2e-3)
        T12 = net(s, t)
    np.testing.assert_allclose(T12.cpu().numpy(), g["full_T12"], atol=3e-3)
    np.testing.assert_allclose(T12[:, :3, :3].cpu().numpy(), g["full_R"], atol=3e-3)
    np.testing.assert_allclose(T12[:, :3, 3].cpu().numpy(), g["full_t"], atol=3e-3)


def test_model_with_T_gt_returns_reference_tuple(dev):
    from houv_amd import synthetic
    net = _model(dev)
    src, tgt, pose = synthetic.make_pairs(3, 128, seed=8)
    out = net(src.to(dev), tgt.to(dev), pose.to(dev))
    assert len(out) == 5 and out[1].shape == (3,) and out[0].ndim == 0


def test_pointnet_embedding_vs_reference_golden(golden, dev):
    from houv_amd.models.dcp import PointNet
    g = golden("g9_dcp.npz")
    pn = PointNet(512)
    missing, unexpected = pn.load_state_dict({k: T(v) for k, v in dcp_weights.make_pointnet_state(99).items()}, strict=False)
    assert not unexpected and all(m.endswith("num_batches_tracked") for m in missing)
    emb = pn.to(dev)(T(g["small_src"]).to(dev))
    np.testing.assert_allclose(emb.transpose(1, 2).cpu().numpy(), g["pointnet_emb"], atol=2e-4, rtol=1e-4)
