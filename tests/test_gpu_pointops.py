"""GPU: FPS / three_nn / gather_points mirrors against direct restatements (parity unpinned: the reference's versions are
CUDA extensions with no fixtures)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _fps_ref(p, m):
    n = len(p)
    md = np.full(n, 1e10, np.float32)
    out = [0]
    for _ in range(1, m):
        q = p[out[-1]]
        d = ((p - q) ** 2).astype(np.float32)
        d = (d[:, 0] + d[:, 1]) + d[:, 2]
        md = np.minimum(md, d)
        out.append(int(np.argmax(md)))           # argmax = lowest index on ties
    return np.array(out, np.int32)


@pytest.mark.parametrize("B,N,m", [(2, 300, 64), (1, 2048, 512), (2, 5000, 100), (3, 100, 100)])
def test_fps(dev, B, N, m):
    from houv_amd.mm3d_pn2 import furthest_point_sample
    gen = torch.Generator().manual_seed(N)
    x = torch.rand(B, N, 3, generator=gen)
    idx = furthest_point_sample(x.to(dev), m).cpu().numpy()
    for b in range(B):
        ref = _fps_ref(x[b].numpy(), m)
        agree = (idx[b] == ref).mean()
        assert agree > 0.98, agree               # fp32 sum-order near-ties can flip a late pick
        assert len(set(idx[b].tolist())) == m or m > N


def test_three_nn_and_gather(dev):
    from houv_amd.mm3d_pn2 import gather_points, three_nn
    gen = torch.Generator().manual_seed(3)
    t = torch.rand(2, 333, 3, generator=gen); s = torch.rand(2, 1500, 3, generator=gen)
    d, i = three_nn(t.to(dev), s.to(dev))
    D = torch.cdist(t.double(), s.double())
    rd, ri = D.topk(3, dim=-1, largest=False)
    np.testing.assert_allclose(d.cpu().numpy(), rd.float().numpy(), atol=1e-5)
    assert (i.cpu().long() == ri).float().mean() > 0.999
    f = torch.randn(2, 7, 1500, generator=gen)
    g = gather_points(f.to(dev), i[:, :, 0].contiguous())
    np.testing.assert_array_equal(g.cpu().numpy(), torch.gather(f, 2, i[:, :, 0].cpu().long().unsqueeze(1).expand(2, 7, 333)).numpy())


def test_combine(dev):
    from houv_amd.train_utils import combine
    a = torch.rand(2, 1500, 3, device=dev); b = torch.rand(2, 1500, 3, device=dev)
    c = combine(a, b)
    assert c.shape == (2, 2048, 3)
    allp = torch.cat([a, b], 1)
    assert torch.equal(c[:, 0], allp[:, 0])                          # FPS starts at point 0
    # every sampled point is exactly the input point FPS chose
    from houv_amd.mm3d_pn2 import furthest_point_sample
    idx = furthest_point_sample(allp.contiguous(), 2048).long()
    assert torch.equal(c, torch.gather(allp, 1, idx.unsqueeze(2).expand(-1, -1, 3)))


def test_wrappers_validate_dtypes_and_indices(dev):
    """torch-default int64 indices (argsort / topk) and non-fp32 features must not be reinterpreted silently;
    out-of-range indices must be refused (the gather kernel does not bounds-check)."""
    from houv_amd import _lib, ops
    from houv_amd.mm3d_pn2 import furthest_point_sample, gather_points, three_nn
    g = torch.Generator().manual_seed(0)
    feats = torch.rand(2, 5, 40, generator=g).to(dev)
    idx64 = torch.argsort(torch.rand(2, 40, generator=g), dim=1)[:, :7].to(dev)          # int64, like torch.topk / argsort give
    want = torch.gather(feats, 2, idx64.unsqueeze(1).expand(2, 5, 7))
    assert torch.equal(gather_points(feats, idx64), want)
    assert torch.equal(gather_points(feats.double(), idx64.int()), want)                 # fp64 features are converted, not reinterpreted
    with pytest.raises(_lib.HouvHipError):
        gather_points(feats, torch.full((2, 3), 40, dtype=torch.int64, device=dev))      # out of range
    with pytest.raises(_lib.HouvHipError):
        gather_points(feats, idx64.float())
    pts = torch.rand(2, 64, 3, generator=g).to(dev)
    assert torch.equal(furthest_point_sample(pts.double(), 9), furthest_point_sample(pts, 9))
    with pytest.raises(_lib.HouvHipError):
        furthest_point_sample(pts, 65)
    d_a, i_a = three_nn(pts.double(), pts[:, :20].double())
    d_b, i_b = three_nn(pts, pts[:, :20].contiguous())
    assert torch.equal(i_a, i_b) and torch.equal(d_a, d_b)
    with pytest.raises(_lib.HouvHipError):
        ops.softmax_rows_(torch.rand(4, 8, device=dev).double())
    with pytest.raises(_lib.HouvHipError):
        ops.layernorm(torch.rand(4, 8, device=dev), torch.ones(7, device=dev), torch.zeros(8, device=dev))
    with pytest.raises(_lib.HouvHipError):
        ops.edgeconv1(pts, torch.zeros(2, 64, 20, dtype=torch.int64, device=dev), torch.rand(64, 6, device=dev),
                      torch.ones(64, device=dev), torch.zeros(64, device=dev))
