"""Mirror of the three `mm3d_pn2` ops the registration side imports (registration/train_utils.py:20:
``from mm3d_pn2 import furthest_point_sample, gather_points``; ``three_nn`` from utils/mm3d_pn2/ops/interpolate),
on the gfx950 kernels of houv_amd/csrc/pointops.hip.  Forward only (the reference's FPS / three_nn are
non-differentiable; gather_points' backward is a scatter that nothing on the HOUV path needs)."""
import torch

from . import _lib

_F32, _I32 = torch.float32, torch.int32


def _f32(t, name):
    """The kernels read raw fp32: convert other float dtypes (the result must stay referenced until after the launch)."""
    if not t.is_floating_point():
        raise _lib.HouvHipError(f"{name}: expected a floating-point tensor, got {t.dtype}")
    return t if t.dtype == _F32 else t.float()


def _i32(t, name, bound):
    """Indices arrive as int32 (the reference's ops) or int64 (torch.topk / argsort defaults): convert the latter, and
    refuse anything outside [0, bound) -- the gather kernels do not bounds-check."""
    if t.dtype not in (_I32, torch.int64):
        raise _lib.HouvHipError(f"{name}: expected int32 / int64 indices, got {t.dtype}")
    if t.numel() and (int(t.min()) < 0 or int(t.max()) >= bound):
        raise _lib.HouvHipError(f"{name}: index out of range [0, {bound})")
    return t if t.dtype == _I32 else t.to(_I32)


def furthest_point_sample(points_xyz, num_points):
    """points_xyz (B,N,3) contiguous, N >= num_points -> (B,num_points) int32 indices (furthest_point_sample.py:15-36)."""
    _lib.require_gpu(points_xyz)
    if points_xyz.dim() != 3 or points_xyz.shape[2] != 3 or not 0 < int(num_points) <= points_xyz.shape[1]:
        raise _lib.HouvHipError("furthest_point_sample: expected points_xyz[B,N,3] and 0 < num_points <= N")
    pts = _f32(points_xyz, "points_xyz")                 # kept alive in a local until the launch is enqueued
    B, N, _ = pts.shape
    out = torch.empty((B, num_points), dtype=_I32, device=pts.device)
    with torch.cuda.device(pts.device):
        ok = _lib.load().houv_furthest_point_sample(_lib.ptr(pts), B, N, int(num_points), _lib.ptr(out),
                                                    _lib.stream_of(pts))
    _lib.check(ok, "houv_furthest_point_sample")
    return out


def gather_points(features, indices):
    """features (B,C,N), indices (B,M) int32 -> (B,C,M) (gather_points.py:14-35)."""
    _lib.require_gpu(features, indices)
    if features.dim() != 3 or indices.dim() != 2 or indices.shape[0] != features.shape[0]:
        raise _lib.HouvHipError("gather_points: expected features[B,C,N], indices[B,M]")
    B, C, N = features.shape
    feats = _f32(features, "features")
    idx = _i32(indices, "indices", N).contiguous()
    M = idx.shape[1]
    out = torch.empty((B, C, M), dtype=_F32, device=feats.device)
    with torch.cuda.device(feats.device):
        ok = _lib.load().houv_gather_points(_lib.ptr(feats), _lib.ptr(idx), B, C, N, M, _lib.ptr(out),
                                            _lib.stream_of(feats))
    _lib.check(ok, "houv_gather_points")
    return out


def three_nn(target, source):
    """target (B,N,3), source (B,M,3) -> (dist (B,N,3) L2 distances, idx (B,N,3)) (three_nn.py:11-37 returns sqrt(dist2))."""
    _lib.require_gpu(target, source)
    if target.dim() != 3 or source.dim() != 3 or target.shape[2] != 3 or source.shape[2] != 3 or source.shape[0] != target.shape[0] \
            or source.shape[1] < 3:
        raise _lib.HouvHipError("three_nn: expected target[B,N,3], source[B,M>=3,3]")
    target, source = _f32(target, "target"), _f32(source, "source")
    B, N, _ = target.shape
    M = source.shape[1]
    d2 = torch.empty((B, N, 3), dtype=_F32, device=target.device)
    idx = torch.empty((B, N, 3), dtype=_I32, device=target.device)
    with torch.cuda.device(target.device):
        ok = _lib.load().houv_knn_cross(_lib.ptr(target), _lib.ptr(source), B, N, M, 3, _lib.ptr(d2), _lib.ptr(idx),
                                        _lib.stream_of(target))
    _lib.check(ok, "houv_knn_cross")
    return torch.sqrt(d2), idx
