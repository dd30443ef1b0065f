"""Mirror of the three `mm3d_pn2` ops the registration side imports (registration/train_utils.py:20:
``from mm3d_pn2 import furthest_point_sample, gather_points``; ``three_nn`` from utils/mm3d_pn2/ops/interpolate),
on the gfx950 kernels of houv_amd/csrc/pointops.hip.  Forward only (the reference's FPS / three_nn are
non-differentiable; gather_points' backward is a scatter that nothing on the HOUV path needs)."""
import torch

from . import _lib

_F32, _I32 = torch.float32, torch.int32


def furthest_point_sample(points_xyz, num_points):
    """points_xyz (B,N,3) contiguous, N >= num_points -> (B,num_points) int32 indices (furthest_point_sample.py:15-36)."""
    _lib.require_gpu(points_xyz)
    B, N, _ = points_xyz.shape
    out = torch.empty((B, num_points), dtype=_I32, device=points_xyz.device)
    with torch.cuda.device(points_xyz.device):
        ok = _lib.load().houv_furthest_point_sample(_lib.ptr(points_xyz.float()), B, N, int(num_points), _lib.ptr(out),
                                                    _lib.stream_of(points_xyz))
    _lib.check(ok, "houv_furthest_point_sample")
    return out


def gather_points(features, indices):
    """features (B,C,N), indices (B,M) int32 -> (B,C,M) (gather_points.py:14-35)."""
    _lib.require_gpu(features, indices)
    B, C, N = features.shape
    M = indices.shape[1]
    out = torch.empty((B, C, M), dtype=_F32, device=features.device)
    with torch.cuda.device(features.device):
        ok = _lib.load().houv_gather_points(_lib.ptr(features), _lib.ptr(indices), B, C, N, M, _lib.ptr(out),
                                            _lib.stream_of(features))
    _lib.check(ok, "houv_gather_points")
    return out


def three_nn(target, source):
    """target (B,N,3), source (B,M,3) -> (dist (B,N,3) L2 distances, idx (B,N,3)) (three_nn.py:11-37 returns sqrt(dist2))."""
    _lib.require_gpu(target, source)
    B, N, _ = target.shape
    M = source.shape[1]
    d2 = torch.empty((B, N, 3), dtype=_F32, device=target.device)
    idx = torch.empty((B, N, 3), dtype=_I32, device=target.device)
    with torch.cuda.device(target.device):
        ok = _lib.load().houv_knn_cross(_lib.ptr(target), _lib.ptr(source), B, N, M, 3, _lib.ptr(d2), _lib.ptr(idx),
                                        _lib.stream_of(target))
    _lib.check(ok, "houv_knn_cross")
    return torch.sqrt(d2), idx
