"""Thin Python operators over the C ABI (include/houv_hip.h), plus their registration as PyTorch-ROCm
custom ops ``torch.ops.houv.*``.  Outputs are caller-allocated and written in place, as in the reference's
pybind module (utils/metrics/CD/chamfer3D/chamfer_cuda.cpp:17-33)."""
import torch

from . import _lib

_F32, _I32, _F64 = torch.float32, torch.int32, torch.float64


def _want(t, dtype, name):
    if t.dtype != dtype:
        raise _lib.HouvHipError(f"{name}: expected {dtype}, got {t.dtype}")


def chamfer_forward(xyz1, xyz2, dist1, dist2, idx1, idx2):
    """``chamfer_3D.forward(xyz1, xyz2, dist1, dist2, idx1, idx2)`` (chamfer_cuda.cpp:17-19). Returns 1."""
    _lib.require_gpu(xyz1, xyz2, dist1, dist2, idx1, idx2)
    for t, d, n in ((xyz1, _F32, "xyz1"), (xyz2, _F32, "xyz2"), (dist1, _F32, "dist1"), (dist2, _F32, "dist2"),
                    (idx1, _I32, "idx1"), (idx2, _I32, "idx2")):
        _want(t, d, n)
    B, N, _ = xyz1.shape
    M = xyz2.shape[1]
    if xyz1.shape[2] != 3 or xyz2.shape[2] != 3 or xyz2.shape[0] != B:
        raise _lib.HouvHipError("chamfer_forward: expected xyz1[B,N,3], xyz2[B,M,3]")
    if dist1.numel() != B * N or idx1.numel() != B * N or dist2.numel() != B * M or idx2.numel() != B * M:
        raise _lib.HouvHipError("chamfer_forward: output shapes do not match [B,N] / [B,M]")
    with torch.cuda.device(xyz1.device):
        ok = _lib.load().houv_chamfer_forward(_lib.ptr(xyz1), _lib.ptr(xyz2), B, N, M, _lib.ptr(dist1), _lib.ptr(dist2),
                                              _lib.ptr(idx1), _lib.ptr(idx2), _lib.stream_of(xyz1))
    _lib.check(ok, "houv_chamfer_forward")
    return 1


def chamfer_backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
    """``chamfer_3D.backward(...)`` with the reference's argument order (chamfer_cuda.cpp:22-26).
    Accumulates into the (caller zero-filled) gradxyz1/gradxyz2."""
    _lib.require_gpu(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2)
    for t, d, n in ((xyz1, _F32, "xyz1"), (xyz2, _F32, "xyz2"), (gradxyz1, _F32, "gradxyz1"),
                    (gradxyz2, _F32, "gradxyz2"), (graddist1, _F32, "graddist1"), (graddist2, _F32, "graddist2"),
                    (idx1, _I32, "idx1"), (idx2, _I32, "idx2")):
        _want(t, d, n)
    B, N, _ = xyz1.shape
    M = xyz2.shape[1]
    if (gradxyz1.shape != xyz1.shape or gradxyz2.shape != xyz2.shape or graddist1.numel() != B * N
            or graddist2.numel() != B * M or idx1.numel() != B * N or idx2.numel() != B * M):
        raise _lib.HouvHipError("chamfer_backward: shape mismatch")
    with torch.cuda.device(xyz1.device):
        ok = _lib.load().houv_chamfer_backward(_lib.ptr(xyz1), _lib.ptr(xyz2), B, N, M, _lib.ptr(graddist1),
                                               _lib.ptr(graddist2), _lib.ptr(idx1), _lib.ptr(idx2), _lib.ptr(gradxyz1),
                                               _lib.ptr(gradxyz2), _lib.stream_of(xyz1))
    _lib.check(ok, "houv_chamfer_backward")
    return 1


def kabsch(src, corr, weights=None):
    """Batched Kabsch: src, corr [B,3,N]; weights [B,1,N] or None -> R[B,3,3], t[B,3]
    (SVDHead.forward, registration/model_utils.py:220-255)."""
    _lib.require_gpu(src, corr, weights)
    _want(src, _F32, "src"); _want(corr, _F32, "corr")
    B, C, N = src.shape
    if C != 3 or corr.shape != src.shape:
        raise _lib.HouvHipError("kabsch: expected src, corr [B,3,N]")
    if weights is not None:
        _want(weights, _F32, "weights")
        if weights.numel() != B * N:
            raise _lib.HouvHipError("kabsch: expected weights [B,1,N]")
    R = torch.empty((B, 3, 3), dtype=_F32, device=src.device)
    t = torch.empty((B, 3), dtype=_F32, device=src.device)
    with torch.cuda.device(src.device):
        ok = _lib.load().houv_kabsch(_lib.ptr(src), _lib.ptr(corr), _lib.ptr(weights), B, N, _lib.ptr(R), _lib.ptr(t),
                                     _lib.stream_of(src))
    _lib.check(ok, "houv_kabsch")
    return R, t


def pose_forward(params, angle_base, trans_mode=0, src=None):
    """params fp32 [n,8] -> (R[n,3,3], T[n,3]) and, with src[n,N,3], moved[n,N,3] (houv.py:94-103)."""
    _lib.require_gpu(params, src)
    _want(params, _F32, "params")
    n = params.shape[0]
    R = torch.empty((n, 3, 3), dtype=_F32, device=params.device)
    T = torch.empty((n, 3), dtype=_F32, device=params.device)
    moved, N = None, 0
    if src is not None:
        _want(src, _F32, "src")
        N = src.shape[1]
        moved = torch.empty_like(src)
    with torch.cuda.device(params.device):
        ok = _lib.load().houv_pose_forward(_lib.ptr(params), n, int(angle_base), int(trans_mode), _lib.ptr(src), N,
                                           _lib.ptr(R), _lib.ptr(T), _lib.ptr(moved), _lib.stream_of(params))
    _lib.check(ok, "houv_pose_forward")
    return (R, T) if src is None else (R, T, moved)


def solve_iterate(src, tgt, state, K, *, steps_done, n_iters, angle_base, trans_mode, use_views, f64_params, k_full,
                  k_view, lr, loss_scale, betas=(0.9, 0.999), eps=1e-8, want_grad=False, want_cd=False):
    """One launch of the fused HOUV loop (houv_solve_iterate).  ``state`` [P*K,24] fp64 is updated in place.
    Returns dict(score[P*K], loss[P*K], R[P*K,3,3], T[P*K,3][, grad[P*K,8]][, cd[P*K,8]]) of the LAST forward."""
    _lib.require_gpu(src, tgt, state)
    _want(src, _F32, "src"); _want(tgt, _F32, "tgt"); _want(state, _F64, "state")
    P, N, _ = src.shape
    M = tgt.shape[1]
    if tgt.shape[0] != P or src.shape[2] != 3 or tgt.shape[2] != 3:
        raise _lib.HouvHipError("solve_iterate: expected src[P,N,3], tgt[P,M,3]")
    if tuple(state.shape) != (P * K, 24):
        raise _lib.HouvHipError(f"solve_iterate: state must be [{P * K},24], got {tuple(state.shape)}")
    dev = src.device
    n = P * K
    out = dict(score=torch.empty(n, dtype=_F32, device=dev), loss=torch.empty(n, dtype=_F32, device=dev),
               R=torch.empty((n, 3, 3), dtype=_F32, device=dev), T=torch.empty((n, 3), dtype=_F32, device=dev))
    if want_grad:
        out["grad"] = torch.empty((n, 8), dtype=_F32, device=dev)
    if want_cd:
        out["cd"] = torch.empty((n, 8), dtype=_F32, device=dev)
    with torch.cuda.device(dev):
        ok = _lib.load().houv_solve_iterate(
            _lib.ptr(src), _lib.ptr(tgt), P, N, M, int(K), _lib.ptr(state), int(steps_done), int(n_iters),
            int(angle_base), int(trans_mode), int(bool(use_views)), int(bool(f64_params)), int(k_full), int(k_view),
            float(lr), float(betas[0]), float(betas[1]), float(eps), float(loss_scale), _lib.ptr(out["score"]),
            _lib.ptr(out["loss"]), _lib.ptr(out["R"]), _lib.ptr(out["T"]), _lib.ptr(out.get("grad")),
            _lib.ptr(out.get("cd")), _lib.stream_of(src))
    _lib.check(ok, "houv_solve_iterate")
    return out


def icp_refine(src, tgt, init=None, max_correspondence_distance=0.02, max_iteration=500, relative_fitness=1e-6,
               relative_rmse=1e-6):
    """Batched point-to-point ICP (houv_icp_refine; Open3D registration_icp semantics, train_ICP.py:148-151).
    src[P,N,3], tgt[P,M,3], init[P,4,4]|None -> dict(T[P,4,4], fitness[P], inlier_rmse[P], iterations[P])."""
    _lib.require_gpu(src, tgt, init)
    _want(src, _F32, "src"); _want(tgt, _F32, "tgt")
    P, N, _ = src.shape
    M = tgt.shape[1]
    if tgt.shape[0] != P or src.shape[2] != 3 or tgt.shape[2] != 3:
        raise _lib.HouvHipError("icp_refine: expected src[P,N,3], tgt[P,M,3]")
    if init is not None:
        _want(init, _F32, "init")
        if tuple(init.shape) != (P, 4, 4):
            raise _lib.HouvHipError("icp_refine: init must be [P,4,4]")
    dev = src.device
    T = torch.empty((P, 4, 4), dtype=_F32, device=dev)
    fit = torch.empty(P, dtype=_F32, device=dev)
    rmse = torch.empty(P, dtype=_F32, device=dev)
    iters = torch.empty(P, dtype=_I32, device=dev)
    with torch.cuda.device(dev):
        ok = _lib.load().houv_icp_refine(_lib.ptr(src), _lib.ptr(tgt), P, N, M, _lib.ptr(init),
                                         float(max_correspondence_distance), int(max_iteration), float(relative_fitness),
                                         float(relative_rmse), _lib.ptr(T), _lib.ptr(fit), _lib.ptr(rmse),
                                         _lib.ptr(iters), _lib.stream_of(src))
    _lib.check(ok, "houv_icp_refine")
    return dict(T=T, fitness=fit, inlier_rmse=rmse, iterations=iters)


# ---------------------------------------------------------------------------------------------------
# torch.ops.houv.* registration (PyTorch-ROCm custom ops; the schema marks the in-place outputs)
# ---------------------------------------------------------------------------------------------------
_registered = False


def register_torch_ops():
    global _registered
    if _registered:
        return
    lib = torch.library.Library("houv", "DEF")
    lib.define("chamfer_forward(Tensor xyz1, Tensor xyz2, Tensor(a!) dist1, Tensor(b!) dist2, Tensor(c!) idx1, "
               "Tensor(d!) idx2) -> int")
    lib.define("chamfer_backward(Tensor xyz1, Tensor xyz2, Tensor(a!) gradxyz1, Tensor(b!) gradxyz2, Tensor graddist1, "
               "Tensor graddist2, Tensor idx1, Tensor idx2) -> int")
    lib.define("kabsch(Tensor src, Tensor corr, Tensor? weights) -> (Tensor, Tensor)")
    lib.impl("chamfer_forward", chamfer_forward, "CUDA")
    lib.impl("chamfer_backward", chamfer_backward, "CUDA")
    lib.impl("kabsch", kabsch, "CUDA")
    register_torch_ops._lib = lib      # keep alive
    _registered = True
