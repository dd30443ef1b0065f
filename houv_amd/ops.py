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


def solve_iterate_out(src, tgt, state, K, steps_done, n_iters, angle_base, trans_mode, use_views, f64_params, k_full,
                      k_view, lr, beta1, beta2, eps, loss_scale, out_score, out_loss, out_R, out_T, out_grad=None,
                      out_cd=None, nn_ws=None, ws_valid=0):
    """The C-ABI call itself, caller-allocated outputs (``houv_solve_iterate`` / ``houv_solve_iterate_pruned`` when a
    workspace is given): what ``torch.ops.houv.solve_iterate[_pruned]`` bind.  ``state`` [P*K,24] fp64 (and ``nn_ws``) are
    updated in place; the out_* tensors receive the LAST forward's values.  Returns 1."""
    _lib.require_gpu(src, tgt, state, out_score, out_loss, out_R, out_T, out_grad, out_cd, nn_ws)
    _want(src, _F32, "src"); _want(tgt, _F32, "tgt"); _want(state, _F64, "state")
    P, N, _ = src.shape
    M = tgt.shape[1]
    if tgt.shape[0] != P or src.shape[2] != 3 or tgt.shape[2] != 3:
        raise _lib.HouvHipError("solve_iterate: expected src[P,N,3], tgt[P,M,3]")
    n = P * int(K)
    if tuple(state.shape) != (n, 24):
        raise _lib.HouvHipError(f"solve_iterate: state must be [{n},24], got {tuple(state.shape)}")
    for t, numel, name in ((out_score, n, "out_score"), (out_loss, n, "out_loss"), (out_R, 9 * n, "out_R"),
                           (out_T, 3 * n, "out_T"), (out_grad, 8 * n, "out_grad"), (out_cd, 8 * n, "out_cd")):
        if t is not None:
            _want(t, _F32, name)
            if t.numel() != numel:
                raise _lib.HouvHipError(f"solve_iterate: {name} must hold {numel} floats, got {t.numel()}")
    common = (_lib.ptr(src), _lib.ptr(tgt), P, N, M, int(K), _lib.ptr(state), int(steps_done), int(n_iters),
              int(angle_base), int(trans_mode), int(bool(use_views)), int(bool(f64_params)), int(k_full), int(k_view),
              float(lr), float(beta1), float(beta2), float(eps), float(loss_scale), _lib.ptr(out_score),
              _lib.ptr(out_loss), _lib.ptr(out_R), _lib.ptr(out_T), _lib.ptr(out_grad), _lib.ptr(out_cd))
    with torch.cuda.device(src.device):
        if nn_ws is None:
            ok = _lib.load().houv_solve_iterate(*common, _lib.stream_of(src))
        else:   # exact pruned search: nn_ws int16 [P*K, 16, stride] (solve_workspace) persists between chunked launches
            if nn_ws.dtype != torch.int16 or nn_ws.dim() != 3 or tuple(nn_ws.shape[:2]) != (n, 16) or not nn_ws.is_contiguous():
                raise _lib.HouvHipError("solve_iterate: nn_ws must be a contiguous int16 [P*K,16,stride] tensor (ops.solve_workspace)")
            ok = _lib.load().houv_solve_iterate_pruned(*common, _lib.ptr(nn_ws), int(ws_valid), nn_ws.shape[2],
                                                       _lib.stream_of(src))
    _lib.check(ok, "houv_solve_iterate" + ("_pruned" if nn_ws is not None else ""))
    return 1


def solve_workspace(n_hypotheses, N, M, device):
    """Workspace of houv_solve_iterate_pruned for n hypotheses on clouds of N and M points: int16 [n, 16, stride] (rows 0..7
    the remembered nearest neighbours per direction and metric, rows 8..15 scratch), stride = max(N, M) rounded up to 8."""
    stride = (max(int(N), int(M)) + 7) // 8 * 8
    return torch.empty((int(n_hypotheses), 16, stride), dtype=torch.int16, device=device)


def solve_iterate(src, tgt, state, K, *, steps_done, n_iters, angle_base, trans_mode, use_views, f64_params, k_full,
                  k_view, lr, loss_scale, betas=(0.9, 0.999), eps=1e-8, want_grad=False, want_cd=False, nn_ws=None,
                  ws_valid=False):
    """One launch of the fused HOUV loop (houv_solve_iterate).  ``state`` [P*K,24] fp64 is updated in place.
    Returns dict(score[P*K], loss[P*K], R[P*K,3,3], T[P*K,3][, grad[P*K,8]][, cd[P*K,8]]) of the LAST forward."""
    _lib.require_gpu(src, tgt, state)
    dev = src.device
    n = src.shape[0] * int(K)
    out = dict(score=torch.empty(n, dtype=_F32, device=dev), loss=torch.empty(n, dtype=_F32, device=dev),
               R=torch.empty((n, 3, 3), dtype=_F32, device=dev), T=torch.empty((n, 3), dtype=_F32, device=dev))
    if want_grad:
        out["grad"] = torch.empty((n, 8), dtype=_F32, device=dev)
    if want_cd:
        out["cd"] = torch.empty((n, 8), dtype=_F32, device=dev)
    solve_iterate_out(src, tgt, state, K, steps_done, n_iters, angle_base, trans_mode, use_views, f64_params, k_full,
                      k_view, lr, betas[0], betas[1], eps, loss_scale, out["score"], out["loss"], out["R"], out["T"],
                      out.get("grad"), out.get("cd"), nn_ws, -1 if ws_valid == "verify" else int(bool(ws_valid)))
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
    # the fused loop: the C ABI's argument list, mutable tensors marked; returns 1 like the C call
    solve_args = ("Tensor src, Tensor tgt, Tensor(a!) state, int K, int steps_done, int n_iters, int angle_base, "
                  "int trans_mode, bool use_views, bool f64_params, int k_full, int k_view, float lr, float beta1, "
                  "float beta2, float eps, float loss_scale, Tensor(b!) out_score, Tensor(c!) out_loss, Tensor(d!) out_R, "
                  "Tensor(e!) out_T, Tensor(f!)? out_grad, Tensor(g!)? out_cd")
    lib.define(f"solve_iterate({solve_args}) -> int")
    lib.define(f"solve_iterate_pruned({solve_args}, Tensor(h!) nn_ws, int ws_valid) -> int")
    lib.define("icp_refine(Tensor src, Tensor tgt, Tensor? init, float max_correspondence_distance, int max_iteration, "
               "float relative_fitness, float relative_rmse) -> (Tensor, Tensor, Tensor, Tensor)")
    lib.define("pose_forward(Tensor params, int angle_base, int trans_mode, Tensor? src) -> (Tensor, Tensor, Tensor)")
    lib.impl("chamfer_forward", chamfer_forward, "CUDA")
    lib.impl("chamfer_backward", chamfer_backward, "CUDA")
    lib.impl("kabsch", kabsch, "CUDA")
    lib.impl("solve_iterate", solve_iterate_out, "CUDA")
    lib.impl("solve_iterate_pruned", solve_iterate_out, "CUDA")

    def _icp(src, tgt, init, max_correspondence_distance, max_iteration, relative_fitness, relative_rmse):
        r = icp_refine(src, tgt, init, max_correspondence_distance, max_iteration, relative_fitness, relative_rmse)
        return r["T"], r["fitness"], r["inlier_rmse"], r["iterations"]

    def _pose(params, angle_base, trans_mode, src):
        r = pose_forward(params, angle_base, trans_mode, src)
        return (r[0], r[1], r[2] if src is not None else params.new_empty((0,)))
    lib.impl("icp_refine", _icp, "CUDA")
    lib.impl("pose_forward", _pose, "CUDA")
    register_torch_ops._lib = lib      # keep alive
    _registered = True


# ---------------------------------------------------------------------------------------------------
# DCP feature-head building blocks (houv_knn, houv_edgeconv1, houv_max_over_k, houv_gemm_f32, houv_layernorm,
# houv_softmax_rows, houv_softmax_corr): thin wrappers; shapes are checked, buffers are allocated with torch.
# ---------------------------------------------------------------------------------------------------
def knn(xyz, k):
    """xyz[B,N,3] -> idx[B,N,k] int32, nearest first, self included (dcp.py:35-42)."""
    _lib.require_gpu(xyz); _want(xyz, _F32, "xyz")
    if xyz.dim() != 3 or xyz.shape[2] != 3:
        raise _lib.HouvHipError("knn: expected xyz[B,N,3]")
    B, N, _ = xyz.shape
    idx = torch.empty((B, N, k), dtype=_I32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        ok = _lib.load().houv_knn(_lib.ptr(xyz), B, N, int(k), _lib.ptr(idx), _lib.stream_of(xyz))
    _lib.check(ok, "houv_knn")
    return idx


def edgeconv1(xyz, idx, W, scale, shift):
    """-> act[B*N*k, 64] = relu(scale * conv1(cat(neighbour, centre)) + shift)."""
    _lib.require_gpu(xyz, idx, W, scale, shift)
    for t, d, n in ((xyz, _F32, "xyz"), (idx, _I32, "idx"), (W, _F32, "W"), (scale, _F32, "scale"), (shift, _F32, "shift")):
        _want(t, d, n)
    if idx.dim() != 3 or xyz.dim() != 3 or xyz.shape[2] != 3 or idx.shape[:2] != xyz.shape[:2]:
        raise _lib.HouvHipError("edgeconv1: expected xyz[B,N,3], idx[B,N,k]")
    if tuple(W.shape) != (64, 6) or scale.numel() != 64 or shift.numel() != 64:
        raise _lib.HouvHipError("edgeconv1: expected W[64,6], scale[64], shift[64]")
    B, N, k = idx.shape
    out = torch.empty((B * N * k, 64), dtype=_F32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        ok = _lib.load().houv_edgeconv1(_lib.ptr(xyz), _lib.ptr(idx), B, N, k, _lib.ptr(W), _lib.ptr(scale),
                                        _lib.ptr(shift), _lib.ptr(out), _lib.stream_of(xyz))
    _lib.check(ok, "houv_edgeconv1")
    return out


def max_over_k(act, k, out, col0):
    """out[:, col0:col0+C] = max over groups of k consecutive rows of act[npts*k, C]."""
    _lib.require_gpu(act); _lib.require_gpu_any(out)
    _want(act, _F32, "act"); _want(out, _F32, "out")
    C = act.shape[1]
    if act.dim() != 2 or act.shape[0] % k or C % 4 or out.stride(0) % 4 or out.stride(1) != 1 \
            or out.shape[0] != act.shape[0] // k or col0 + C > out.shape[1]:
        raise _lib.HouvHipError("max_over_k: expected act[npts*k,C], out[npts, >= col0+C] (C and the row stride multiples of 4)")
    npts = act.shape[0] // k
    view = out[:, col0:col0 + C]
    with torch.cuda.device(act.device):
        ok = _lib.load().houv_max_over_k(_lib.ptr(act), npts, int(k), C, ctypes_ptr(view), out.stride(0),
                                         _lib.stream_of(act))
    _lib.check(ok, "houv_max_over_k")


def ctypes_ptr(t):
    import ctypes
    return ctypes.c_void_p(t.data_ptr())


# bench.py sets this to a list to collect (start_event, end_event, flops) per houv_gemm_f32 launch
GEMM_LOG = None


def gemm(A, B, C=None, *, trans_b=True, alpha=1.0, scale=None, shift=None, residual=None, relu=False):
    """2-D or batched (3-/4-D leading dims = (outer[, inner])) fp32 GEMM on the MFMA kernel.
    A[..., M, K]; B[..., N, K] if trans_b else B[..., K, N]; returns C[..., M, N].  Inner-most dims may be strided
    row-major views (row stride = lda), which is how per-head attention operands are addressed without copies."""
    _lib.require_gpu_any(A, B, C, scale, shift, residual)
    lead = A.shape[:-2]
    M, K = A.shape[-2:]
    N = B.shape[-2] if trans_b else B.shape[-1]
    if C is None:
        C = torch.empty(lead + (M, N), dtype=_F32, device=A.device)
    outer = lead[0] if len(lead) >= 1 else 1
    inner = lead[1] if len(lead) == 2 else 1

    def strides(t):
        ld = t.stride(-2)
        assert t.stride(-1) == 1, "innermost dimension must be contiguous"
        so = t.stride(0) if len(lead) >= 1 else 0
        si = t.stride(1) if len(lead) == 2 else 0
        return ld, so, si
    lda, sAo, sAi = strides(A)
    ldb, sBo, sBi = strides(B)
    ldc, sCo, sCi = strides(C)
    ldr, sRo, sRi = strides(residual) if residual is not None else (0, 0, 0)
    if GEMM_LOG is not None:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(A.device))
    with torch.cuda.device(A.device):
        ok = _lib.load().houv_gemm_f32(ctypes_ptr(A), ctypes_ptr(B), ctypes_ptr(C), M, N, K, lda, ldb, ldc,
                                       int(bool(trans_b)), outer, inner, sAo, sAi, sBo, sBi, sCo, sCi, float(alpha),
                                       _lib.ptr(scale), _lib.ptr(shift),
                                       None if residual is None else ctypes_ptr(residual), ldr, sRo, sRi,
                                       int(bool(relu)), _lib.stream_of(A))
    _lib.check(ok, "houv_gemm_f32")
    if GEMM_LOG is not None:
        ev1.record(torch.cuda.current_stream(A.device))
        GEMM_LOG.append((ev0, ev1, 2.0 * outer * inner * M * N * K))
    return C


# set to False to route models.dcp's attention through gemm -> softmax_rows -> gemm (the [P,H,Nq,Nk] scores materialised)
FUSED_ATTENTION = True


def attention(Q, K, V, scale):
    """Fused multi-head attention (dcp.py:26-32): Q[P,Nq,H,128], K/V[P,Nk,H,128] (views of token-major [P,N,H*128] buffers)
    -> context[P,Nq,H,128] = softmax(Q K^T * scale) V per head; the scores never reach HBM (houv_attention_f32)."""
    _lib.require_gpu_any(Q, K, V)
    P, Nq, H, dk = Q.shape
    Nk = K.shape[1]
    if dk != 128 or tuple(K.shape) != (P, Nk, H, dk) or tuple(V.shape) != (P, Nk, H, dk):
        raise _lib.HouvHipError("attention: expected Q[P,Nq,H,128], K[P,Nk,H,128], V[P,Nk,H,128]")
    for t in (Q, K, V):
        if t.stride(3) != 1 or t.stride(2) != dk:
            raise _lib.HouvHipError("attention: heads must be contiguous 128-float slices of a token row")
    out = torch.empty((P, Nq, H, dk), dtype=_F32, device=Q.device)
    if GEMM_LOG is not None:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(Q.device))
    with torch.cuda.device(Q.device):
        ok = _lib.load().houv_attention_f32(ctypes_ptr(Q), ctypes_ptr(K), ctypes_ptr(V), ctypes_ptr(out), P, H, Nq, Nk, dk,
                                            Q.stride(1), K.stride(1), V.stride(1), out.stride(1), Q.stride(0) if P else 0,
                                            K.stride(0) if P else 0, V.stride(0) if P else 0, out.stride(0) if P else 0,
                                            float(scale), _lib.stream_of(Q))
    _lib.check(ok, "houv_attention_f32")
    if GEMM_LOG is not None:
        ev1.record(torch.cuda.current_stream(Q.device))
        GEMM_LOG.append((ev0, ev1, 4.0 * P * H * Nq * Nk * dk))
    return out


def layernorm(x, a, b, eps=1e-6, residual=None):
    _lib.require_gpu(x, a, b, residual)
    for t, n in ((x, "x"), (a, "a"), (b, "b"), (residual, "residual")):
        if t is not None:
            _want(t, _F32, n)
    D = x.shape[-1]
    if a.numel() != D or b.numel() != D or (residual is not None and residual.shape != x.shape):
        raise _lib.HouvHipError("layernorm: a, b must have D elements and residual the shape of x")
    rows = x.numel() // D
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        ok = _lib.load().houv_layernorm(_lib.ptr(x), rows, D, _lib.ptr(a), _lib.ptr(b), float(eps), _lib.ptr(residual),
                                        _lib.ptr(out), _lib.stream_of(x))
    _lib.check(ok, "houv_layernorm")
    return out


def softmax_rows_(x):
    _lib.require_gpu(x); _want(x, _F32, "x")
    L = x.shape[-1]
    with torch.cuda.device(x.device):
        ok = _lib.load().houv_softmax_rows(_lib.ptr(x), x.numel() // L, L, _lib.stream_of(x))
    _lib.check(ok, "houv_softmax_rows")
    return x


def softmax_corr(scores, pts):
    """scores[P,N,M], pts[P,M,3] -> corr[P,3,N] = pts^T softmax(scores)^T."""
    _lib.require_gpu(scores, pts); _want(scores, _F32, "scores"); _want(pts, _F32, "pts")
    if scores.dim() != 3 or tuple(pts.shape) != (scores.shape[0], scores.shape[2], 3):
        raise _lib.HouvHipError("softmax_corr: expected scores[P,N,M], pts[P,M,3]")
    P, N, M = scores.shape
    corr = torch.empty((P, 3, N), dtype=_F32, device=scores.device)
    with torch.cuda.device(scores.device):
        ok = _lib.load().houv_softmax_corr(_lib.ptr(scores), P, N, M, _lib.ptr(pts), _lib.ptr(corr), _lib.stream_of(scores))
    _lib.check(ok, "houv_softmax_corr")
    return corr
