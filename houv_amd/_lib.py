"""ctypes binding of libhouv_hip.so (include/houv_hip.h).

There is NO fallback: if the HIP library is missing or a call fails, this raises.  The oracle under
``oracle/`` is never imported from here.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# HOUV_HIP_LIB lets the diagnostic scripts load the stamped build variant; the default is the product library
LIB_PATH = os.environ.get("HOUV_HIP_LIB") or os.path.join(_HERE, "lib", "libhouv_hip.so")
ABI_VERSION = 2

_c_f = ctypes.c_void_p     # device pointers travel as plain addresses
_int = ctypes.c_int
_dbl = ctypes.c_double
_flt = ctypes.c_float

_SIGNATURES = {
    "houv_abi_version": (ctypes.c_int, []),
    "houv_last_error": (ctypes.c_char_p, []),
    "houv_build_id": (ctypes.c_char_p, []),
    "houv_debug_set": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_longlong]),
    "houv_chamfer_forward": (ctypes.c_int, [_c_f, _c_f, _int, _int, _int, _c_f, _c_f, _c_f, _c_f, _c_f]),
    "houv_chamfer_backward": (ctypes.c_int, [_c_f, _c_f, _int, _int, _int, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f]),
    "houv_kabsch": (ctypes.c_int, [_c_f, _c_f, _c_f, _int, _int, _c_f, _c_f, _c_f]),
    "houv_solve_iterate": (ctypes.c_int, [_c_f, _c_f, _int, _int, _int, _int, _c_f, _int, _int, _int, _int, _int, _int,
                                          _int, _int, _dbl, _dbl, _dbl, _dbl, _flt, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f,
                                          _c_f]),
    "houv_solve_iterate_pruned": (ctypes.c_int, [_c_f, _c_f, _int, _int, _int, _int, _c_f, _int, _int, _int, _int, _int,
                                                 _int, _int, _int, _dbl, _dbl, _dbl, _dbl, _flt, _c_f, _c_f, _c_f, _c_f,
                                                 _c_f, _c_f, _c_f, _int, _int, _c_f]),
    "houv_solve_variant": (ctypes.c_int, [_int, _int, _int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                          ctypes.POINTER(ctypes.c_int)]),
    "houv_icp_refine": (ctypes.c_int, [_c_f, _c_f, _int, _int, _int, _c_f, _flt, _int, _flt, _flt, _c_f, _c_f, _c_f, _c_f, _c_f]),
    "houv_knn": (ctypes.c_int, [_c_f, _int, _int, _int, _c_f, _c_f]),
    "houv_edgeconv1": (ctypes.c_int, [_c_f, _c_f, _int, _int, _int, _c_f, _c_f, _c_f, _c_f, _c_f]),
    "houv_max_over_k": (ctypes.c_int, [_c_f, ctypes.c_longlong, _int, _int, _c_f, _int, _c_f]),
    "houv_gemm_f32": (ctypes.c_int, [_c_f, _c_f, _c_f, _int, _int, _int, _int, _int, _int, _int, _int, _int] +
                      [ctypes.c_longlong] * 6 + [_flt, _c_f, _c_f, _c_f, _int, ctypes.c_longlong, ctypes.c_longlong,
                                                 _int, _c_f]),
    "houv_attention_f32": (ctypes.c_int, [_c_f, _c_f, _c_f, _c_f, _int, _int, _int, _int, _int, _int, _int, _int, _int] +
                           [ctypes.c_longlong] * 4 + [_flt, _c_f]),
    "houv_layernorm": (ctypes.c_int, [_c_f, ctypes.c_longlong, _int, _c_f, _c_f, _flt, _c_f, _c_f, _c_f]),
    "houv_softmax_rows": (ctypes.c_int, [_c_f, ctypes.c_longlong, _int, _c_f]),
    "houv_softmax_corr": (ctypes.c_int, [_c_f, _int, _int, _int, _c_f, _c_f, _c_f]),
    "houv_furthest_point_sample": (ctypes.c_int, [_c_f, _int, _int, _int, _c_f, _c_f]),
    "houv_knn_cross": (ctypes.c_int, [_c_f, _c_f, _int, _int, _int, _int, _c_f, _c_f, _c_f]),
    "houv_gather_points": (ctypes.c_int, [_c_f, _c_f, _int, _int, _int, _int, _c_f, _c_f]),
    "houv_pose_forward": (ctypes.c_int, [_c_f, _int, _int, _int, _c_f, _int, _c_f, _c_f, _c_f, _c_f]),
}

_lib = None


class HouvHipError(RuntimeError):
    pass


def exported_symbols():
    return sorted(_SIGNATURES)


def load():
    """Load libhouv_hip.so once; raise (never fall back) when it is absent or ABI-mismatched."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HouvHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C houv_amd/csrc`). houv_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = symbol missing from the build
        fn.restype = res
        fn.argtypes = args
    if lib.houv_abi_version() != ABI_VERSION:
        raise HouvHipError(f"libhouv_hip.so ABI {lib.houv_abi_version()} != expected {ABI_VERSION}")
    _lib = lib
    return lib


def solve_variant(N, M, pruned=False, with_mode=False):
    """(threads per workgroup, points per lane[, prune mode]) of the solve_kernel that serves clouds of N and M points."""
    b, q, m = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    check(load().houv_solve_variant(int(N), int(M), int(bool(pruned)), ctypes.byref(b), ctypes.byref(q), ctypes.byref(m)),
          "houv_solve_variant")
    return (b.value, q.value, m.value) if with_mode else (b.value, q.value)


def build_id():
    """Hash of the sources the loaded library was built from (houv_build_id)."""
    return load().houv_build_id().decode()


def debug_set(name, value):
    """Diagnostic switch of the library (houv_debug_set; see houv_amd/csrc/houv_common.h `DebugKnobs`)."""
    check(load().houv_debug_set(name.encode(), int(value)), "houv_debug_set")


def last_error():
    return load().houv_last_error().decode("utf-8", "replace")


def check(ok, what):
    if ok != 1:
        raise HouvHipError(f"{what} failed: {last_error()}")


def ptr(t):
    """Device address of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_of(t):
    """The torch current stream of t's device as a raw hipStream_t."""
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def require_gpu(*tensors, dtype=None):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise HouvHipError("houv_amd ops run on an MI355X only: got a CPU tensor (there is no CPU fallback)")
        if not t.is_contiguous():
            raise HouvHipError("houv_amd ops need contiguous tensors")
    dev = [t.device for t in tensors if t is not None]
    if any(d != dev[0] for d in dev):
        raise HouvHipError("all tensors must live on the same device")


def require_gpu_any(*tensors):
    """Like require_gpu but allows strided (row-major, inner-contiguous) views."""
    ts = [t for t in tensors if t is not None]
    for t in ts:
        if not t.is_cuda:
            raise HouvHipError("houv_amd ops run on an MI355X only: got a CPU tensor (there is no CPU fallback)")
        if t.dtype != torch.float32:
            raise HouvHipError(f"expected float32, got {t.dtype}")
    if any(t.device != ts[0].device for t in ts):
        raise HouvHipError("all tensors must live on the same device")
