"""ctypes loader of oracle/chamfer_ref.c (TEST INFRASTRUCTURE ONLY: imported by tests/ and __graft_entry__.smoke())."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libhouv_oracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "chamfer_ref.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "all"])
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def chamfer_forward(xyz1, xyz2):
    """numpy fp32 [B,N,3], [B,M,3] -> dist1, dist2 (fp32), idx1, idx2 (int32)."""
    xyz1 = np.ascontiguousarray(xyz1, np.float32)
    xyz2 = np.ascontiguousarray(xyz2, np.float32)
    B, N, _ = xyz1.shape
    M = xyz2.shape[1]
    d1 = np.empty((B, N), np.float32); d2 = np.empty((B, M), np.float32)
    i1 = np.empty((B, N), np.int32); i2 = np.empty((B, M), np.int32)
    assert load().oracle_chamfer_forward(_f(xyz1), _f(xyz2), B, N, M, _f(d1), _f(d2), _i(i1), _i(i2)) == 1
    return d1, d2, i1, i2


def chamfer_backward(xyz1, xyz2, g1, g2, idx1, idx2):
    xyz1 = np.ascontiguousarray(xyz1, np.float32); xyz2 = np.ascontiguousarray(xyz2, np.float32)
    g1 = np.ascontiguousarray(g1, np.float32); g2 = np.ascontiguousarray(g2, np.float32)
    idx1 = np.ascontiguousarray(idx1, np.int32); idx2 = np.ascontiguousarray(idx2, np.int32)
    B, N, _ = xyz1.shape
    M = xyz2.shape[1]
    gx1 = np.zeros_like(xyz1); gx2 = np.zeros_like(xyz2)
    assert load().oracle_chamfer_backward(_f(xyz1), _f(xyz2), B, N, M, _f(g1), _f(g2), _i(idx1), _i(idx2), _f(gx1), _f(gx2)) == 1
    return gx1, gx2
