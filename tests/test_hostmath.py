"""CPU checks of houv_amd/csrc/houv_math.h (the per-instance math the HIP kernels inline) against the oracle."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import houv_ref_cpu as orc
from tests import hostmath

F = ctypes.POINTER(ctypes.c_float)
D = ctypes.POINTER(ctypes.c_double)


def fp(a):
    return a.ctypes.data_as(F)


def dp(a):
    return a.ctypes.data_as(D)


@pytest.fixture(scope="module")
def hm():
    return hostmath.load()


def pack(V, a, tc, ts):
    return np.ascontiguousarray(np.concatenate([V, a, tc, ts], axis=1), dtype=np.float32)


@pytest.mark.parametrize("mode", ["houv", "solve"])
@pytest.mark.parametrize("base", [0, 1, 2, 3])
def test_pose_forward_and_backward(hm, base, mode):
    n = 64
    V, a, tc, ts = orc.houv_init_params(n, seed=5)
    P = pack(V, a, tc, ts)
    R = np.zeros((n, 9), np.float32)
    T = np.zeros((n, 3), np.float32)
    tm = 0 if mode == "houv" else 1
    hm.hm_pose_forward(fp(P), n, base, tm, fp(R), fp(T))
    tv = [torch.tensor(x, requires_grad=True) for x in (V, a, tc, ts)]
    src = torch.tensor(np.random.default_rng(0).standard_normal((n, 17, 3)).astype(np.float32))
    moved, Rr, Tr = orc.houv_forward(src, *tv, base, mode)
    np.testing.assert_allclose(R.reshape(n, 3, 3), Rr.detach().numpy(), atol=3e-7)
    np.testing.assert_allclose(T, Tr.detach().numpy()[:, 0], atol=1e-7)
    # backward: random upstream gradient G on the moved cloud
    G = torch.tensor(np.random.default_rng(1).standard_normal((n, 17, 3)).astype(np.float32))
    (moved * G).sum().backward()
    gT = G.sum(1).numpy().astype(np.float32)
    M = torch.einsum("bni,bnj->bij", G, src).numpy().astype(np.float32).reshape(n, 9)
    g = np.zeros((n, 8), np.float32)
    hm.hm_pose_backward(fp(P), n, base, tm, fp(np.ascontiguousarray(gT)), fp(np.ascontiguousarray(M)), fp(g))
    ref = np.concatenate([t.grad.numpy() for t in tv], axis=1)
    scale = np.abs(ref).max(axis=0, keepdims=True) + 1e-6
    np.testing.assert_allclose(g / scale, ref / scale, atol=2e-5)


def test_adam_matches_torch_f32_and_f64(hm):
    rng = np.random.default_rng(3)
    for dt, fn, cp in ((np.float32, "hm_adam_f32", fp), (np.float64, "hm_adam_f64", dp)):
        n = 50
        p0 = rng.standard_normal(n).astype(dt)
        p = p0.copy(); m = np.zeros(n, dt); v = np.zeros(n, dt)
        tp = torch.tensor(p0.copy(), requires_grad=True)
        opt = torch.optim.Adam([tp], lr=0.01)
        for step in range(1, 30):
            g = (rng.standard_normal(n) * 10.0 ** rng.integers(-6, 1, n)).astype(dt)
            getattr(hm, fn)(cp(p), cp(m), cp(v), cp(g), n, step, ctypes.c_double(0.01), ctypes.c_double(0.9),
                            ctypes.c_double(0.999), ctypes.c_double(1e-8))
            tp.grad = torch.tensor(g.copy())
            opt.step()
            tol = 2e-7 if dt == np.float32 else 1e-15
            np.testing.assert_allclose(p, tp.detach().numpy(), rtol=tol, atol=tol)


def test_svd_and_kabsch(hm, golden):
    rng = np.random.default_rng(9)
    n = 200
    H = rng.standard_normal((n, 3, 3)).astype(np.float32)
    H[:20] *= np.array([1, 1e-3, 1e-6], np.float32)      # ill conditioned
    H[20:30, :, 2] = H[20:30, :, 0]                      # rank deficient
    H[30] = 0
    Hc = np.ascontiguousarray(H.reshape(n, 9))
    U = np.zeros((n, 9), np.float32); S = np.zeros((n, 3), np.float32); V = np.zeros((n, 9), np.float32)
    hm.hm_svd3x3_f32(fp(Hc), n, fp(U), fp(S), fp(V))
    U = U.reshape(n, 3, 3); V = V.reshape(n, 3, 3)
    rec = U @ (S[:, :, None] * V.transpose(0, 2, 1))
    np.testing.assert_allclose(rec, H, atol=2e-6 * max(1.0, np.abs(H).max()))
    np.testing.assert_allclose(S, np.linalg.svd(H.astype(np.float64), compute_uv=False), atol=3e-6)
    eye = np.broadcast_to(np.eye(3, dtype=np.float32), (n, 3, 3))
    np.testing.assert_allclose(V.transpose(0, 2, 1) @ V, eye, atol=2e-6)
    np.testing.assert_allclose(U.transpose(0, 2, 1) @ U, eye, atol=2e-5)
    # Kabsch rotation against the reference's SVDHead outputs (G7): build H the way model_utils.py:221-227 does
    g = golden("g7_svdhead.npz")
    src, corr = g["src"], g["corr"]
    Hk = (src - src.mean(2, keepdims=True)) @ (corr - corr.mean(2, keepdims=True)).transpose(0, 2, 1)
    Hk = np.ascontiguousarray(Hk.reshape(-1, 9).astype(np.float32))
    R = np.zeros((len(Hk), 9), np.float32)
    hm.hm_kabsch_rotation_f32(fp(Hk), len(Hk), fp(R))
    np.testing.assert_allclose(R.reshape(-1, 3, 3), g["R"], atol=2e-5)
