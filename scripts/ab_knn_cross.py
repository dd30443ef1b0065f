"""houv_knn_cross (three_nn generalised, k = 1 / 3 / 8): time and output checksums; HOUV_HIP_LIB selects the build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from houv_amd import _lib


def knn_cross(a, b, k):
    B, N, _ = a.shape; M = b.shape[1]
    d2 = torch.empty((B, N, k), dtype=torch.float32, device=a.device); idx = torch.empty((B, N, k), dtype=torch.int32, device=a.device)
    ok = _lib.load().houv_knn_cross(_lib.ptr(a), _lib.ptr(b), B, N, M, k, _lib.ptr(d2), _lib.ptr(idx), _lib.stream_of(a))
    _lib.check(ok, 'houv_knn_cross')
    return d2, idx


dev = torch.device("cuda:0")
torch.manual_seed(0)
out = []
for k in (1, 3, 8):
    for lattice in (False, True):
        a = torch.rand(64, 2048, 3, device=dev); b = torch.rand(64, 1900, 3, device=dev)
        if lattice: a = torch.round(a * 5) / 5; b = torch.round(b * 5) / 5
        for _ in range(3): d, i = knn_cross(a, b, k)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): d, i = knn_cross(a, b, k)
        e1.record(); torch.cuda.synchronize()
        w = torch.arange(1, k + 1, device=dev, dtype=torch.int64)
        out.append(f"k={k}{' lattice' if lattice else ''} {e0.elapsed_time(e1) / 10 * 1e3:.1f} us chk {int((i.long() * w).sum())} {float(d.double().sum()):.6f}")
print(os.path.basename(os.environ.get("HOUV_HIP_LIB", "libhouv_hip.so")), " | ".join(out), flush=True)
