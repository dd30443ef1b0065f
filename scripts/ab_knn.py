"""k-NN (houv_knn, k = 20, the DCP head's graph feature): time and output checksum; HOUV_HIP_LIB selects the build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from houv_amd import ops, synthetic
dev = torch.device("cuda:0")
torch.manual_seed(0)
out = []
for name, B, N in (("uniform", 64, 2048), ("mvp", 64, 2048), ("uniform", 64, 777), ("lattice", 16, 2048)):
    if name == "uniform": x = torch.rand(B, N, 3, device=dev)
    elif name == "lattice": x = torch.round(torch.rand(B, N, 3, device=dev) * 6) / 6      # many exact ties
    else:
        s, t, _ = synthetic.make_pairs(B, N, seed=3); x = s.to(dev).contiguous()
    for _ in range(3): idx = ops.knn(x, 20)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): idx = ops.knn(x, 20)
    e1.record(); torch.cuda.synchronize()
    w = torch.arange(1, 21, device=dev, dtype=torch.int64)
    out.append(f"{name}:{B}x{N} {e0.elapsed_time(e1) / 10 * 1e3:.1f} us chk {int((idx.long() * w).sum())}")
print(os.path.basename(os.environ.get("HOUV_HIP_LIB", "libhouv_hip.so")), " | ".join(out), flush=True)
