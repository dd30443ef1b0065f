import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from houv_amd import _lib, ops
dev = torch.device("cuda:0")
x = torch.rand(16, 2048, 3, device=dev)
for mode in (0, 1):
    _lib.debug_set("knn_split", mode)
    for _ in range(3): ops.knn(x, 20)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): ops.knn(x, 20)
    b.record(); torch.cuda.synchronize()
    print(f"knn20 16 x 2048, knn_split={mode}: {a.elapsed_time(b) / 10 * 1e3:.1f} us")
_lib.debug_set("knn_split", 1)
