"""Diagnostic: sub-tile selectivity of the pruned search (needs `make -C houv_amd/csrc stamps`)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["HOUV_HIP_LIB"] = os.environ.get("HOUV_STAMPS_LIB", os.path.join(ROOT, "houv_amd", "lib", "libhouv_hip_stamps.so"))
sys.path.insert(0, ROOT)
import numpy as np, torch
from houv_amd import _lib, solver, synthetic
dev = torch.device("cuda:0")
P, K, N = 16, 64, 2048
src, tgt, _ = synthetic.make_pairs(P, N, seed=1)
src, tgt = src.to(dev), tgt.to(dev)
p0 = solver.houv_init_params(P * K)
lib = _lib.load()
buf = (ctypes.c_ulonglong * 8)()
for views in (True, False):
    lib.houv_debug_read_prune_stats(buf, 1)
    for (a, b) in ((0, 10), (10, 50), (50, 200)):
        pass
    solver.run_stage(src, tgt, p0, K, 200, angle_base=0, trans_mode=0, use_views=views, f64_params=False, lr=0.01, pruned=True)
    torch.cuda.synchronize()
    lib.houv_debug_read_prune_stats(buf, 1)
    asked, steps, slots = buf[0], buf[1], buf[2]
    print(f"views={views}: sub-tile visits asked per lane and sweep {asked/(slots*64):.1f}; wave steps per sweep {steps/slots:.1f}; "
          f"wall kcycles per wave-sweep: bounds {buf[3]/slots/1e3:.1f}, masks {buf[4]/slots/1e3:.1f}, walk {buf[5]/slots/1e3:.1f}; "
          f"per-list wave union (64 consecutive queries) {buf[6]/(slots*4):.1f} sub-tiles (sum over the 4 lists {buf[6]/slots:.1f}), "
          f"union over the wave's 256 queries {buf[7]/slots:.1f}")
