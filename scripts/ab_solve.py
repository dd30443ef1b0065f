"""One leg of a same-device A/B of the fused loop: HOUV_HIP_LIB selects the library build; prints us per hypothesis-iteration
for the brute-force kernel (views on / off) and the pruned kernel at BASELINE configs[1]'s shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from houv_amd import solver, synthetic
dev = torch.device("cuda:0")
P, K, N, iters = int(os.environ.get("P", 64)), 64, int(os.environ.get("N", 2048)), int(os.environ.get("ITERS", 20))
src, tgt, _ = synthetic.make_pairs(P, N, seed=1)
src, tgt = solver.spatial_sort(src.to(dev)), solver.spatial_sort(tgt.to(dev))
p0 = solver.houv_init_params(P * K)
out = []
for label, views, pruned in (("brute/views", True, False), ("brute/noviews", False, False), ("pruned/views", True, True), ("pruned/noviews", False, True)):
    def run():
        return solver.run_stage(src, tgt, p0, K, iters, angle_base=0, trans_mode=0, use_views=views, f64_params=False, lr=0.01, pruned=pruned)
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); o, st = run(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    out.append(f"{label} {min(ts) * 1e3 / (P * K * iters):.4f}")
    chk = float(o["score"].double().sum())
print(os.path.basename(os.environ.get("HOUV_HIP_LIB", "libhouv_hip.so")), " | ".join(out), f"| checksum {chk:.9f}", flush=True)
