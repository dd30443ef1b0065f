"""Fused-loop time per hypothesis-iteration across cloud sizes (VALU floor = 2*N*M/64 * 26.75 clk / (1024 SIMDs * 2.2 GHz))."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from houv_amd import solver, synthetic
dev = torch.device("cuda:0")
K = 64
for N in (128, 256, 512, 768, 1024, 1536, 2048, 3072, 4096):
    P = max(8, min(256, int(32 * (2048 / N) ** 2)))
    src, tgt, _ = synthetic.make_pairs(P, N, seed=1)
    src, tgt = src.to(dev), tgt.to(dev)
    p0 = solver.houv_init_params(P * K)
    for views in (True, False):
        us = {}
        for pruned in (False, True):
            if pruned and N > solver.PRUNED_MAX_POINTS:
                continue
            f = lambda: solver.run_stage(src, tgt, p0, K, 50, angle_base=0, trans_mode=0, use_views=views, f64_params=False, lr=0.01,
                                         pruned=pruned)
            f(); torch.cuda.synchronize()
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); f(); b.record(); torch.cuda.synchronize()
            us[pruned] = a.elapsed_time(b) * 1e3 / (P * K * 50)
        clk = 26.75 if views else 7.2 * 2      # 4-metric vs single-metric sweep cost per point pair and wave (clk)
        floor = 2.0 * N * N / 64 * clk / (1024 * 2.2e9) * 1e6
        pr = f"   pruned {us[True]:8.4f} us ({floor / us[True]:5.2f} of the brute-force floor)" if True in us else ""
        print(f"N={N:5d} P={P:4d} views={views!s:5s}: brute {us[False]:8.4f} us/hyp-iter   sweep floor {floor:8.4f}   ratio {floor/us[False]:5.2f}{pr}", flush=True)
