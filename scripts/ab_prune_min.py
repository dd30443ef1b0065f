"""Does the pruned search (k-d leaves, balanced walk) pay below 513 points?  houv_debug_set("prune_min_points") + solver.PRUNED_MIN_POINTS."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from houv_amd import _lib, solver, synthetic
dev = torch.device("cuda:0")
K = 64
for N in [int(x) for x in os.environ.get("SIZES", "512,448,384,320,768").split(",")]:
    P = 256
    src0, tgt0, _ = synthetic.make_pairs(P, N, seed=1)
    src, tgt = solver.spatial_sort(src0.to(dev)), solver.spatial_sort(tgt0.to(dev))
    p0 = solver.houv_init_params(P * K)
    for views in (True, False):
        out = []
        for label, minpts, pruned in (("brute", 2049, False), ("pruned", 257, True)):
            _lib.debug_set("prune_min_points", minpts); solver.PRUNED_MIN_POINTS = minpts
            f = lambda: solver.run_stage(src, tgt, p0, K, 100, angle_base=0, trans_mode=0 if views else 1, use_views=views,
                                         f64_params=not views, lr=0.01, pruned=pruned)
            f(); torch.cuda.synchronize()
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); o, st = f(); b.record(); torch.cuda.synchronize()
            out.append((label, a.elapsed_time(b) * 1e3 / (P * K * 100), st.clone(), _lib.solve_variant(N, N, pruned, with_mode=True)))
        print(f"N={N} views={views!s:5s}: " + "  ".join(f"{l} {t:.4f} us {v}" for l, t, _, v in out) + f"  identical={torch.equal(out[0][2], out[1][2])}", flush=True)
_lib.debug_set("prune_min_points", 257)
