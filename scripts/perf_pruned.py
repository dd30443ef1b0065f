import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from houv_amd import solver, synthetic
dev = torch.device("cuda:0")
P, K, N = int(os.environ.get("P", 32)), 64, 2048
src, tgt, _ = synthetic.make_pairs(P, N, seed=1)
src, tgt = solver.spatial_sort(src.to(dev)), solver.spatial_sort(tgt.to(dev))
p0 = solver.houv_init_params(P * K)
def timed(fn, n=2):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
for iters in (20, 200):
    for views in (True, False):
        for pruned in (False, True):
            t = timed(lambda: solver.run_stage(src, tgt, p0, K, iters, angle_base=0, trans_mode=0, use_views=views,
                                               f64_params=False, lr=0.01, pruned=pruned), n=1 if iters > 50 else 2)
            print(f"iters={iters:3d} views={views!s:5s} pruned={pruned!s:5s}: {t:8.1f} ms  {t*1e3/(P*K*iters):.3f} us/hyp-iter", flush=True)
