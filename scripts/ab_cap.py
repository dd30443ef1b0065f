"""Same-device A/B of the pruned walk's lock-step cap (houv_debug_set("prune_cap_slack")): us per hypothesis-iteration at
BASELINE configs[1]'s shape, plus the bit-identity of every variant with the fused-loop-only walk."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from houv_amd import _lib, solver, synthetic
dev = torch.device("cuda:0")
P, K, N, iters = int(os.environ.get("P", 64)), 64, int(os.environ.get("N", 2048)), int(os.environ.get("ITERS", 50))
src, tgt, _ = synthetic.make_pairs(P, N, seed=1)
src, tgt = solver.spatial_sort(src.to(dev)), solver.spatial_sort(tgt.to(dev))
p0 = solver.houv_init_params(P * K)
ref = None
for views in (True, False):
    for slack in [int(x) for x in os.environ.get("SLACKS", "-1,0,1,2,3,5,64").split(",")]:
        _lib.debug_set("prune_cap_slack", slack)
        def run():
            return solver.run_stage(src, tgt, p0, K, iters, angle_base=0, trans_mode=0 if views else 1, use_views=views,
                                    f64_params=not views, lr=0.01, pruned=True)
        run(); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); o, st = run(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        if slack == -1:
            ref = (o["score"].clone(), st.clone())
        same = ref is not None and torch.equal(o["score"], ref[0]) and torch.equal(st, ref[1])
        buf = torch.zeros(8, dtype=torch.int64, device=dev)
        _lib.debug_set("solve_stats", buf.data_ptr()); run(); torch.cuda.synchronize(); _lib.debug_set("solve_stats", 0)
        v = [int(x) for x in buf.cpu()]
        print(f"views={views!s:5s} slack={slack:3d}: {min(ts) * 1e3 / (P * K * iters):.4f} us/hyp-iter  identical_to_fused={same}  "
              f"steps/sweep {v[1] / max(v[2], 1):.1f} asked/lane {v[0] / max(v[2], 1) / 64:.1f}", flush=True)
_lib.debug_set("prune_cap_slack", 1)
