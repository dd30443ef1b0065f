"""Same-device A/B of houv_debug_set("prune_refresh") for the balanced pruned walk at BASELINE configs[1]'s shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from houv_amd import _lib, solver, synthetic
dev = torch.device("cuda:0")
P, K, N, iters = int(os.environ.get("P", 128)), 64, int(os.environ.get("N", 2048)), int(os.environ.get("ITERS", 50))
src, tgt, _ = synthetic.make_pairs(P, N, seed=1)
src, tgt = solver.spatial_sort(src.to(dev)), solver.spatial_sort(tgt.to(dev))
p0 = solver.houv_init_params(P * K)
ref = None
for r in [int(x) for x in os.environ.get("REFRESH", "2,1,3,4,6,8,0").split(",")]:
    _lib.debug_set("prune_refresh", r)
    def run():
        return solver.run_stage(src, tgt, p0, K, 150, angle_base=0, trans_mode=0, use_views=True, f64_params=False, lr=0.01, pruned=True,
                                iters_per_launch=iters)
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(2):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); o, st = run(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    if ref is None:
        ref = st.clone()
    buf = torch.zeros(8, dtype=torch.int64, device=dev)
    _lib.debug_set("solve_stats", buf.data_ptr()); run(); torch.cuda.synchronize(); _lib.debug_set("solve_stats", 0)
    v = [int(x) for x in buf.cpu()]
    print(f"refresh={r}: {min(ts) * 1e3 / (P * K * 150):.4f} us/hyp-iter  identical={torch.equal(st, ref)}  asked/query {v[0] / max(v[2], 1) / 64 / 4:.2f}", flush=True)
_lib.debug_set("prune_refresh", 4)
