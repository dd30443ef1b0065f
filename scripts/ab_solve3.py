"""Round-3 same-device A/B at BASELINE configs[1]'s shape (or N): brute-force sweep, pruned search with the owner walk
(houv_debug_set("prune_owner_walk")) and with the balanced walk; bit-identity with brute force; counters of the walk."""
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
res = {}
for views in (True, False):
    for label, pruned, legacy in (("brute", False, 0), ("pruned/owner-walk", True, 1), ("pruned/balanced", True, 0)):
        _lib.debug_set("prune_owner_walk", legacy)
        def run():
            return solver.run_stage(src, tgt, p0, K, iters, angle_base=0, trans_mode=0 if views else 1, use_views=views,
                                    f64_params=not views, lr=0.01, pruned=pruned)
        run(); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); o, st = run(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        res[(views, label)] = (o["score"].clone(), st.clone())
        extra = ""
        if pruned:
            buf = torch.zeros(8, dtype=torch.int64, device=dev)
            _lib.debug_set("solve_stats", buf.data_ptr()); run(); torch.cuda.synchronize(); _lib.debug_set("solve_stats", 0)
            v = [int(x) for x in buf.cpu()]
            same = all(torch.equal(x, y) for x, y in zip(res[(views, label)], res[(views, "brute")]))
            extra = (f"  bit-identical to brute: {same}  steps/wave-sweep {v[1] / max(v[2], 1):.1f} "
                     f"asked/query {v[0] / max(v[2], 1) / 64 / _lib.solve_variant(N, N)[1]:.1f} clock {v[4] / max(v[5], 1) * 0.1:.2f} GHz")
        print(f"views={views!s:5s} {label:14s}: {min(ts) * 1e3 / (P * K * iters):.4f} us/hyp-iter{extra}", flush=True)
_lib.debug_set("prune_owner_walk", 0)
