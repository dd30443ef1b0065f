"""Same-device A/B of the spatial sort feeding the pruned search: Morton curve (rounds 1-2) vs balanced k-d leaves (round 3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from houv_amd import _lib, solver, synthetic
dev = torch.device("cuda:0")
P, K, N, iters = int(os.environ.get("P", 128)), 64, int(os.environ.get("N", 2048)), int(os.environ.get("ITERS", 50))
src0, tgt0, _ = synthetic.make_pairs(P, N, seed=1)
p0 = solver.houv_init_params(P * K)
for views in (True, False):
    for mode in ("morton", "kd", "kd-area"):
        solver.SPATIAL_SORT = "kd" if mode.startswith("kd") else "morton"
        solver.KD_RULE = "area" if mode.endswith("area") else "extent"
        solver._SORTED.clear()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        x = src0.to(dev)
        solver.spatial_sort(x.clone()); e0.record(); solver.spatial_sort(x.clone()); e1.record(); torch.cuda.synchronize()
        sort_ms = e0.elapsed_time(e1)
        src, tgt = solver.spatial_sort(src0.to(dev)), solver.spatial_sort(tgt0.to(dev))
        def run(pruned=True):
            return solver.run_stage(src, tgt, p0, K, 150, angle_base=0, trans_mode=0 if views else 1, use_views=views,
                                    f64_params=not views, lr=0.01, pruned=pruned, iters_per_launch=iters)
        run(); torch.cuda.synchronize()
        ts = []
        for _ in range(2):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); o, st = run(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ob, stb = run(pruned=False)
        same = torch.equal(st, stb) and torch.equal(o["score"], ob["score"])
        buf = torch.zeros(8, dtype=torch.int64, device=dev)
        _lib.debug_set("solve_stats", buf.data_ptr()); run(); torch.cuda.synchronize(); _lib.debug_set("solve_stats", 0)
        v = [int(x) for x in buf.cpu()]
        print(f"views={views!s:5s} sort={mode:7s}: {min(ts) * 1e3 / (P * K * 150):.4f} us/hyp-iter  bit-identical to brute force on the same clouds: {same}  "
              f"asked/query {v[0] / max(v[2], 1) / 64 / 4:.2f}  sort of one [{P},{N},3] cloud {sort_ms:.1f} ms  best score mean {float(o['score'].reshape(P, K).min(1)[0].mean()):.5f}", flush=True)
