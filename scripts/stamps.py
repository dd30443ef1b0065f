"""Diagnostic: per-phase wave-cycle shares of solve_kernel from the -DHOUV_STAMPS build (make -C houv_amd/csrc stamps)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["HOUV_HIP_LIB"] = os.environ.get("HOUV_STAMPS_LIB", os.path.join(ROOT, "houv_amd", "lib", "libhouv_hip_stamps.so"))
sys.path.insert(0, ROOT)
import numpy as np, torch
from houv_amd import _lib, solver, synthetic
dev = torch.device("cuda:0")
P, K, N, iters = int(os.environ.get("P", 32)), 64, int(os.environ.get("N", 2048)), int(os.environ.get("ITERS", 10))
src, tgt, _ = synthetic.make_pairs(P, N, seed=1)
src, tgt = src.to(dev), tgt.to(dev)
p0 = solver.houv_init_params(P * K)
solver.PRUNED = bool(int(os.environ.get('PRUNED', '0')))
src, tgt = solver.spatial_sort(src), solver.spatial_sort(tgt)
lib = _lib.load()
if os.environ.get("OWNER_WALK") == "1":
    _lib.debug_set("prune_owner_walk", 1)
NW = _lib.solve_variant(N, N, solver.PRUNED)[0] // 64        # waves per workgroup of the variant that runs
buf = (ctypes.c_ulonglong * 16)()
pbuf = (ctypes.c_ulonglong * 8)()
for views in (True, False):
    solver.run_stage(src, tgt, p0, K, 2, angle_base=0, trans_mode=0, use_views=views, f64_params=False, lr=0.01)
    torch.cuda.synchronize()
    lib.houv_debug_read_stamps(buf, 1)
    lib.houv_debug_read_prune_stats(pbuf, 1)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    solver.run_stage(src, tgt, p0, K, iters, angle_base=0, trans_mode=0, use_views=views, f64_params=False, lr=0.01)
    e1.record(); torch.cuda.synchronize()
    lib.houv_debug_read_stamps(buf, 1)
    v = np.array(list(buf), dtype=np.float64)
    names = ["move+sync", "sweepA", "epilogueA", "sweepB", "epilogueB", "barrier", "tail+sync", "-"]
    tot = v[:7].sum()
    per = v / (P * K * iters * NW)     # per wave per iteration
    print(f"views={views} pruned={solver.PRUNED} waves/WG={NW}: {e0.elapsed_time(e1):.1f} ms, {e0.elapsed_time(e1)*1e3/(P*K*iters):.3f} us/hyp-iter")
    for n, x, y in zip(names[:7], v[:7], per[:7]):
        print(f"   {n:10s} {100*x/tot:5.1f} %   {y/1e3:8.1f} kcycles per wave-iteration")
    for n, i in (("  epi:select", 9), ("  epi:rescan+sums+wave-reduce", 8), ("  epi:barrier+final-sum", 11)):
        print(f"   {n:16s} {per[i]/1e3:8.1f} kcycles per wave-iteration (all metrics, both directions)")
    if solver.PRUNED and os.environ.get("OWNER_WALK") != "1":
        lib.houv_debug_read_prune_stats(pbuf, 1)
        pv = np.array(list(pbuf), dtype=np.float64)
        for n, i in (("bounds", 3), ("box tests", 4), ("sort", 5), ("walk", 6), ("end barrier", 7)):
            print(f"   sweep:{n:12s} {2 * pv[i] / max(pv[2], 1) / 1e3:8.1f} kcycles per wave-iteration (both sweeps)")
    print(f"   prediction: A rescanned {v[12]/v[15]:.3f}, won by A {v[13]/v[15]:.3f}, repaired {v[14]/v[15]:.4f} of the metric-iterations")
