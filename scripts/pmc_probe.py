"""The fused loop at bench.py's base-stage launch shape (P pairs x 64 hypotheses x ITERS iterations per launch), for the
rocprofv3 PMC passes of scripts/prof_r3.sh.  SOLVER=pruned|brute|both; prints the HIP-event time and the library build id."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from houv_amd import _lib, solver, synthetic
dev = torch.device("cuda:0")
P, K, N, iters = int(os.environ.get("P", 256)), 64, int(os.environ.get("N", 2048)), int(os.environ.get("ITERS", 50))
launches = int(os.environ.get("LAUNCHES", 3))
src, tgt, _ = synthetic.make_pairs(P, N, seed=2021)
src, tgt = solver.spatial_sort(src.to(dev)), solver.spatial_sort(tgt.to(dev))
p0 = solver.houv_init_params(P * K)
which = os.environ.get("SOLVER", "both")
for name in (("pruned", "brute") if which == "both" else (which,)):
    # `launches` chunks of one stage: the 2nd and later launches of the pruned search start from a valid workspace, like
    # launches 2..4 of a 200-iteration bench stage
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    solver.run_stage(src, tgt, p0, K, iters * launches, angle_base=0, trans_mode=0, use_views=True, f64_params=False, lr=0.01,
                     iters_per_launch=iters, pruned=(name == "pruned"))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"{name}: {ms:.1f} ms for {launches} launches of {P * K} hypotheses x {iters} iterations = "
          f"{ms * 1e3 / (P * K * iters * launches):.4f} us per hypothesis-iteration; build {_lib.build_id()}", flush=True)
