"""Quick kernel timing probe (hipEvents via torch on the launch stream)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from houv_amd import ops, solver, synthetic
from houv_amd.metrics import cd

dev = torch.device("cuda:0")
def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts), float(np.median(ts))

# stand-alone Chamfer op, 2048x2048
for B in (() if os.environ.get('SOLVE_ONLY') else (256, 2048, 8192)):
    a = torch.rand(B, 2048, 3, device=dev); b = torch.rand(B, 2048, 3, device=dev)
    d1 = torch.empty(B, 2048, device=dev); d2 = torch.empty_like(d1)
    i1 = torch.empty(B, 2048, dtype=torch.int32, device=dev); i2 = torch.empty_like(i1)
    mn, md = timed(lambda: ops.chamfer_forward(a, b, d1, d2, i1, i2))
    evals = 2.0 * B * 2048 * 2048
    print(f"chamfer_fwd B={B}: {mn:.3f} ms  {evals/mn/1e9:.1f} Geval/s  {evals*8/mn/1e9:.1f} TFLOP/s(8 flop/eval)  alg {B*81920/mn/1e6:.1f} GB/s", flush=True)

# fused loop
P, K, N = int(os.environ.get("P", 32)), 64, 2048
src, tgt, _ = synthetic.make_pairs(P, N, seed=1)
src, tgt = src.to(dev), tgt.to(dev)
p0 = solver.houv_init_params(P * K)
for views in (True, False):
    for iters in (int(os.environ.get("ITERS", 4)),):
        def run():
            solver.run_stage(src, tgt, p0, K, iters, angle_base=0, trans_mode=0, use_views=views, f64_params=False, lr=0.01)
        mn, md = timed(run, n=2)
        ii = P * K * iters
        evals = ii * 2.0 * N * N
        print(f"solve views={views} P={P} K={K} N={N} iters={iters}: {mn:.1f} ms -> {mn*1e3/ii:.3f} us/inst-iter, {evals/mn/1e9:.1f} G(4-metric)eval/s, "
              f"pairs/s(200it base stage) {P/(mn/1e3*200/iters):.2f}", flush=True)
