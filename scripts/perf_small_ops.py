"""Timing probe for the HBM-bound small ops: Kabsch, Chamfer backward, pose/move, ICP."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from houv_amd import ops, synthetic
dev = torch.device("cuda:0")
def timed(fn, n=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
for B, N in ((16384, 2048), (2048, 2048), (256, 2048)):
    src = torch.randn(B, 3, N, device=dev); corr = torch.randn(B, 3, N, device=dev); w = torch.rand(B, 1, N, device=dev)
    t = timed(lambda: ops.kabsch(src, corr)); byts = B * 2 * 3 * N * 4
    print(f"kabsch        B={B} N={N}: {t:.3f} ms  {byts/t/1e6:.0f} GB/s algorithmic ({byts/t/1e6/8000*100:.1f} % of 8 TB/s)")
    t = timed(lambda: ops.kabsch(src, corr, w)); byts = B * (2 * 3 + 1) * N * 4
    print(f"kabsch(w)     B={B} N={N}: {t:.3f} ms  {byts/t/1e6:.0f} GB/s algorithmic")
for B in (4096,):
    N = 2048
    a = torch.rand(B, N, 3, device=dev); b = torch.rand(B, N, 3, device=dev)
    d1 = torch.empty(B, N, device=dev); d2 = torch.empty_like(d1); i1 = torch.empty(B, N, dtype=torch.int32, device=dev); i2 = torch.empty_like(i1)
    ops.chamfer_forward(a, b, d1, d2, i1, i2)
    g1 = torch.rand(B, N, device=dev); g2 = torch.rand(B, N, device=dev)
    ga = torch.zeros_like(a); gb = torch.zeros_like(b)
    t = timed(lambda: ops.chamfer_backward(a, b, ga, gb, g1, g2, i1, i2))
    byts = B * N * (2 * 12 + 2 * 4 + 2 * 4 + 2 * 12 * 2)       # xyz, grad_dist, idx reads; grads RMW
    print(f"chamfer_bwd   B={B} N={N}: {t:.3f} ms  {byts/t/1e6:.0f} GB/s algorithmic; atomics {B*N*12*4/t/1e6:.0f} GB/s of added bytes")
    p = torch.randn(B, 8, device=dev)
    t = timed(lambda: ops.pose_forward(p, 0, 0, a)); byts = B * N * 24
    print(f"pose+move     B={B} N={N}: {t:.3f} ms  {byts/t/1e6:.0f} GB/s algorithmic")
P, N = 256, 2048
s, tg, pose = synthetic.make_pairs(P, N, seed=3)
s, tg, pose = s.to(dev), tg.to(dev), pose.to(dev)
out = ops.icp_refine(s, tg, pose, 0.02, 500)
t = timed(lambda: ops.icp_refine(s, tg, pose, 0.02, 500), n=3)
its = out["iterations"].float()
print(f"icp_refine    P={P} N={N}: {t:.2f} ms, iterations mean {its.mean():.1f} max {its.max():.0f}; {t*1e3/ (its.max()+1):.1f} us per (max) iteration")
