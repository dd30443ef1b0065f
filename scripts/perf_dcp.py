"""Timing probe for the DCP head: GEMM shapes of the model and the whole forward at 2048 points."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch
from houv_amd import ops, synthetic
from houv_amd.models.dcp import Model
dev = torch.device("cuda:0")
def timed(fn, n=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
for (M, N, K, tb, name) in [(8 * 2048 * 20, 64, 64, True, "conv2"), (8 * 2048 * 20, 128, 64, True, "conv3"),
                            (8 * 2048 * 20, 256, 128, True, "conv4"), (8 * 2048, 512, 512, True, "linear 512"),
                            (8 * 2048, 1024, 512, True, "ff w1"), (8 * 2048, 512, 1024, True, "ff w2"),
                            (4096, 4096, 4096, True, "square 4096 NT"), (4096, 4096, 4096, False, "square 4096 NN")]:
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev) if tb else torch.randn(K, N, device=dev)
    C = torch.empty(M, N, device=dev)
    t = timed(lambda: ops.gemm(A, B, C, trans_b=tb))
    print(f"gemm {name:16s} M={M} N={N} K={K}: {t:.3f} ms  {2.0*M*N*K/t/1e9:.1f} TFLOP/s  ({2.0*M*N*K/t/1e9/157.3*100:.0f} % of fp32 peak); bytes {(M*K+N*K+M*N)*4/t/1e6:.0f} GB/s")
P, H, Nq = 8, 4, 2048
Q = torch.randn(P, Nq, H, 128, device=dev); Kt = torch.randn(P, Nq, H, 128, device=dev); V = torch.randn(P, Nq, H, 128, device=dev)
S = torch.empty(P, H, Nq, Nq, device=dev); ctx = torch.empty(P, Nq, H, 128, device=dev)
t = timed(lambda: ops.gemm(Q.permute(0, 2, 1, 3), Kt.permute(0, 2, 1, 3), S, trans_b=True, alpha=0.1))
print(f"attn QK^T P=8: {t:.3f} ms {2.0*P*H*Nq*Nq*128/t/1e9:.1f} TFLOP/s; write {P*H*Nq*Nq*4/t/1e6:.0f} GB/s")
t = timed(lambda: ops.softmax_rows_(S)); print(f"softmax: {t:.3f} ms {2*P*H*Nq*Nq*4/t/1e6:.0f} GB/s")
t = timed(lambda: ops.gemm(S, V.permute(0, 2, 1, 3), ctx.permute(0, 2, 1, 3), trans_b=False))
print(f"attn PV   P=8: {t:.3f} ms {2.0*P*H*Nq*Nq*128/t/1e9:.1f} TFLOP/s; read {P*H*Nq*Nq*4/t/1e6:.0f} GB/s")
t = timed(lambda: ops.attention(Q, Kt, V, 0.1))
print(f"fused attention P=8: {t:.3f} ms {4.0*P*H*Nq*Nq*128/t/1e9:.1f} TFLOP/s (QK^T + softmax + PV in one kernel)")
x = torch.rand(16, 2048, 3, device=dev)
t = timed(lambda: ops.knn(x, 20)); print(f"knn20 B=16 N=2048: {t:.3f} ms")
net = Model(None, pairs_per_chunk=8).to(dev)
src, tgt, _ = synthetic.make_pairs(32, 2048, seed=2)
src, tgt = src.to(dev), tgt.to(dev)
for fused in (False, True):
    ops.FUSED_ATTENTION = fused
    t = timed(lambda: net(src, tgt), n=3)
    print(f"fused_attention={fused}: DCP forward 32 pairs x 2048 pts: {t:.1f} ms -> {32/(t/1e3):.1f} pairs/s")
flop_pair = 2 * (2048*20*(64*64+64*128+128*256)*2 + 2048*512*512*2) + 2 * (  # two clouds DGCNN; two transformer passes:
    3 * (4*2048*512*512*2 + 2*2048*2048*512*2) + 2*2 * 2048*512*1024*2) + 2048*2048*512*2
print(f"DCP forward 32 pairs x 2048 pts: {t:.1f} ms -> {32/(t/1e3):.1f} pairs/s, ~{flop_pair*32/t/1e9:.1f} TFLOP/s of GEMM work")
