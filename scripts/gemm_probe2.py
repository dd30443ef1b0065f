"""One GEMM shape in a loop, for PMC passes: SHAPE=M,N,K."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from houv_amd import ops
dev = torch.device("cuda:0")
M, N, K = (int(x) for x in os.environ.get("SHAPE", "16384,512,512").split(","))
A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
for _ in range(20):
    ops.gemm(A, B, C, trans_b=True)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    ops.gemm(A, B, C, trans_b=True)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 50
print(f"gemm {M}x{N}x{K}: {t:.4f} ms {2.0*M*N*K/t/1e9:.1f} TFLOP/s")
