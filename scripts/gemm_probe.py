import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from houv_amd import ops
dev = torch.device("cuda:0")
M = N = K = 4096
A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
for _ in range(3): ops.gemm(A, B, C, trans_b=True)
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): ops.gemm(A, B, C, trans_b=True)
b.record(); torch.cuda.synchronize()
t = a.elapsed_time(b) / 5
print(f"gemm 4096^3 NT: {t:.3f} ms {2.0*M*N*K/t/1e9:.1f} TFLOP/s")
