"""fp32 GEMM on the bf16 matrix pipe (houv_debug_set("gemm_split", 6 | 3)) against the fp32-input MFMA kernel: error against an fp64
product and time, at the DCP head's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from houv_amd import _lib, ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
shapes = [("conv2 64->64", 327680, 64, 64), ("conv3 64->128", 327680, 128, 64), ("conv4 128->256", 327680, 256, 128),
          ("conv5 512->512", 32768, 512, 512), ("linear 512", 32768, 512, 512), ("ff w1", 32768, 1024, 512), ("ff w2", 32768, 512, 1024),
          ("scores 2048x2048x512 (x16)", 2048, 2048, 512), ("square 4096", 4096, 4096, 4096)]
for name, M, N, K in shapes:
    batch = 16 if name.startswith("scores") else 1
    A = torch.randn(batch, M, K, device=dev) * torch.rand(batch, M, 1, device=dev).mul(4).exp()       # rows of different scale
    B = torch.randn(batch, N, K, device=dev)
    ref = None
    if M * N * batch <= 2048 * 2048 * 16:
        ref = torch.matmul(A[:, :4096].double(), B.double().transpose(1, 2))
    out = {}
    for mode in (0, 6, 3):
        _lib.debug_set("gemm_split", mode)
        C = torch.empty(batch, M, N, device=dev)
        for _ in range(2): ops.gemm(A, B, C, trans_b=True)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.gemm(A, B, C, trans_b=True)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        err = ""
        if ref is not None:
            d = (C[:, :4096].double() - ref).abs()
            scale = (A[:, :4096].double().abs() @ B.double().abs().transpose(1, 2))     # sum |a||b|: the natural error scale
            err = f"  max err / sum|a||b| = {float((d / scale).max()):.2e}  mean {float((d / scale).mean()):.2e}"
        out[mode] = ms
        print(f"{name:28s} split={mode}: {ms:7.3f} ms {2.0 * batch * M * N * K / ms / 1e9:7.1f} TFLOP/s{err}", flush=True)
_lib.debug_set("gemm_split", 0)
