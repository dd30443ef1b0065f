"""houv_attention_f32 on the bf16 matrix pipe (houv_debug_set("attn_split", 1)) against the fp32-input MFMA kernel: error against
float64 and time at the DCP head's shape (16 pairs x 4 heads x 2048 x 2048 x 128)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from houv_amd import _lib, ops
dev = torch.device("cuda:0")
H, dk = 4, 128
for P, Nq, Nk, spread in ((2, 256, 128, 1.0), (2, 2048, 2048, 1.0), (16, 2048, 2048, 1.0), (2, 2048, 2048, 4.0)):
    g = torch.Generator().manual_seed(P + Nq)
    q = (torch.randn(P, Nq, H * dk, generator=g) * spread).to(dev); k = torch.randn(P, Nk, H * dk, generator=g).to(dev)
    v = torch.randn(P, Nk, H * dk, generator=g).to(dev)
    scale = 1.0 / np.sqrt(dk)
    ref = None
    if P <= 2:
        qd, kd, vd = (t.double().view(P, -1, H, dk).permute(0, 2, 1, 3) for t in (q, k, v))
        ref = (torch.softmax(qd @ kd.transpose(-1, -2) * scale, dim=-1) @ vd).permute(0, 2, 1, 3)
    for mode in (0, 1):
        _lib.debug_set("attn_split", mode)
        f = lambda: ops.attention(q.view(P, Nq, H, dk), k.view(P, Nk, H, dk), v.view(P, Nk, H, dk), scale)
        out = f(); f(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        err = f"  max |err| {float((out.double() - ref).abs().max()):.2e}  mean {float((out.double() - ref).abs().mean()):.2e}" if ref is not None else ""
        print(f"P={P:2d} {Nq}x{Nk} spread {spread}: attn_split={mode}: {ms:7.3f} ms {4.0 * P * H * Nq * Nk * dk / ms / 1e9:7.1f} TFLOP/s{err}", flush=True)
_lib.debug_set("attn_split", 0)
