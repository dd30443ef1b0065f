"""A/B of the stand-alone Chamfer op on one device: HOUV_CHAMFER_DIRECT=1 selects the direct-difference sweep,
default = the expanded-form filter + exact recovery.  BASELINE.md section 3: B' = 4096..16384 x 2048 x 2048."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from houv_amd import _lib, ops, synthetic
if os.environ.get("HOUV_CHAMFER_DIRECT") == "1":   # the script's own switch -> the library's diagnostic knob
    _lib.debug_set("chamfer_direct", 1)
dev = torch.device("cuda:0")
torch.manual_seed(0)   # same clouds for every variant: the checksums must agree
out = []
CASES = (("uniform", 4096, 2048), ("uniform", 8192, 2048), ("mvp", 4096, 2048), ("uniform", 4096, 1024), ("uniform", 2048, 4096))
if os.environ.get("ONLY"):
    CASES = tuple(c for c in CASES if "%s%d_%d" % c == os.environ["ONLY"])
for name, B, N in CASES:
    if name == "uniform":
        a = torch.rand(B, N, 3, device=dev) - 0.5; b = torch.rand(B, N, 3, device=dev) - 0.5
    else:
        s, t, _ = synthetic.make_pairs(64, N, seed=3)
        a = s.repeat(B // 64, 1, 1).to(dev).contiguous(); b = t.repeat(B // 64, 1, 1).to(dev).contiguous()
    d1 = torch.empty(B, N, device=dev); d2 = torch.empty_like(d1)
    i1 = torch.empty(B, N, dtype=torch.int32, device=dev); i2 = torch.empty_like(i1)
    # sustained rate: the clock ramps over the first milliseconds of a burst (GRBM_GUI_ACTIVE: 1.9 GHz inside an isolated
    # 2.7 ms launch, 2.3 GHz inside a 1.4 s one), so warm up and time a train of back-to-back launches
    reps = int(os.environ.get("REPS", 30))
    for _ in range(10):
        ops.chamfer_forward(a, b, d1, d2, i1, i2)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.chamfer_forward(a, b, d1, d2, i1, i2)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    t = min(ts)
    out.append(f"{name}:{B}x{N}^2 {t:.3f} ms ({2.0 * B * N * N / t / 1e9:.2f} Tpair/s) chk {int(i1.long().sum() + i2.long().sum())} {float(d1.double().sum()):.6f}")
print("DIRECT" if os.environ.get("HOUV_CHAMFER_DIRECT") == "1" else "FILTER", " | ".join(out), flush=True)
