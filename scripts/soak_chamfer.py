"""One-off soak of the stand-alone Chamfer op against the C oracle: many random shapes (incl. > 1024 points, where the
assembly sub-tile loop runs, and > 2048, where the references stream through LDS in passes), value ranges, lattices
(exact ties) and duplicates; bit-exact distances and indices required.  EXAMPLES=n python scripts/soak_chamfer.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import c_oracle
from houv_amd.metrics import cd
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(os.environ.get("SEED", 1)))
n = int(os.environ.get("EXAMPLES", 300))
bad = 0
for it in range(n):
    B = int(rng.integers(1, 4)); N = int(rng.integers(1, 3000)); M = int(rng.integers(1025, 5000))
    scale = float(rng.choice([1e-3, 1.0, 37.5, 1e3])); quant = int(rng.choice([0, 0, 4, 64, 1024])); dup = bool(rng.integers(0, 2))
    off = float(rng.choice([0.0, 0.0, 5.0, 300.0])) * scale
    a = (rng.random((B, N, 3)) - 0.5) * scale + off
    b = (rng.random((B, M, 3)) - 0.5) * scale + off
    if quant:
        a = np.round(a / scale * quant) / quant * scale; b = np.round(b / scale * quant) / quant * scale
    if dup and M > 3:
        b[:, rng.integers(0, M, M // 3)] = b[:, rng.integers(0, M, M // 3)]
    a = a.astype(np.float32); b = b.astype(np.float32)
    d1, d2, i1, i2 = cd()(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev))
    o1, o2, j1, j2 = c_oracle.chamfer_forward(a, b)
    ok = (np.array_equal(d1.cpu().numpy().view(np.uint32), o1.view(np.uint32)) and np.array_equal(d2.cpu().numpy().view(np.uint32), o2.view(np.uint32))
          and np.array_equal(i1.cpu().numpy(), j1) and np.array_equal(i2.cpu().numpy(), j2))
    if not ok:
        bad += 1
        print("MISMATCH", B, N, M, scale, quant, dup, off, flush=True)
    if it % 50 == 49:
        print(f"{it + 1} examples, {bad} mismatches", flush=True)
print(f"done: {n} examples, {bad} mismatches")
sys.exit(1 if bad else 0)
