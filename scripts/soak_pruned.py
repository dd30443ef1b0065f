"""Random-size soak of the exact pruned search against the brute-force kernel: same clouds, optimiser state and outputs must be
bit-identical.  Sizes 257..4096 (edges included), ragged N != M without the view terms, chunked launches, all four angle bases."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from houv_amd import _lib, solver, synthetic
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(os.environ.get("SEED", 7)))
edges = [257, 258, 288, 289, 511, 512, 513, 767, 768, 769, 1023, 1024, 1025, 1535, 1536, 1537, 2047, 2048, 2049, 2050, 2111, 2112,
         3071, 3072, 3073, 4031, 4095, 4096]
n_cases, bad = int(os.environ.get("CASES", 120)), 0
for c in range(n_cases):
    views = bool(rng.integers(0, 2))
    N = int(edges[c % len(edges)] if c < 2 * len(edges) else rng.integers(257, 4097))
    M = N if views else int(rng.choice([N, rng.integers(40, 4097), rng.integers(max(N - 64, 33), min(N + 64, 4096) + 1)]))
    if max(N, M) < 257:
        M = 300
    P, K = int(rng.integers(1, 3)), 26
    if not views and int(N * 0.5) > M:                    # top-k needs k_full <= M (the reference raises otherwise)
        M = max(M, int(N * 0.5) + 1)
    if not views and int(N * 0.5) < 1:
        continue
    mx = max(N, M)
    src, tgt, _ = synthetic.make_pairs(P, mx, seed=1000 + c)
    leaf = solver.sort_leaf(N, M)
    src = solver.spatial_sort(src[:, :N].contiguous().to(dev), leaf)
    tgt = solver.spatial_sort(tgt[:, :M].contiguous().to(dev), leaf)
    p0 = solver.houv_init_params(P * K, seed=2021 + c) if views else np.random.default_rng(c).standard_normal((P * K, 8))
    iters, chunk = int(rng.integers(3, 14)), int(rng.choice([50, 1, 4]))
    kw = dict(angle_base=int(rng.integers(0, 4)), trans_mode=0 if views else 1, use_views=views, f64_params=not views,
              lr=0.01 if views else 0.1, want_grad=True, want_cd=True, iters_per_launch=chunk)
    ref, st_ref = solver.run_stage(src, tgt, p0, K, iters, pruned=False, **kw)
    out, st = solver.run_stage(src, tgt, p0, K, iters, pruned=True, **kw)
    ok = torch.equal(st, st_ref) and all(torch.equal(out[k], ref[k]) for k in ("score", "loss", "R", "T", "grad", "cd"))
    mode = _lib.solve_variant(N, M, True, with_mode=True)
    if not ok or c % 20 == 0:
        print(f"case {c:3d} N={N:4d} M={M:4d} views={views!s:5s} P={P} iters={iters:2d} chunk={chunk:2d} variant={mode}: {'identical' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print(f"{n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
