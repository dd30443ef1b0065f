"""Sizes at which the GPU tests compare the fused loop with the CPU oracle (test_gpu_solve.py), shared with the CPU-side
test that enumerates the kernel variant table (test_host_logic.py) and fails when a variant has no such size.

ORACLE_CASES: (N, M, angle_base, mode) -- mode "houv" runs solve_kernel<.., NMET=4> (HOUV module: view terms on, fp32 Adam),
"solve" runs <.., NMET=1> (train_utils.solve twin: no views, float64 leaves).
PRUNED_CASES: (N, M, views, f64_params, trans_mode) -- the pruned kernels are compared BIT FOR BIT with the brute-force
kernel of the same variant, which in turn is compared with the oracle above.  Up to 256 points the library serves the
pruned entry point with the brute-force kernel (prune mode 0): those sizes check exactly that."""

ORACLE_CASES = [
    (64, 64, 3, "houv"), (40, 64, 1, "solve"),                 # <64,1>
    (128, 128, 0, "houv"), (100, 128, 2, "solve"),             # <128,1>
    (200, 200, 1, "houv"), (96, 160, 1, "solve"),              # <256,1>
    (400, 400, 2, "houv"), (300, 300, 3, "solve"),             # <256,2>
    (600, 600, 0, "houv"), (700, 640, 2, "solve"),             # <256,3>
    (1000, 1000, 2, "houv"), (900, 1024, 1, "solve"),          # <256,4>
    (1100, 1100, 1, "houv"), (1500, 1200, 0, "solve"),         # <512,3>
    (2048, 2048, 1, "houv"), (2048, 1800, 3, "solve"),         # <512,4>  (BASELINE configs[1]'s kernel)
    (2500, 2500, 2, "houv"), (3000, 2200, 0, "solve"),         # <1024,3>
    (4096, 4096, 0, "houv"), (3500, 4000, 2, "solve"),         # <1024,4>
]

PRUNED_CASES = [
    (200, 200, True, False, 0), (180, 256, False, True, 1),    # <256,1>: the pruned entry point runs the brute-force kernel
    (400, 400, True, False, 0), (300, 512, False, True, 1),    # <256,2>
    (700, 700, True, False, 0), (768, 600, False, True, 1),    # <256,3>
    (1000, 1000, True, False, 0), (900, 1024, False, True, 1),  # <256,4>
    (1400, 1400, True, False, 0), (1000, 1300, False, True, 1),  # <512,3>
    (2048, 2048, True, False, 0), (2048, 1700, False, True, 1),  # <512,4>, balanced walk
    (2500, 2500, True, False, 0), (3000, 2200, False, True, 1),  # <1024,3>, balanced walk over 64-point super-tiles (PRUNE = 3)
    (4096, 4096, True, False, 0), (3500, 4000, False, True, 1),  # <1024,4>
]


def oracle_batch(N, M):
    """Hypotheses per case: the oracle materialises float64 [P, N, M] temporaries (several per metric under autograd)."""
    mx = max(N, M)
    return 30 if mx <= 1100 else (8 if mx <= 3000 else 4)
