"""Diagnostic (not a test): per-element gradient error of one fused forward/backward vs the fp32 oracle and an
all-float64 evaluation of the same formulas."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import houv_ref_cpu as orc
from houv_amd import ops, synthetic

def oracle_grads(src, tgt, params, base, dtype):
    tv = [torch.tensor(params[:, a:b].astype(dtype), requires_grad=True) for a, b in ((0, 3), (3, 4), (4, 7), (7, 8))]
    moved, R, T = orc.houv_forward(src.to(tv[0].dtype), *tv, base, "houv")
    loss, min1 = orc.predict_loss(moved, tgt.to(tv[0].dtype))
    loss.mean().backward()
    return np.concatenate([t.grad.numpy() for t in tv], 1), loss.detach().numpy()

dev = torch.device("cuda:0")
for (N, M, base) in [(96, 96, 1), (1100, 1100, 1), (600, 600, 0)]:
    P = 30
    src, tgt, _ = synthetic.make_pairs(P, max(N, M), seed=77)
    src, tgt = src[:, :N].contiguous(), tgt[:, :M].contiguous()
    rng = np.random.default_rng(N + base)
    params = rng.standard_normal((P, 8)).astype(np.float32).astype(np.float64)
    g32, l32 = oracle_grads(src, tgt, params, base, np.float32)
    g64, l64 = oracle_grads(src, tgt, params, base, np.float64)
    state = torch.zeros((P, 24), dtype=torch.float64, device=dev)
    state[:, :8] = torch.tensor(params).to(dev)
    out = ops.solve_iterate(src.to(dev), tgt.to(dev), state, 1, steps_done=0, n_iters=1, angle_base=base, trans_mode=0,
                            use_views=True, f64_params=False, k_full=int(N * 0.5), k_view=N, lr=0.01, loss_scale=1.0 / P,
                            want_grad=True, want_cd=True)
    g = out["grad"].cpu().numpy().astype(np.float64)
    scale = np.abs(g64).max(axis=1, keepdims=True)
    e_gpu64 = np.abs(g - g64) / scale
    e_3264 = np.abs(g32 - g64) / scale
    e_gpu32 = np.abs(g - g32) / scale
    print(f"N={N} M={M} base={base}: max rel err  gpu-vs-f64 {e_gpu64.max():.2e}  oracle32-vs-f64 {e_3264.max():.2e}  gpu-vs-oracle32 {e_gpu32.max():.2e}")
    worst = np.argsort(-e_gpu64.reshape(-1))[:5]
    for w in worst:
        r, c = divmod(w, 8)
        print(f"   hyp {r} param {c}: gpu {g[r,c]: .6e} o32 {g32[r,c]: .6e} o64 {g64[r,c]: .6e} scale {scale[r,0]:.3e}")
    print("   loss err gpu-vs-f64", np.abs(out["loss"].cpu().numpy() - l64).max(), " o32-vs-f64", np.abs(l32 - l64).max())
