"""Load-balance proxy for BASELINE configs[2] on ONE GPU (VERDICT r1 #4b): solve N_PAIRS synthetic MVP-shaped pairs as
8 contiguous and as 8 interleaved shards (what 8 ranks would each do), record hypothesis-iterations and time per shard,
and report the predicted multi-GPU efficiency = mean / max over shards.  Stragglers can only come from the data-dependent
retry stages (a pair whose base-0 score is > 0.030 costs 4x)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from houv_amd import solver, synthetic
from houv_amd.models.houv import HOUV, predict_model

dev = torch.device("cuda:0")
NP, N, K, IT, W = int(os.environ.get("PAIRS", 2000)), int(os.environ.get("POINTS", 2048)), 64, 200, 8
solver.PRUNED = os.environ.get("HOUV_SOLVER", "pruned") == "pruned"       # bit-identical search, ~1.7x faster
src, tgt, pose = synthetic.make_pairs(NP, N, seed=2021)
src, tgt = src.to(dev), tgt.to(dev)


def solve_shard(idx):
    s, t = src[idx].contiguous(), tgt[idx].contiguous()
    net = HOUV(len(idx) * K, 0).to(dev)
    solver.LAUNCH_LOG = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ans, score, retry = solver.best_of_k_with_retry(
        lambda ss, tt, base: predict_model(net, ss, tt, kernel=K, num_epochs=IT, angle_base=base), s, t)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    hi = sum(n * it for _, _, n, it, *_ in solver.LAUNCH_LOG)
    solver.LAUNCH_LOG = None
    return dt, hi, int(retry.numel())


out = {}
for mode in ("contiguous", "interleaved"):
    rows = []
    for r in range(W):
        per = -(-NP // W)
        idx = torch.arange(r * per, min((r + 1) * per, NP)) if mode == "contiguous" else torch.arange(r, NP, W)
        dt, hi, nretry = solve_shard(idx.to(dev))
        rows.append(dict(rank=r, pairs=len(idx), seconds=dt, hypothesis_iterations=hi, retried_pairs=nretry))
        print(mode, rows[-1], flush=True)
    work = np.array([x["hypothesis_iterations"] for x in rows], float)
    secs = np.array([x["seconds"] for x in rows])
    out[mode] = dict(shards=rows, efficiency_by_work=float(work.mean() / work.max()), efficiency_by_time=float(secs.mean() / secs.max()),
                     pairs_per_s_one_gpu=float(NP / secs.sum()), predicted_pairs_per_s_8_gpus=float(NP / secs.max()))
out["config"] = dict(pairs=NP, points=N, kernel=K, iters=IT, world=W, solver="pruned" if solver.PRUNED else "brute")
print(json.dumps(out))
