import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from houv_amd import solver, synthetic
dev = torch.device("cuda:0")
src, tgt, _ = synthetic.make_pairs(256, 2048, seed=1)
src = src.to(dev)
for rule in ("extent", "area"):
    solver.KD_RULE = rule
    for _ in range(2): solver.kd_sort(src, 32)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): solver.kd_sort(src + 0.0, 32)
    torch.cuda.synchronize(); print(rule, "kd_sort of 256 x 2048 points: %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
solver.KD_RULE = "extent"
