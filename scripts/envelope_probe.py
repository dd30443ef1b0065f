"""Exploration: GPU-vs-reference divergence per horizon next to the reference's own chaos envelope (G16/G17/G20/G18)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from houv_amd.models.houv import HOUV, predict_model
from houv_amd.train_utils import getPredict_angle
dev = torch.device("cuda:0")


def q(x):
    return "q50 %.2e q90 %.2e q99 %.2e max %.2e  frac>1e-3 %.3f" % (np.median(x), np.quantile(x, .9), np.quantile(x, .99), x.max(), (x > 1e-3).mean())


for name in ("g16_envelope128.npz", "g17_envelope512.npz", "g20_envelope2048.npz", "g18_twin_envelope.npz", "g21_twin_envelope2048.npz"):
    path = os.path.join(ROOT, "tests", "golden", name)
    if not os.path.exists(path):
        continue
    g = np.load(path)
    K = int(g["kernel"])
    s, t = torch.tensor(g["src"]).to(dev), torch.tensor(g["tgt"]).to(dev)
    print(name, "pairs", s.shape[0], "points", s.shape[1], "K", K)
    for h in g["horizons"]:
        h = int(h)
        if "twin" in name:
            np.random.seed(int(g["np_seed"]))
            m1, R, T, ts = getPredict_angle(s, t, kernel=K, num_epochs=h, angle_base=1)
        else:
            m1, R, T = predict_model(HOUV(s.shape[0] * K, 0), s, t, kernel=K, num_epochs=h, angle_base=0)
        R = R.reshape(-1, 9).cpu().numpy(); m1 = m1.reshape(-1).cpu().numpy()
        ref = g[f"ref_n{h}_R"].reshape(-1, 9)
        envR = np.maximum(np.abs(ref - g[f"pertA_n{h}_R"].reshape(-1, 9)).max(1), np.abs(ref - g[f"pertB_n{h}_R"].reshape(-1, 9)).max(1))
        gpuR = np.abs(R - ref).max(1)
        envm = np.maximum(np.abs(g[f"ref_n{h}_min1"] - g[f"pertA_n{h}_min1"]), np.abs(g[f"ref_n{h}_min1"] - g[f"pertB_n{h}_min1"]))
        gpum = np.abs(m1 - g[f"ref_n{h}_min1"])
        print(f"  h={h:4d} R   env: {q(envR)}\n          R   gpu: {q(gpuR)}\n          m1  env: {q(envm)}\n          m1  gpu: {q(gpum)}")
        both = (gpuR > 1e-3) & (envR > 1e-3)
        print(f"          hypotheses diverged (>1e-3) gpu-only {(gpuR > 1e-3).sum() - both.sum()}, env-only {(envR > 1e-3).sum() - both.sum()}, both {both.sum()}")
