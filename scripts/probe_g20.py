import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from houv_amd import solver
from houv_amd.models.houv import HOUV, predict_model
dev = torch.device("cuda:0")
g = np.load(os.path.join(sys.path[0], "tests", "golden", "g20_envelope2048.npz"))
K = int(g["kernel"]); s, t = torch.tensor(g["src"]).to(dev), torch.tensor(g["tgt"]).to(dev)
def div(a, ref): return np.abs(a.reshape(len(ref), -1) - ref.reshape(len(ref), -1)).max(axis=1)
for mode, pr in (("unsorted/brute", False), ("morton/pruned", True), ("kd/pruned", True)):
    solver.PRUNED = pr; solver.SPATIAL_SORT = "kd" if mode.startswith("kd") else "morton"
    for h in (int(x) for x in g["horizons"]):
        m1, R, Tt = predict_model(HOUV(K, 0), s, t, kernel=K, num_epochs=h, angle_base=0)
        out = []
        for key, val in (("R", R), ("T", Tt), ("min1", m1)):
            ref = g[f"ref_n{h}_{key}"]
            env = np.maximum(div(g[f"pertA_n{h}_{key}"], ref), div(g[f"pertB_n{h}_{key}"], ref))
            d = div(val.cpu().numpy(), ref)
            out.append(f"{key}: q50 {np.median(d):.1e}/{np.median(env):.1e} q90 {np.quantile(d,.9):.1e}/{np.quantile(env,.9):.1e} max {d.max():.1e}/{env.max():.1e}")
        print(f"{mode:16s} h={h:3d}  " + "  ".join(out), flush=True)
