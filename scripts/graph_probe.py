import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from houv_amd import ops
dev = torch.device("cuda:0")
P, N, H, dk = 2, 256, 4, 128
q = torch.randn(P, N, H * dk, device=dev); k = torch.randn(P, N, H * dk, device=dev); v = torch.randn(P, N, H * dk, device=dev)
f = lambda: ops.attention(q.view(P, N, H, dk), k.view(P, N, H, dk), v.view(P, N, H, dk), 0.1)
ref = f(); torch.cuda.synchronize()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    f()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        out = f()
    q.mul_(1.0)   # same inputs
    g.replay(); torch.cuda.synchronize()
    print("graph replay ok, equal:", torch.equal(out, ref))
    q.copy_(torch.randn_like(q)); ref2 = f(); g.replay(); torch.cuda.synchronize()
    print("second replay equal:", torch.equal(out, ref2))
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:300])
