#!/usr/bin/env python3
"""Golden vectors for the DCP head (BASELINE configs[4]) from the REAL reference, CPU only.

    python tests/golden/make_golden_dcp.py

Imports registration/models/dcp.py with the same stubs as make_golden.py (+ an empty `h5py`, and `torch.arange`
ignoring device='cuda', because get_graph_feature hard-codes the device, dcp.py:48-50).  The repository ships no
trained DCP weights, so the model is seeded-random-initialised (BatchNorm running statistics randomised too, eval
mode); the weights come from tests/golden/dcp_weights.py (seeded numpy), loaded into the reference model here and into houv_amd's model in the tests."""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    torch.set_num_threads(8)
    sys.modules["h5py"] = types.ModuleType("h5py")
    mg.import_reference()
    _arange = torch.arange
    torch.arange = lambda *a, **k: _arange(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
    import models.dcp as dcp

    import dcp_weights
    net = dcp.Model(args=None)
    state = {k: torch.tensor(v) for k, v in dcp_weights.make_state(1234).items()}
    missing, unexpected = net.load_state_dict(state, strict=False)
    assert not unexpected and all(m.endswith("num_batches_tracked") or m == "head.reflect" for m in missing), (missing, unexpected)
    net.eval()
    rng = np.random.default_rng(77)
    out = {}
    cases = {"small": (2, 48), "mid": (2, 256)}
    big = len(sys.argv) > 1 and sys.argv[1] == "2048"     # `make_golden_dcp.py 2048`: BASELINE configs[4]'s cloud size -> g19
    if big:
        cases = {"full": (1, 2048)}
        rng = np.random.default_rng(2048)
    for name, (B, N) in cases.items():
        pairs = [mg.synth_pair(rng, N, 45) for _ in range(B)]
        src = torch.tensor(np.stack([p[0] for p in pairs]))
        tgt = torch.tensor(np.stack([p[1] for p in pairs]))
        with torch.no_grad():
            s = src.transpose(1, 2).contiguous()
            t = tgt.transpose(1, 2).contiguous()
            idx = dcp.knn(s, 20)
            es = net.emb_nn(s)
            et = net.emb_nn(t)
            ps, pt = net.pointer(es, et)
            R, tr = net.head(es + ps, et + pt, s, t)
            T12 = net(src, tgt)
        out.update({f"{name}_src": src.numpy(), f"{name}_tgt": tgt.numpy(), f"{name}_knn_src": idx.numpy().astype(np.int32),
                    f"{name}_R": R.numpy(), f"{name}_t": tr.numpy(), f"{name}_T12": T12.numpy()})
        if name == "small":      # full intermediate tensors only at the small size (keeps the fixture small)
            out.update({f"{name}_emb_src": es.numpy(), f"{name}_emb_tgt": et.numpy(), f"{name}_ptr_src": ps.numpy(),
                        f"{name}_ptr_tgt": pt.numpy()})
        else:                    # a strided sample of the embeddings at the larger size
            out.update({f"{name}_emb_src_s": es.numpy()[:, ::8, ::8], f"{name}_ptr_tgt_s": pt.numpy()[:, ::8, ::8]})
    if big:
        np.savez_compressed(f"{OUT}/g19_dcp2048.npz", **out)
        print("wrote g19_dcp2048.npz")
        return
    # PointNet embedding (dcp.py:246-266) with its own seeded weights
    pn = dcp.PointNet(512)
    pstate = dcp_weights.make_pointnet_state(99)
    pn.load_state_dict({k: torch.tensor(v) for k, v in pstate.items()}, strict=False)
    pn.eval()
    with torch.no_grad():
        out["pointnet_emb"] = pn(torch.tensor(out["small_src"]).transpose(1, 2).contiguous()).numpy()
    np.savez_compressed(f"{OUT}/g9_dcp.npz", **out)
    print("wrote g9_dcp.npz (weights are regenerated from tests/golden/dcp_weights.py, seed 1234)")


if __name__ == "__main__":
    main()
