#!/usr/bin/env python3
"""G14: end-to-end results of the REAL reference's functional twin ``train_utils.solve`` (the path test.py:64 takes:
500 iterations hard-coded, lr 0.1, float64 leaves drawn from the GLOBAL numpy RNG, loss 6*min_1, retry stages) on 32
synthetic 128-point pairs, kernel=26, 2 batches of 16, ``np.random.seed(1000 + batch)`` before each call (the reference
leaves it unseeded; the harness seeds it, as for G6).

Run ONLY in the build container (needs /root/reference, CPU only, ~15 minutes):

    python tests/golden/make_golden_twin.py

Only data is written: tests/golden/g14_twin.npz (inputs, poses, the reference's ans[B,4,4])."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from make_golden import import_reference  # noqa: E402


def main():
    torch.set_num_threads(8)
    _, _, train_utils, _, _ = import_reference()
    from houv_amd import synthetic
    P, N, K, BATCH = 32, 128, 26, 16
    src, tgt, pose = synthetic.make_pairs(P, N, seed=4321)
    answers = []
    for b in range(0, P, BATCH):
        np.random.seed(1000 + b // BATCH)
        ans = train_utils.solve(src[b:b + BATCH], tgt[b:b + BATCH], kernel=K, prefix='test')
        answers.append(ans.detach().numpy())
        print("batch", b, "done", flush=True)
    ans = np.concatenate(answers)
    np.savez_compressed(os.path.join(HERE, "g14_twin.npz"), src=src.numpy(), tgt=tgt.numpy(), pose=pose.numpy(), ans=ans,
                        kernel=np.int64(K), batch=np.int64(BATCH), seed0=np.int64(1000))
    R, Rg = ans[:, :3, :3], pose.numpy()[:, :3, :3]
    c = np.clip((np.trace(np.einsum('bij,bkj->bik', R, Rg), axis1=1, axis2=2) - 1) / 2, -1, 1)
    r = np.degrees(np.arccos(c))
    print("RotE mean %.3f median %.3f solved %.3f" % (r.mean(), np.median(r), (r < 5).mean()))


if __name__ == "__main__":
    main()
