#!/usr/bin/env python3
"""Round-2 golden vectors from the REAL reference (CPU, build container only; needs /root/reference):

    python tests/golden/make_golden_r2.py traj2048       # G15: trajectory at 2048 points (BASELINE configs[1]'s cloud size)
    python tests/golden/make_golden_r2.py envelope 128   # G16: chaos envelope of predict_model, G12's first 16 pairs
    python tests/golden/make_golden_r2.py envelope 512   # G17: same on G13's first 6 pairs (configs[0]'s cloud size)
    python tests/golden/make_golden_r2.py envelope 2048  # G20: same on G15's 2048x2048-point pair (the bench's cloud size; ~1.5 h of CPU)
    python tests/golden/make_golden_r2.py twin           # G18: chaos envelope of train_utils.getPredict_angle (lr 0.1)
    python tests/golden/make_golden_r2.py twin 2048      # G21: the same on G15's 2048x2048-point pair

The reference is imported exactly as make_golden.py does (its own chamfer_python.py stands in for the CUDA extension).
Nothing of it is edited: per-iteration snapshots are taken from OUTSIDE, by wrapping the module-level functions its
loops call (registration/models/houv.py:120-126 calls ``net(src)`` and ``Predict_loss``; train_utils.py:397-440 calls
``rotation``, ``translation`` and ``calc_cd_percent``), so one 200-iteration run yields every horizon.  Only data is
written (tests/golden/g15..g18*.npz)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from make_golden import import_reference  # noqa: E402

HORIZONS = (1, 2, 5, 20, 50, 100, 200)


def record_predict_model(houv, src, tgt, kernel, num_epochs, base, horizons, want_grad=False):
    """Run the reference's predict_model once and return {h: dict(min1,loss,R,T,params)} for the FORWARD of iteration h
    (1-based), i.e. exactly what predict_model(num_epochs=h) returns, plus the parameters that forward read."""
    snaps, state = {}, {"it": 0}
    net = houv.HOUV(src.shape[0] * kernel, 0)

    def fwd_hook(mod, inp, out):
        state["it"] += 1
        if state["it"] in horizons:
            snaps[state["it"]] = dict(
                R=out[1].detach().numpy().copy(), T=out[2].detach().numpy().copy().reshape(-1, 3),
                params=np.concatenate([mod.V_c.detach().numpy(), mod.angle_c.detach().numpy(),
                                       mod.tran_c.detach().numpy(), mod.tran_s.detach().numpy()], 1).copy())

    orig_loss = houv.Predict_loss

    def loss_wrap(a, b, *args, **kw):
        loss, min1 = orig_loss(a, b, *args, **kw)
        if state["it"] in horizons:
            snaps[state["it"]]["min1"] = min1.detach().numpy().copy()
            snaps[state["it"]]["loss"] = loss.detach().numpy().copy()
        return loss, min1

    h = net.register_forward_hook(fwd_hook)
    houv.Predict_loss = loss_wrap
    try:
        m1, R, T = houv.predict_model(net, src, tgt, kernel=kernel, num_epochs=num_epochs, angle_base=base)
    finally:
        houv.Predict_loss = orig_loss
        h.remove()
    last = snaps[num_epochs]
    assert np.array_equal(last["min1"].reshape(m1.shape), m1.detach().numpy())          # the hooks see what it returns
    assert np.array_equal(last["R"].reshape(R.shape), R.detach().numpy())
    if want_grad:
        g = np.concatenate([net.V_c.grad.numpy(), net.angle_c.grad.numpy(), net.tran_c.grad.numpy(),
                            net.tran_s.grad.numpy()], 1).copy()
        return snaps, g
    return snaps


def record_twin(train_utils, src, tgt, kernel, num_epochs, base, horizons, np_seed):
    """Same for train_utils.getPredict_angle (float64 leaves from the global numpy RNG, lr 0.1, sigma = sin(s pi))."""
    snaps, state = {}, {"it": 0}
    o_rot, o_tr, o_cd = train_utils.rotation, train_utils.translation, train_utils.calc_cd_percent

    def rot(angle, V, *a, **k):
        state["it"] += 1
        R = o_rot(angle, V, *a, **k)
        if state["it"] in horizons:
            snaps[state["it"]] = dict(R=R.detach().numpy().copy())
        return R

    def tr(tran, s, *a, **k):
        T = o_tr(tran, s, *a, **k)
        if state["it"] in horizons:
            snaps[state["it"]]["T"] = T.detach().numpy().copy().reshape(-1, 3)
        return T

    def cdp(a, b, *args, **kw):
        c = o_cd(a, b, *args, **kw)
        if state["it"] in horizons:
            snaps[state["it"]]["min1"] = torch.minimum(c[0], c[1]).detach().numpy().copy()
        return c

    train_utils.rotation, train_utils.translation, train_utils.calc_cd_percent = rot, tr, cdp
    try:
        np.random.seed(np_seed)
        m1, R, T, ts = train_utils.getPredict_angle(src, tgt, kernel=kernel, num_epochs=num_epochs, angle_base=base)
    finally:
        train_utils.rotation, train_utils.translation, train_utils.calc_cd_percent = o_rot, o_tr, o_cd
    assert np.array_equal(snaps[num_epochs]["R"].reshape(R.shape), R.detach().numpy())
    assert np.array_equal(snaps[num_epochs]["min1"].reshape(m1.shape), m1.detach().numpy())
    return snaps, ts.detach().numpy().copy()


def pack(out, tag, snaps):
    for h, s in snaps.items():
        for k, v in s.items():
            out[f"{tag}_n{h}_{k}"] = v


def main():
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", "8")))
    what = sys.argv[1]
    _, houv, train_utils, _, _ = import_reference()
    from houv_amd import synthetic
    if what == "traj2048":
        # one 2048x2048-point pair, K=26 (the fewest restarts reset_weight accepts), bases 0 and 2, forwards 1,2,3,6,21:
        # forward n+1 reads the parameters AFTER n Adam steps, so this pins 1/2/5/20-step states like G5 does
        K, hs = 26, (1, 2, 3, 6, 21)
        src, tgt, pose = synthetic.make_pairs(1, 2048, seed=1515)
        out = dict(src=src.numpy(), tgt=tgt.numpy(), kernel=np.int64(K))
        for base in (0, 2):
            snaps = record_predict_model(houv, src, tgt, K, 21, base, hs)
            pack(out, f"b{base}", snaps)
            _, g = record_predict_model(houv, src, tgt, K, 1, base, (1,), want_grad=True)
            out[f"b{base}_grad"] = g
            print("traj2048 base", base, "min1", snaps[21]["min1"][:4], flush=True)
        np.savez_compressed(os.path.join(HERE, "g15_traj2048.npz"), **out)
    elif what == "envelope":
        N = int(sys.argv[2])
        K, hs = 26, (20, 50, 100, 200)
        if N == 128:
            g = np.load(os.path.join(HERE, "g12_stat.npz")); P, name = 16, "g16_envelope128.npz"
        elif N == 512:
            g = np.load(os.path.join(HERE, "g13_stat512.npz")); P, name = 6, "g17_envelope512.npz"
        else:
            g = np.load(os.path.join(HERE, "g15_traj2048.npz")); P, name = 1, "g20_envelope2048.npz"   # BASELINE configs[1]'s cloud size
        src, tgt = torch.tensor(g["src"][:P]), torch.tensor(g["tgt"][:P])
        out = dict(src=src.numpy(), tgt=tgt.numpy(), kernel=np.int64(K), horizons=np.array(hs))
        # two independent relative 1e-7 perturbations of the inputs: the envelope is the larger of the two divergences
        runs = (("ref", 1.0, 1.0), ("pertA", 1.0 + 1e-7, 1.0), ("pertB", 1.0, 1.0 - 1e-7))
        for tag, fs, ft in runs:
            snaps = record_predict_model(houv, (src * fs).float(), (tgt * ft).float(), K, 200, 0, hs)
            pack(out, tag, snaps)
            print("envelope", N, tag, "done", flush=True)
        np.savez_compressed(os.path.join(HERE, name), **out)
    elif what == "twin":
        K, hs = 26, (5, 20, 50, 100)
        big = len(sys.argv) > 2 and sys.argv[2] == "2048"      # `twin 2048`: G21, G15's 2048x2048-point pair
        g = np.load(os.path.join(HERE, "g15_traj2048.npz" if big else "g12_stat.npz"))
        P = 1 if big else 8
        src, tgt = torch.tensor(g["src"][:P]), torch.tensor(g["tgt"][:P])
        out = dict(src=src.numpy(), tgt=tgt.numpy(), kernel=np.int64(K), horizons=np.array(hs), np_seed=np.int64(31))
        for tag, fs, ft in (("ref", 1.0, 1.0), ("pertA", 1.0 + 1e-7, 1.0), ("pertB", 1.0, 1.0 - 1e-7)):
            snaps, ts = record_twin(train_utils, (src * fs).float(), (tgt * ft).float(), K, 100, 1, hs, 31)
            pack(out, tag, snaps)
            out[f"{tag}_tran_s"] = ts
            print("twin", tag, "done", flush=True)
        np.savez_compressed(os.path.join(HERE, "g21_twin_envelope2048.npz" if big else "g18_twin_envelope.npz"), **out)
    else:
        raise SystemExit("unknown target " + what)


if __name__ == "__main__":
    main()
