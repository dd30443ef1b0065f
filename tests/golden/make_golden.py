#!/usr/bin/env python3
"""Generate the golden vectors G1..G8 from the REAL reference.

Run ONLY in the build container (needs /root/reference, CPU only):

    python tests/golden/make_golden.py

The reference's Python is imported on CPU exactly as SURVEY.md App. B describes:
its own pure-torch Chamfer (utils/metrics/CD/chamfer_python.py, the oracle its
unit_test.py trusts) stands in for the CUDA extension, and modules that are not
installed here (open3d, mm3d_pn2) are empty stubs that are never called on this
path.  Only *data* (inputs + the reference's outputs) is written to
tests/golden/*.npz; no reference source travels anywhere.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def import_reference():
    import matplotlib
    matplotlib.use("Agg")
    chamfer_python = _load_by_path("ref_chamfer_python", f"{REF}/utils/metrics/CD/chamfer_python.py")
    fscore_mod = _load_by_path("ref_fscore", f"{REF}/utils/metrics/CD/fscore.py")

    class cd(torch.nn.Module):  # stands in for the JIT CUDA extension
        def forward(self, a, b):
            return chamfer_python.distChamfer(a.contiguous(), b.contiguous())

    metrics = types.ModuleType("metrics")
    metrics.cd, metrics.fscore, metrics.emd = cd, fscore_mod.fscore, None
    sys.modules["metrics"] = metrics
    mm3d = types.ModuleType("mm3d_pn2")
    for n in ("furthest_point_sample", "gather_points", "grouping_operation", "ball_query", "three_nn"):
        setattr(mm3d, n, None)
    sys.modules["mm3d_pn2"] = mm3d
    sys.modules["open3d"] = types.ModuleType("open3d")
    ident = lambda self, *a, **k: self
    torch.Tensor.cuda = ident
    torch.nn.Module.cuda = ident
    sys.path.insert(0, f"{REF}/registration")
    import models.houv as houv          # noqa
    import train_utils                  # noqa
    import model_utils                  # noqa
    import model_utils_completion       # noqa
    return chamfer_python, houv, train_utils, model_utils, model_utils_completion


def synth_pair(rng, n, max_angle_deg, partial=True):
    """Small MVP-like pair: two partial views of a random closed surface, posed."""
    m = 4 * n
    pts = rng.standard_normal((m, 3))
    pts /= np.linalg.norm(pts, axis=1, keepdims=True)
    pts *= 0.5 * (0.6 + 0.4 * np.abs(np.sin(3 * pts[:, :1]) * np.cos(2 * pts[:, 1:2])))
    def view(d):
        d = d / np.linalg.norm(d)
        order = np.argsort(-(pts @ d))
        return pts[order[:n]] if partial else pts[rng.permutation(m)[:n]]
    a = view(rng.standard_normal(3))
    b = view(rng.standard_normal(3))
    def pose(max_angle, max_trans):
        ax = rng.standard_normal(3); ax /= np.linalg.norm(ax)
        ang = rng.random() * max_angle
        A = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = np.eye(3) + np.sin(ang) * A + (1 - np.cos(ang)) * A @ A
        t = rng.standard_normal(3); t /= np.linalg.norm(t); t *= rng.random() * max_trans
        P = np.eye(4); P[:3, :3] = R; P[:3, 3] = t
        return P
    p1 = pose(np.pi, 0.5)
    tr = pose(np.deg2rad(max_angle_deg), 0.25)
    p2 = tr @ p1
    src = a @ p1[:3, :3].T + p1[:3, 3]
    tgt = b @ p2[:3, :3].T + p2[:3, 3]
    return src.astype(np.float32), tgt.astype(np.float32), tr.astype(np.float32)


def main():
    torch.set_num_threads(8)
    chamfer_python, houv, train_utils, model_utils, muc = import_reference()
    rng = np.random.default_rng(2021)

    # ---- G1: Chamfer fwd/bwd at the reference's own unit-test shapes (unit_test.py:15-20)
    torch.manual_seed(2021)
    p1 = torch.rand(4, 100, 3)
    p2 = torch.rand(4, 200, 3, requires_grad=True)
    d1, d2, i1, i2 = chamfer_python.distChamfer(p1, p2)
    torch.sum(d1).backward()
    big_a = torch.rand(2, 2048, 3)
    big_b = torch.rand(2, 2048, 3)
    with torch.no_grad():
        bd1, bd2, bi1, bi2 = chamfer_python.distChamfer(big_a, big_b)
    # second bwd case: both directions weighted
    q1 = torch.rand(3, 64, 3, requires_grad=True)
    q2 = torch.rand(3, 96, 3, requires_grad=True)
    w1 = torch.rand(3, 64)
    w2 = torch.rand(3, 96)
    e1, e2, j1, j2 = chamfer_python.distChamfer(q1, q2)
    ((e1 * w1).sum() + (e2 * w2).sum()).backward()
    np.savez_compressed(
        f"{OUT}/g1_chamfer.npz",
        p1=p1.numpy(), p2=p2.detach().numpy(), dist1=d1.detach().numpy(), dist2=d2.detach().numpy(),
        idx1=i1.numpy(), idx2=i2.numpy(), grad_p2=p2.grad.numpy(),
        big_a=big_a.numpy(), big_b=big_b.numpy(), big_dist1=bd1.numpy(), big_dist2=bd2.numpy(),
        big_idx1=bi1.numpy(), big_idx2=bi2.numpy(),
        q1=q1.detach().numpy(), q2=q2.detach().numpy(), w1=w1.numpy(), w2=w2.numpy(),
        e1=e1.detach().numpy(), e2=e2.detach().numpy(), j1=j1.numpy(), j2=j2.numpy(),
        grad_q1=q1.grad.numpy(), grad_q2=q2.grad.numpy())

    # ---- G2: loss glue values + grads w.r.t. the moving cloud
    pairs = [synth_pair(rng, 128, 45) for _ in range(3)]
    mv = torch.tensor(np.stack([p[0] for p in pairs]), requires_grad=True)
    tg = torch.tensor(np.stack([p[1] for p in pairs]))
    cdp = muc.calc_cd_percent(mv, tg, percent=0.5)
    lv = [muc.loss_view(mv, tg, dim=d) for d in range(3)]
    loss, min1 = houv.Predict_loss(mv, tg)
    loss.mean().backward()
    np.savez_compressed(
        f"{OUT}/g2_loss.npz", moved=mv.detach().numpy(), target=tg.numpy(),
        cd_percent=np.stack([c.detach().numpy() for c in cdp]),
        views=np.stack([np.stack([c.detach().numpy() for c in v]) for v in lv]),
        loss=loss.detach().numpy(), min_1=min1.detach().numpy(), grad_moved=mv.grad.numpy())

    # ---- G3/G4: reset_weight params + forward for bases 0..3
    net = houv.HOUV(32, 0)
    net.reset_weight(32, 0, seed=2021)
    g3 = dict(V=net.V_c.detach().numpy(), angle=net.angle_c.detach().numpy(),
              tran_c=net.tran_c.detach().numpy(), tran_s=net.tran_s.detach().numpy())
    srcs = torch.tensor(rng.standard_normal((32, 50, 3)).astype(np.float32) * 0.3)
    g4 = dict(src=srcs.numpy())
    for base in range(4):
        net.reset_weight(32, base, seed=2021)
        with torch.no_grad():
            st, R, T = net(srcs)
        g4[f"moved_b{base}"] = st.numpy()
        g4[f"R_b{base}"] = R.numpy()
        g4[f"T_b{base}"] = T.numpy()
    np.savez_compressed(f"{OUT}/g3_g4_params_forward.npz", **g3, **g4)

    # ---- G5: parameter trajectory after 1,2,5,20 steps (B=2,N=256,K=16)
    pairs = [synth_pair(rng, 256, 45) for _ in range(2)]
    s5 = torch.tensor(np.stack([p[0] for p in pairs]))
    t5 = torch.tensor(np.stack([p[1] for p in pairs]))
    g5 = dict(src=s5.numpy(), tgt=t5.numpy())
    for base in (0, 2):
        for steps in (1, 2, 5, 20):
            net = houv.HOUV(32, 0)
            m1, R, T = houv.predict_model(net, s5, t5, kernel=16, num_epochs=steps, angle_base=base)
            g5[f"b{base}_n{steps}_min1"] = m1.detach().numpy()
            g5[f"b{base}_n{steps}_R"] = R.detach().numpy()
            g5[f"b{base}_n{steps}_T"] = T.detach().numpy()
            g5[f"b{base}_n{steps}_V"] = net.V_c.detach().numpy()
            g5[f"b{base}_n{steps}_angle"] = net.angle_c.detach().numpy()
            g5[f"b{base}_n{steps}_tran_c"] = net.tran_c.detach().numpy()
            g5[f"b{base}_n{steps}_tran_s"] = net.tran_s.detach().numpy()
            if steps == 1:
                g5[f"b{base}_grad_V"] = net.V_c.grad.numpy().copy()
                g5[f"b{base}_grad_angle"] = net.angle_c.grad.numpy().copy()
                g5[f"b{base}_grad_tran_c"] = net.tran_c.grad.numpy().copy()
                g5[f"b{base}_grad_tran_s"] = net.tran_s.grad.numpy().copy()
    np.savez_compressed(f"{OUT}/g5_trajectory.npz", **g5)

    # ---- G6: solve_model end-to-end incl. a >=120 deg pair (retry stage) and solve(prefix='test')
    pl = [synth_pair(rng, 128, 40), synth_pair(rng, 128, 40)]
    big = synth_pair(rng, 128, 180)
    while np.degrees(np.arccos(np.clip((np.trace(big[2][:3, :3]) - 1) / 2, -1, 1))) < 120:
        big = synth_pair(rng, 128, 180)
    pl.append(big)
    s6 = torch.tensor(np.stack([p[0] for p in pl]))
    t6 = torch.tensor(np.stack([p[1] for p in pl]))
    pose6 = torch.tensor(np.stack([p[2] for p in pl]))
    net = houv.HOUV(3 * 16, 0)
    r_err, t_err, ans = houv.solve_model(net, s6, t6, pose6, kernel=16, num_epochs=30)
    # stage outputs too, to let the test find the retry set
    m1_0, R0, T0 = houv.predict_model(houv.HOUV(48, 0), s6, t6, kernel=16, num_epochs=30, angle_base=0)
    ans_test = houv.solve_model(houv.HOUV(48, 0), s6, t6, None, kernel=16, num_epochs=30, prefix='test')
    g6 = dict(src=s6.numpy(), tgt=t6.numpy(), pose=pose6.numpy(), r_err=r_err.detach().numpy(),
              t_err=t_err.detach().numpy(), ans=ans.detach().numpy(), base0_min1=m1_0.detach().numpy(),
              ans_test=ans_test.detach().numpy())
    # functional twin: 500 iterations are hard-coded (train_utils.py:488) -> keep N,K tiny
    ps = [synth_pair(rng, 64, 40), synth_pair(rng, 64, 40)]
    s6b = torch.tensor(np.stack([p[0] for p in ps]))
    t6b = torch.tensor(np.stack([p[1] for p in ps]))
    np.random.seed(7)
    ans_solve = train_utils.solve(s6b, t6b, kernel=4, prefix='test')
    g6.update(solve_src=s6b.numpy(), solve_tgt=t6b.numpy(), solve_ans=ans_solve.detach().numpy(),
              solve_np_seed=np.int64(7))
    # and a short-horizon getPredict_angle trace (20 its) for the step-level ladder
    np.random.seed(11)
    m1, R, T, ts = train_utils.getPredict_angle(s6b, t6b, kernel=4, num_epochs=20, angle_base=1)
    g6.update(gpa_min1=m1.detach().numpy(), gpa_R=R.detach().numpy(), gpa_T=T.detach().numpy(),
              gpa_np_seed=np.int64(11))
    np.savez_compressed(f"{OUT}/g6_solve.npz", **g6)

    # ---- G7: SVDHead with/without weights incl. a det<0 case
    class A:  # SVDHead only reads these two attributes (model_utils.py:216)
        use_fpfh = False
        descriptor_size = 512
    head = model_utils.SVDHead(A())
    src7 = torch.tensor(rng.standard_normal((6, 3, 40)).astype(np.float32))
    Rg = []
    for i in range(6):
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] *= -1
        Rg.append(q)
    Rg = torch.tensor(np.stack(Rg).astype(np.float32))
    tg7 = torch.tensor(rng.standard_normal((6, 3, 1)).astype(np.float32))
    corr7 = Rg @ src7 + tg7 + 0.01 * torch.tensor(rng.standard_normal((6, 3, 40)).astype(np.float32))
    # force reflections: mirrored (planar-ish, noisy) correspondences
    corr7[4] = torch.tensor(np.diag([1.0, 1.0, -1.0]).astype(np.float32)) @ src7[4]
    src7[5, 2] *= 1e-3
    corr7[5] = torch.tensor(np.diag([1.0, -1.0, 1.0]).astype(np.float32)) @ src7[5] + 0.02 * torch.tensor(
        rng.standard_normal((3, 40)).astype(np.float32))
    w7 = torch.tensor(rng.random((6, 1, 40)).astype(np.float32))
    w7 = w7 / w7.sum(dim=2, keepdim=True)
    with torch.no_grad():
        R_a, t_a = head(src7, corr7)
        R_w, t_w = head(src7, corr7, w7)
    np.savez_compressed(f"{OUT}/g7_svdhead.npz", src=src7.numpy(), corr=corr7.numpy(), w=w7.numpy(),
                        R=R_a.numpy(), t=t_a.numpy(), R_w=R_w.numpy(), t_w=t_w.numpy(),
                        det=np.array([float(torch.det(r)) for r in R_a]))

    # ---- G8: metrics
    Ra = torch.tensor(np.stack([synth_pair(rng, 8, 180)[2] for _ in range(16)]))
    Rb = torch.tensor(np.stack([synth_pair(rng, 8, 180)[2] for _ in range(16)]))
    Rb[0] = Ra[0]
    pts = torch.tensor(rng.standard_normal((16, 33, 3)).astype(np.float32))
    np.savez_compressed(
        f"{OUT}/g8_metrics.npz", Ta=Ra.numpy(), Tb=Rb.numpy(), pts=pts.numpy(),
        rot_err=train_utils.rotation_error(Ra[:, :3, :3], Rb[:, :3, :3]).numpy(),
        trans_err=train_utils.translation_error(Ra[:, :3, 3], Rb[:, :3, 3]).numpy(),
        rmse=train_utils.rmse_loss(pts, Ra, Rb).numpy())
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
