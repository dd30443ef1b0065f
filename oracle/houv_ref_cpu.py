"""CPU oracle for the HOUV registration hot path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch PyTorch-CPU restatement of the reference algorithm
(Dizzy-cell/HOUV).  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  Nothing under ``houv_amd/`` imports it, and the product path never falls
back to it.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real
reference (CPU, with the stubs of SURVEY.md App. B) in the build container and
writes golden vectors G1..G8 to ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function below against them.

Each function cites the reference file:line (relative to the reference root)
whose arithmetic it restates.  All arithmetic follows the reference's CPU
path: Chamfer in float64 expanded form cast back to float32, everything else
float32, gradients by autograd, optimiser = torch.optim.Adam.
"""
import math

import numpy as np
import torch

# registration/models/houv.py:19 -- pi is acos(0)*2 evaluated in fp32 then widened.
PI = torch.acos(torch.zeros(1)).item() * 2

RETRY_THRESHOLD = 0.030  # houv.py:156 / train_utils.py:494


# --------------------------------------------------------------------------
# L1: Chamfer nearest-neighbour op
# --------------------------------------------------------------------------
def chamfer_nn(a, b):
    """Bidirectional squared NN distance + argmin.

    utils/metrics/CD/chamfer_python.py:18-39 (``distChamfer``): float64,
    ``|x|^2 + |y|^2 - 2 x.y``, ``min`` over each axis (lowest index on ties),
    results cast to float32 / int32.  Same contract as the CUDA op
    utils/metrics/CD/chamfer3D/chamfer3D.cu:12-154.

    a: [B,N,D]  b: [B,M,D]  ->  dist_a[B,N], dist_b[B,M], idx_a[B,N], idx_b[B,M]
    """
    x = a.double()
    y = b.double()
    sx = torch.pow(x, 2).sum(2)                       # [B,N]
    sy = torch.pow(y, 2).sum(2)                       # [B,M]
    cross = torch.bmm(x, y.transpose(2, 1))           # [B,N,M]
    P = sx.unsqueeze(2) + sy.unsqueeze(1) - 2 * cross
    da, ia = torch.min(P, 2)
    db, ib = torch.min(P, 1)
    return da.float(), db.float(), ia.int(), ib.int()


def chamfer_nn_chunked(a, b, chunk=8):
    """``chamfer_nn`` evaluated ``chunk`` batch rows at a time (the [B,N,M]
    float64 temp is 34 MB per instance at 2048x2048).  No autograd."""
    outs = [[], [], [], []]
    with torch.no_grad():
        for s in range(0, a.shape[0], chunk):
            r = chamfer_nn(a[s:s + chunk], b[s:s + chunk])
            for o, v in zip(outs, r):
                o.append(v)
    return tuple(torch.cat(o, 0) for o in outs)


def chamfer_backward_closed_form(xyz1, xyz2, idx1, idx2, g1, g2):
    """Closed form of NmDistanceGradKernel (chamfer3D.cu:155-174), both
    directions (chamfer3D.cu:184-185): for every query i with NN j:
    grad_q[i] += 2 g[i] (q_i - r_j);  grad_r[j] -= 2 g[i] (q_i - r_j)."""
    B, N, _ = xyz1.shape
    M = xyz2.shape[1]
    gx1 = torch.zeros_like(xyz1)
    gx2 = torch.zeros_like(xyz2)
    nn1 = torch.gather(xyz2, 1, idx1.long().unsqueeze(2).expand(B, N, 3))
    d1 = 2 * g1.unsqueeze(2) * (xyz1 - nn1)
    gx1 += d1
    gx2.scatter_add_(1, idx1.long().unsqueeze(2).expand(B, N, 3), -d1)
    nn2 = torch.gather(xyz1, 1, idx2.long().unsqueeze(2).expand(B, M, 3))
    d2 = 2 * g2.unsqueeze(2) * (xyz2 - nn2)
    gx2 += d2
    gx1.scatter_add_(1, idx2.long().unsqueeze(2).expand(B, M, 3), -d2)
    return gx1, gx2


# --------------------------------------------------------------------------
# L2: loss glue
# --------------------------------------------------------------------------
def calc_cd_percent(output, gt, percent=1.0):
    """registration/model_utils_completion.py:83-100.  Chamfer is called as
    (gt, output) (:89); k = int(output_points*percent) (:85-86); k smallest of
    each direction via sorted topk (:91-92); mean of sqrt (:94-95).
    Returns (cd over gt points, cd over output points)."""
    k = int(output.shape[1] * percent)
    d_gt, d_out, _, _ = chamfer_nn(gt, output)
    d_gt, _ = d_gt.topk(k, dim=1, largest=False, sorted=True)
    d_out, _ = d_out.topk(k, dim=1, largest=False, sorted=True)
    return torch.sqrt(d_gt).mean(1), torch.sqrt(d_out).mean(1)


def loss_view(src, tgt, dim=0, percent=1.0):
    """model_utils_completion.py:157-166: zero coordinate ``dim`` of both
    clouds (by multiplying with a 0/1 mask), then ``calc_cd_percent``."""
    keep = torch.ones((1, 1, 3), dtype=src.dtype)
    keep[:, :, dim] = 0
    mask = torch.zeros_like(src) + keep
    return calc_cd_percent(src * mask, tgt * mask, percent=percent)


def predict_loss(moved, target, alpha=0.5):
    """registration/models/houv.py:209-222 (``Predict_loss``):
    6*min(cd pair @percent alpha) + sum_d min(view-d cd pair); also returns the
    selection score min_1."""
    c0, c1 = calc_cd_percent(moved, target, percent=alpha)
    min_1 = torch.minimum(c0, c1)
    total = min_1 * 6
    views = 0
    for d in range(3):
        v0, v1 = loss_view(moved, target, dim=d)
        views = views + torch.minimum(v0, v1)
    return total + views, min_1


# --------------------------------------------------------------------------
# L3: HOUV parameterisation
# --------------------------------------------------------------------------
LATTICE_AXES = np.array([(x, y, z) for x in (-1, 0, 1) for y in (-1, 0, 1)
                         for z in (-1, 0, 1) if (x, y, z) != (0, 0, 0)], dtype=np.float64)


def houv_init_params(n_inst, seed=2021):
    """houv.py:40-61 (``reset_weight``): every tensor is drawn right after
    re-seeding numpy with the same seed; axis rows 0..25 are overwritten with
    the 26 lattice directions (no bounds check: n_inst < 26 raises, :47-51).
    Returns float32 arrays V[n,3], angle[n,1], tran_c[n,3], tran_s[n,1]."""
    np.random.seed(seed)
    V = np.random.randn(n_inst, 3)
    V[:26] = LATTICE_AXES          # IndexError-equivalent when n_inst < 26
    np.random.seed(seed)
    ang = np.random.randn(n_inst, 1)
    np.random.seed(seed)
    tc = np.random.randn(n_inst, 3)
    np.random.seed(seed)
    ts = np.random.randn(n_inst, 1)
    return (V.astype(np.float32), ang.astype(np.float32),
            tc.astype(np.float32), ts.astype(np.float32))


def rodrigues(theta, V):
    """houv.py:69-86 / train_utils.py:113-131: u = V/|V|; A = [u]x;
    R = I + sin(theta) A + (1-cos(theta)) A A.  theta [n,1], V [n,3]."""
    n = theta.shape[0]
    u = V / torch.sqrt((V * V).sum(dim=1, keepdim=True))
    A = torch.zeros((n, 3, 3), dtype=V.dtype)
    A[:, 0, 1] = -u[:, 2]
    A[:, 0, 2] = u[:, 1]
    A[:, 1, 0] = u[:, 2]
    A[:, 1, 2] = -u[:, 0]
    A[:, 2, 0] = -u[:, 1]
    A[:, 2, 1] = u[:, 0]
    eye = torch.zeros((n, 3, 3), dtype=V.dtype) + torch.eye(3, dtype=V.dtype)
    return eye + torch.sin(theta).unsqueeze(2) * A + (1 - torch.cos(theta)).unsqueeze(2) * torch.bmm(A, A)


def houv_forward(src, V, ang, tc, ts, angle_base, trans_mode="houv"):
    """houv.py:94-103 (``HOUV.forward``) for trans_mode="houv":
      theta = sin(a pi) pi/8 + pi/8 + base pi/4 ; sigma = sin(s pi)/8 + 1/8
    train_utils.py:403-404 (``getPredict_angle``) for trans_mode="solve":
      sigma = sin(s pi) * 1
    T = sigma * c/|c| ; moved = src @ R^T + T.  Returns (moved, R, T[n,1,3])."""
    theta = torch.sin(ang * PI) * PI / 8 + PI / 8 + angle_base * PI / 4
    R = rodrigues(theta, V)
    if trans_mode == "houv":
        sigma = torch.sin(ts * PI) * 0.125 + 0.125
    else:
        sigma = torch.sin(ts * PI) * 1
    T = (tc / torch.sqrt((tc * tc).sum(dim=1, keepdim=True)) * sigma).unsqueeze(1)
    moved = torch.bmm(src, R.transpose(1, 2)) + T
    return moved, R, T


def _replicate(x, kernel):
    n = x.shape[1]
    return x.unsqueeze(1).expand((-1, kernel, -1, -1)).reshape(-1, n, 3)


def predict_model(src, tgt, kernel=64, num_epochs=500, angle_base=0, seed=2021, lr=0.01,
                  trace=None):
    """houv.py:106-138 (``predict_model``).  K-fold replicate both clouds
    (:111-112), fresh params + fresh Adam(lr) (:116-118), ``num_epochs`` x
    {forward, Predict_loss, mean, backward, step} (:120-126).  Outputs are those
    of the LAST forward (the last step is not observed, :134-136).
    ``trace``: optional list receiving a dict of params after selected steps."""
    B = src.shape[0]
    s = _replicate(src, kernel)
    t = _replicate(tgt, kernel)
    V, ang, tc, ts = [torch.nn.Parameter(torch.from_numpy(p)) for p in houv_init_params(B * kernel, seed)]
    opt = torch.optim.Adam([V, ang, tc, ts], lr=lr)
    loss_i = min_1 = R = T = None
    for it in range(num_epochs):
        opt.zero_grad()
        moved, R, T = houv_forward(s, V, ang, tc, ts, angle_base, "houv")
        loss_i, min_1 = predict_loss(moved, t)
        loss_i.mean().backward()
        if trace is not None and trace.get("want_grads") and it == 0:
            trace["grads0"] = [p.grad.detach().clone().numpy() for p in (V, ang, tc, ts)]
        opt.step()
        if trace is not None and (it + 1) in trace.get("steps", ()):
            trace.setdefault("params", {})[it + 1] = [p.detach().clone().numpy() for p in (V, ang, tc, ts)]
            trace.setdefault("loss", {})[it + 1] = loss_i.detach().clone().numpy()
            trace.setdefault("min_1", {})[it + 1] = min_1.detach().clone().numpy()
    return (min_1.detach().reshape(B, kernel), R.detach().reshape(B, kernel, 3, 3),
            T.detach().reshape(B, kernel, 3))


def _solve_driver(stage_fn, src, tgt):
    """Shared best-of-K + retry logic of houv.py:152-197 and
    train_utils.py:488-545: base-0 stage; pairs whose best score > 0.030 are
    re-solved at bases 1,2,3 and replaced where strictly better; ans[B,4,4]
    keeps row 3 all-zero."""
    B = src.shape[0]
    score, R, T = stage_fn(src, tgt, 0)
    best, _ = score.topk(1, dim=1, largest=False, sorted=True)
    retry = [j for j in range(B) if best[j][0] > RETRY_THRESHOLD]
    if len(retry) > 0:
        retry = np.array(retry).astype(int)
        s_add, t_add = src[retry], tgt[retry]
        for base in range(1, 4):
            score_a, R_a, T_a = stage_fn(s_add, t_add, base)
            best_a, _ = score_a.topk(1, dim=1, largest=False, sorted=True)
            flag = torch.nonzero((best_a < best[retry]).reshape(-1)).reshape(-1)
            ge = retry[flag.int().numpy()]
            R[ge] = R_a[flag]
            score[ge] = score_a[flag]
            T[ge] = T_a[flag]
            best[ge] = best_a[flag]
    ans = torch.zeros((B, 4, 4))
    for i in range(B):
        _, k = score[i].topk(1, dim=0, largest=False, sorted=True)
        ans[i, :3, :3] = R[i][k]
        ans[i, :3, 3] = T[i][k]
    return ans, score, retry


def solve_model(src, tgt, pose=None, kernel=64, num_epochs=200, prefix="train"):
    """houv.py:142-206 (``solve_model``)."""
    ans, score, retry = _solve_driver(
        lambda s, t, base: predict_model(s, t, kernel=kernel, num_epochs=num_epochs, angle_base=base),
        src, tgt)
    if prefix == "test":
        return ans
    r_err = rotation_error(ans[:, :3, :3], pose[:, :3, :3])
    t_err = translation_error(ans[:, :3, 3], pose[:, :3, 3])
    return r_err, t_err, ans


def get_predict_angle(src, tgt, kernel=64, num_epochs=500, angle_base=0, trace=None):
    """train_utils.py:359-456 (``getPredict_angle``): float64 host parameters
    drawn from the *global, unseeded* numpy RNG in the order V, angle, tran_c,
    tran_s, angle_XYZ (:381-386; the last is never used but consumes RNG
    state); Adam(lr=0.1) on the float64 leaves (:388-389); each iteration casts
    them to float32 (:397-401); sigma = sin(s pi) (:404); loss = 6*min_1 only
    (:433)."""
    B = src.shape[0]
    s = _replicate(src, kernel)
    t = _replicate(tgt, kernel)
    n = B * kernel
    V = torch.from_numpy(np.random.randn(n, 3)).requires_grad_(True)
    ang = torch.from_numpy(np.random.randn(n, 1)).requires_grad_(True)
    tc = torch.from_numpy(np.random.randn(n, 3)).requires_grad_(True)
    ts = torch.from_numpy(np.random.randn(n, 1)).requires_grad_(True)
    xyz = torch.from_numpy(np.random.randn(n, 3)).requires_grad_(True)
    opt = torch.optim.Adam([V, ang, tc, ts, xyz], lr=0.1)
    min_1 = R = T = None
    for it in range(num_epochs):
        opt.zero_grad()
        moved, R, T = houv_forward(s, V.float(), ang.float(), tc.float(), ts.float(), angle_base, "solve")
        c0, c1 = calc_cd_percent(moved, t, percent=0.5)
        min_1 = torch.minimum(c0, c1)
        (min_1 * 6).mean().backward()
        opt.step()
        if trace is not None and (it + 1) in trace.get("steps", ()):
            trace.setdefault("params", {})[it + 1] = [p.detach().clone().numpy() for p in (V, ang, tc, ts)]
    return (min_1.detach().reshape(B, kernel), R.detach().reshape(B, kernel, 3, 3),
            T.detach().reshape(B, kernel, 3))


def solve(src, tgt, pose=None, kernel=64, num_epochs=500, prefix="train", _iters=500):
    """train_utils.py:467-572 (``solve``).  NB the reference ignores its own
    ``num_epochs`` argument and hard-codes 500 (:488,:503); ``_iters`` exposes
    that constant so tests can shorten it."""
    ans, score, retry = _solve_driver(
        lambda s, t, base: get_predict_angle(s, t, kernel=kernel, num_epochs=_iters, angle_base=base),
        src, tgt)
    if prefix == "test":
        return ans
    r_err = rotation_error(ans[:, :3, :3], pose[:, :3, :3])
    t_err = translation_error(ans[:, :3, 3], pose[:, :3, 3])
    return r_err, t_err, ans


# --------------------------------------------------------------------------
# metrics (train_utils.py:82-95)
# --------------------------------------------------------------------------
def rotation_error(R, R_gt):
    """train_utils.py:82-85."""
    c = (torch.einsum('bij,bij->b', R, R_gt) - 1) / 2
    return torch.acos(torch.clamp(c, -1, 1)) * 180 / math.pi


def translation_error(t, t_gt):
    """train_utils.py:88-89."""
    return torch.norm(t - t_gt, dim=1)


def rmse_loss(pts, T, T_gt):
    """train_utils.py:92-95."""
    a = pts @ T[:, :3, :3].transpose(1, 2) + T[:, :3, 3].unsqueeze(1)
    b = pts @ T_gt[:, :3, :3].transpose(1, 2) + T_gt[:, :3, 3].unsqueeze(1)
    return torch.norm(a - b, dim=2).mean(dim=1)


# --------------------------------------------------------------------------
# L1': Kabsch (registration/model_utils.py:213-255, SVDHead.forward)
# --------------------------------------------------------------------------
def kabsch_svd(src, corr, weights=None):
    """src, corr [B,3,N]; weights [B,1,N] or None -> R[B,3,3], t[B,3].
    Centre by the UNWEIGHTED mean (:221-222); H = (src_c*w) corr_c^T (:224-227);
    per sample svd, r = v u^T, if det(r)<0 flip the last column of v (:232-240);
    t = -R mean(src) + mean(corr), or weighted sums when weights given
    (:251-254)."""
    sc = src - src.mean(dim=2, keepdim=True)
    cc = corr - corr.mean(dim=2, keepdim=True)
    if weights is None:
        H = torch.matmul(sc, cc.transpose(2, 1))
    else:
        H = torch.matmul(sc * weights, cc.transpose(2, 1))
    reflect = torch.eye(3)
    reflect[2, 2] = -1
    Rs = []
    for i in range(src.shape[0]):
        u, s, v = torch.svd(H[i])
        r = v @ u.t()
        if torch.det(r) < 0:
            r = (v @ reflect) @ u.t()
        Rs.append(r)
    R = torch.stack(Rs, 0)
    if weights is None:
        t = torch.matmul(-R, src.mean(dim=2, keepdim=True)) + corr.mean(dim=2, keepdim=True)
    else:
        t = torch.matmul(-R, (weights * src).sum(dim=2, keepdim=True)) + (weights * corr).sum(dim=2, keepdim=True)
    return R, t.view(src.shape[0], 3)
