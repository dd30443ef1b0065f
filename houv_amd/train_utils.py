"""Hot-path slice of registration/train_utils.py: the metrics (:82-95), the Rodrigues/translation helpers
(:113-148), and the functional HOUV twin ``getPredict_angle`` / ``solve`` (:359-456, :467-572) that the
reference's test drivers call (test.py:64, test_mult_modelnet.py:43).  The dead experimental variants
(getPredict_cd_keba* etc.) are deliberately absent."""
import math

import torch

from . import solver


class AverageValueMeter(object):
    """train_utils.py:22-36."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0.0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def rotation_error(R, R_gt):
    """Geodesic angle in degrees (train_utils.py:82-85)."""
    cos_theta = (torch.einsum('bij,bij->b', R, R_gt) - 1) / 2
    return torch.acos(torch.clamp(cos_theta, -1, 1)) * 180 / math.pi


def translation_error(t, t_gt):
    """train_utils.py:88-89."""
    return torch.norm(t - t_gt, dim=1)


def rmse_loss(pts, T, T_gt):
    """Mean point displacement between two rigid transforms (train_utils.py:92-95)."""
    pred = pts @ T[:, :3, :3].transpose(1, 2) + T[:, :3, 3].unsqueeze(1)
    gt = pts @ T_gt[:, :3, :3].transpose(1, 2) + T_gt[:, :3, 3].unsqueeze(1)
    return torch.norm(pred - gt, dim=2).mean(dim=1)


def rotation(angle, V, device='cuda'):
    """Rodrigues rotation from angle [n,1] and un-normalised axis [n,3] (train_utils.py:113-131); differentiable."""
    u = V / torch.sqrt((V * V).sum(dim=1, keepdim=True))
    zero = torch.zeros_like(u[:, 0])
    A = torch.stack([torch.stack([zero, -u[:, 2], u[:, 1]], 1),
                     torch.stack([u[:, 2], zero, -u[:, 0]], 1),
                     torch.stack([-u[:, 1], u[:, 0], zero], 1)], 1)
    eye = torch.eye(3, dtype=V.dtype, device=V.device).expand_as(A)
    return eye + torch.sin(angle).unsqueeze(2) * A + (1 - torch.cos(angle)).unsqueeze(2) * torch.bmm(A, A)


def translation(tran, s):
    """train_utils.py:144-148."""
    tran = tran / torch.sqrt((tran * tran).sum(dim=1, keepdim=True))
    return (tran * s).unsqueeze(1)


def getPredict_angle(src, src_rotated, pose=None, src_ori=None, tgt_ori=None, angle_t=None, label=None, kernel=64,
                     num_epochs=1000, angle_base=0):
    """train_utils.py:359-456 on the fused kernel: float64 leaves drawn from the global numpy RNG, Adam(lr=0.1)
    in float64, sigma = sin(s*pi), loss = 6*min_1 (no view terms).  Returns (min_1[B,K], R[B,K,3,3], T[B,K,3], tran_s)."""
    B = src.shape[0]
    n = B * kernel
    params = solver.solve_twin_init_params(n)
    out, state = solver.run_stage(src, src_rotated, params, kernel, num_epochs, angle_base=angle_base, trans_mode=1,
                                  use_views=False, f64_params=True, lr=0.1, want_last_params=True)
    pi = torch.acos(torch.zeros(1)).item() * 2
    # tran_s of the LAST forward (train_utils.py:404,456), i.e. from the parameters before the final optimizer.step()
    tran_s = torch.sin(out["last_params"][:, 7:8].float() * pi) * 1
    return (out["score"].reshape(B, kernel), out["R"].reshape(B, kernel, 3, 3), out["T"].reshape(B, kernel, 3), tran_s)


def solve(src, src_rotated, pose=None, src_ori=None, tgt_ori=None, angle_t=None, label=None, kernel=64, num_epochs=500,
          prefix='train', _iters=500):
    """train_utils.py:467-572.  The reference ignores ``num_epochs`` and hard-codes 500 iterations per stage
    (:488, :503); ``_iters`` exposes that constant (tests shorten it).  prefix == 'test' returns ans[B,4,4] on the
    host (:548-549); otherwise (r_err, t_err, ans)."""
    def stage(s, t, base):
        m1, R, T, _ = getPredict_angle(s, t, kernel=kernel, num_epochs=_iters, angle_base=base)
        return m1, R, T

    ans, score, _ = solver.best_of_k_with_retry(stage, src, src_rotated)
    if prefix == 'test':
        return ans.cpu()
    r_err = rotation_error(ans[:, :3, :3], pose[:, :3, :3])
    t_err = translation_error(ans[:, :3, 3], pose[:, :3, 3])
    print(r_err.mean(), t_err.mean(), score.min(dim=1)[0].max())
    return r_err, t_err, ans


def combine(src, tgt):
    """train_utils.py:459-464: concatenate both clouds and furthest-point-sample 2048 of them.  (The reference hands the
    [B,3,2N] transposed tensor to an op that expects [B,N,3]; the evident intent -- sampling points -- is what this does.)"""
    from .mm3d_pn2 import furthest_point_sample, gather_points
    pts = torch.cat([src, tgt], dim=1).contiguous()                   # [B, 2N, 3]
    data = pts.transpose(1, 2).contiguous()                           # [B, 3, 2N]
    sample_idx = furthest_point_sample(pts, 2048)
    return gather_points(data, sample_idx).transpose(1, 2).contiguous()
