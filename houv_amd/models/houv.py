"""Mirror of registration/models/houv.py: ``HOUV`` (:13-103), ``predict_model`` (:106-138), ``solve_model``
(:142-206), ``Predict_loss`` (:209-222) -- same names, signatures and return shapes -- with the optimisation
loop running in the fused gfx950 kernel (houv_solve_iterate) instead of 200 autograd iterations."""
import numpy as np
import torch
import torch.nn as nn

from .. import solver
from ..model_utils_completion import calc_cd_percent, loss_view
from ..train_utils import rotation_error, translation_error


class HOUV(nn.Module):
    """The 8 unconstrained scalars per hypothesis: rotation axis V_c[3], angle_c, translation direction tran_c[3],
    magnitude tran_s.  ``forward`` stays differentiable torch (houv.py:94-103) so user code that back-propagates
    through it keeps working; ``predict_model`` does not go through it."""

    def __init__(self, batch_size, angle_base):
        super(HOUV, self).__init__()
        self.batch_size = batch_size
        self.angle_base = angle_base
        self.pi = torch.acos(torch.zeros(1)).item() * 2
        # houv.py:21-36: unseeded draws; lattice rows only while they fit (`if num >= batch_size: continue`)
        vc = np.random.randn(batch_size, 3)
        nl = min(26, batch_size)
        vc[:nl] = solver.LATTICE_AXES[:nl]
        self.V_c = nn.Parameter(torch.from_numpy(vc.astype(np.float32)))
        self.angle_c = nn.Parameter(torch.from_numpy(np.random.randn(batch_size, 1).astype(np.float32)))
        self.tran_c = nn.Parameter(torch.from_numpy(np.random.randn(batch_size, 3).astype(np.float32)))
        # quirk kept (houv.py:36,61,99): __init__ creates tran_s_cpu, forward reads tran_s (made by reset_weight)
        self.tran_s_cpu = nn.Parameter(torch.from_numpy(np.random.randn(batch_size, 1).astype(np.float32)))

    @classmethod
    def blank_like(cls, other):
        """A module of the same kind WITHOUT touching the global numpy RNG (the constructor draws from it, houv.py:21-36):
        the concurrent retry stages of solve_model each need their own parameter holder."""
        self = cls.__new__(cls)
        nn.Module.__init__(self)
        self.batch_size, self.angle_base, self.pi = other.batch_size, other.angle_base, other.pi
        dev = other.V_c.device
        for name, width in (("V_c", 3), ("angle_c", 1), ("tran_c", 3), ("tran_s_cpu", 1)):
            setattr(self, name, nn.Parameter(torch.zeros((1, width), device=dev)))
        return self

    def reset_weight(self, batch_size, angle_base, seed=2021):
        self.batch_size = batch_size
        self.angle_base = angle_base
        p = solver.houv_init_params(batch_size, seed).astype(np.float32)
        dev = self.V_c.device
        self.V_c = nn.Parameter(torch.from_numpy(p[:, 0:3].copy()).to(dev))
        self.angle_c = nn.Parameter(torch.from_numpy(p[:, 3:4].copy()).to(dev))
        self.tran_c = nn.Parameter(torch.from_numpy(p[:, 4:7].copy()).to(dev))
        self.tran_s = nn.Parameter(torch.from_numpy(p[:, 7:8].copy()).to(dev))

    def packed_params(self):
        """[n,8] = (V, angle, tran_c, tran_s), the kernel's parameter block."""
        return torch.cat([self.V_c, self.angle_c, self.tran_c, self.tran_s], dim=1)

    def load_packed_params(self, p):
        with torch.no_grad():
            self.V_c.copy_(p[:, 0:3]); self.angle_c.copy_(p[:, 3:4]); self.tran_c.copy_(p[:, 4:7]); self.tran_s.copy_(p[:, 7:8])

    def cd_rotation(self, angle, V, device='cuda'):
        from ..train_utils import rotation
        return rotation(angle, V)

    def translation(self, tran, s):
        from ..train_utils import translation
        return translation(tran, s)

    def forward(self, src):
        src = src.squeeze(0)
        angle = torch.sin(self.angle_c * self.pi) * self.pi / 8 + self.pi / 8 + self.angle_base * self.pi / 4
        R = self.cd_rotation(angle, self.V_c)
        tran_s = torch.sin(self.tran_s * self.pi) * 0.125 + 0.125
        T = self.translation(self.tran_c, tran_s)
        return torch.bmm(src, R.transpose(1, 2)) + T, R, T


def predict_model(net, src, src_rotated, pose=None, src_ori=None, tgt_ori=None, angle_t=None, label=None, kernel=64,
                  num_epochs=500, angle_base=0, device='cuda', seed=2021):
    """houv.py:106-138.  ``kernel`` restarts per pair; fresh seeded parameters and a fresh Adam(lr=0.01) per call;
    returns (min_1[B,K], R[B,K,3,3], T[B,K,3]) of the last forward.  The clouds are NOT replicated K-fold
    (the reference materialises K copies of both, :111-112); ``net`` ends up holding the post-optimisation
    parameters exactly as after the reference's ``optimizer.step()`` calls."""
    B = src.shape[0]
    if net.V_c.device != src.device:
        net.to(src.device)
    net.reset_weight(B * kernel, angle_base, seed=seed)
    out, state = solver.run_stage(src, src_rotated, net.packed_params().detach().double(), kernel, num_epochs,
                                  angle_base=angle_base, trans_mode=0, use_views=True, f64_params=False, lr=0.01)
    net.load_packed_params(state[:, :8].float())
    return (out["score"].reshape(B, kernel), out["R"].reshape(B, kernel, 3, 3), out["T"].reshape(B, kernel, 3))


def solve_model(net, src, src_rotated, pose=None, src_ori=None, tgt_ori=None, angle_t=None, label=None, kernel=64,
                num_epochs=200, prefix='train'):
    """houv.py:142-206: base-0 solve, retry of pairs with best min_1 > 0.030 at bases 1..3, ans[B,4,4]
    (row 3 all-zero), then (r_err, t_err, ans) or ``ans.cpu()`` for prefix == 'test'."""
    def stage(s, t, base):
        # the retry stages may run concurrently on side streams: bases 1 and 2 get their own parameter holder, base 3
        # (the last one the reference runs) leaves its parameters in `net` like the reference does
        holder = net if base in (0, 3) else HOUV.blank_like(net)
        return predict_model(holder, s, t, kernel=kernel, num_epochs=num_epochs, angle_base=base)

    ans, _, _ = solver.best_of_k_with_retry(stage, src, src_rotated)
    if prefix == 'test':
        return ans.cpu()
    r_err = rotation_error(ans[:, :3, :3], pose[:, :3, :3])
    t_err = translation_error(ans[:, :3, 3], pose[:, :3, 3])
    print("Rotation error:", r_err.mean(), "Translation error:", t_err.mean())
    return r_err, t_err, ans


def Predict_loss(src, src_rotated, alpha=0.5):
    """houv.py:209-222, un-fused differentiable form: 6*min(cd pair at percent alpha) + the three view terms."""
    cd_t, cd_p = calc_cd_percent(src, src_rotated, percent=alpha)
    min_1, _ = torch.min(torch.cat([cd_t.unsqueeze(1), cd_p.unsqueeze(1)], dim=1), dim=1)
    views = 0
    for d in range(3):
        a, b = loss_view(src, src_rotated, dim=d)
        v, _ = torch.min(torch.cat([a.unsqueeze(1), b.unsqueeze(1)], dim=1), dim=1)
        views = views + v
    return min_1 * 6 + views, min_1
