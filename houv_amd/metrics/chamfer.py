"""Chamfer distance operator: the interface of utils/metrics/CD/chamfer3D/dist_chamfer_3D.py:26-73 on top of
the gfx950 kernels (houv_chamfer_forward / houv_chamfer_backward)."""
import torch
from torch import nn
from torch.autograd import Function

from .. import ops


class _Chamfer3DModule:
    """Stands where the reference's JIT-built pybind module ``chamfer_3D`` stands
    (dist_chamfer_3D.py:12-16): same two entry points, same positional arguments, same 1/0 return."""

    @staticmethod
    def forward(xyz1, xyz2, dist1, dist2, idx1, idx2):
        return ops.chamfer_forward(xyz1, xyz2, dist1, dist2, idx1, idx2)

    @staticmethod
    def backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
        return ops.chamfer_backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2)


chamfer_3D = _Chamfer3DModule()


class chamfer_3DFunction(Function):
    """(xyz1[B,N,3], xyz2[B,M,3]) -> (dist1[B,N], dist2[B,M], idx1, idx2), differentiable in both clouds
    (dist_chamfer_3D.py:26-64).  Outputs are allocated directly on the device (the reference builds them
    on the host and copies four tensors per call, :32-41)."""

    @staticmethod
    def forward(ctx, xyz1, xyz2):
        B, N, _ = xyz1.size()
        M = xyz2.size(1)
        dev = xyz1.device
        dist1 = torch.empty((B, N), dtype=torch.float32, device=dev)
        dist2 = torch.empty((B, M), dtype=torch.float32, device=dev)
        idx1 = torch.empty((B, N), dtype=torch.int32, device=dev)
        idx2 = torch.empty((B, M), dtype=torch.int32, device=dev)
        chamfer_3D.forward(xyz1, xyz2, dist1, dist2, idx1, idx2)
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, dist2, idx1, idx2

    @staticmethod
    def backward(ctx, graddist1, graddist2, gradidx1, gradidx2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        gradxyz1 = torch.zeros_like(xyz1)
        gradxyz2 = torch.zeros_like(xyz2)
        chamfer_3D.backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1.contiguous(), graddist2.contiguous(), idx1, idx2)
        return gradxyz1, gradxyz2


class chamfer_3DDist(nn.Module):
    """``metrics.cd()`` (dist_chamfer_3D.py:67-73)."""

    def forward(self, input1, input2):
        return chamfer_3DFunction.apply(input1.contiguous(), input2.contiguous())
