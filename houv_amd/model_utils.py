"""Hot-path slice of registration/model_utils.py: ``SVDHead`` (:213-255) and ``nearest_neighbor`` (:33-37)."""
import torch
import torch.nn as nn

from . import ops


def nearest_neighbor(src, dst):
    """src, dst [3,N] / [3,M] -> (negated squared distance of the NN, its index), model_utils.py:33-37."""
    inner = -2 * torch.matmul(src.transpose(1, 0).contiguous(), dst)
    distances = -torch.sum(src ** 2, dim=0, keepdim=True).transpose(1, 0).contiguous() - inner - torch.sum(
        dst ** 2, dim=0, keepdim=True)
    return distances.topk(k=1, dim=-1)


class SVDHead(nn.Module):
    """Kabsch rigid solve.  ``forward(src[B,3,N], src_corr[B,3,N], weights[B,1,N]|None) -> (R[B,3,3], t[B,3])``.
    The per-sample Python loop of torch.svd calls (model_utils.py:232-240) is one HIP launch: 3x3 weighted
    covariance reduction + register-resident Jacobi SVD with the reference's reflection fix."""

    def __init__(self, args=None):
        super(SVDHead, self).__init__()
        if args is not None:
            self.emb_dims = 33 if getattr(args, "use_fpfh", False) else getattr(args, "descriptor_size", 512)
        self.reflect = nn.Parameter(torch.eye(3), requires_grad=False)
        self.reflect[2, 2] = -1

    def forward(self, src, src_corr, weights=None):
        w = None if weights is None else weights.contiguous().float()
        return ops.kabsch(src.contiguous().float(), src_corr.contiguous().float(), w)
