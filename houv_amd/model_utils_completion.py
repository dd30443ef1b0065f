"""Loss glue on top of ``metrics.cd``: the hot-path slice of registration/model_utils_completion.py
(calc_cd :69-80, calc_cd_percent :83-100, calc_cd_percent_aligned :103-117, loss_view :157-166).
These are the un-fused, differentiable forms (one Chamfer launch + torch.topk each); the optimisation
loop itself never calls them -- it runs inside houv_solve_iterate."""
import torch

from .metrics import cd


def calc_cd(output, gt, calc_f1=False):
    if calc_f1:
        raise NotImplementedError("fscore is a completion-net metric outside the HOUV hot path")
    dist1, dist2, _, _ = cd()(gt, output)
    cd_p = (torch.sqrt(dist1).mean(1) + torch.sqrt(dist2).mean(1)) / 2
    cd_t = dist1.mean(1) + dist2.mean(1)
    return cd_p, cd_t


def calc_cd_percent(output, gt, calc_f1=False, percent=1):
    if calc_f1:
        raise NotImplementedError("fscore is a completion-net metric outside the HOUV hot path")
    k = int(output.shape[1] * percent)
    dist1, dist2, _, _ = cd()(gt, output)
    dist1, _ = dist1.topk(k, dim=1, largest=False, sorted=True)
    dist2, _ = dist2.topk(k, dim=1, largest=False, sorted=True)
    return torch.sqrt(dist1).mean(1), torch.sqrt(dist2).mean(1)


def calc_cd_percent_aligned(output, gt, percent=1):
    k = int(output.shape[1] * percent)
    dist1, dist2, idx1, idx2 = cd()(gt, output)
    dist1, idxx1 = dist1.topk(k, dim=1, largest=False, sorted=True)
    dist2, idxx2 = dist2.topk(k, dim=1, largest=False, sorted=True)
    return torch.sqrt(dist1).mean(1), torch.sqrt(dist2).mean(1), idx1, idx2, idxx1, idxx2


def loss_view(src, tgt, dim=0, percent=1):
    keep = torch.ones((1, 1, 3), dtype=src.dtype, device=src.device)
    keep[:, :, dim] = 0
    return calc_cd_percent(src * keep, tgt * keep, percent=percent)
