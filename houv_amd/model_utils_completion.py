"""Loss glue on top of ``metrics.cd``: the hot-path slice of registration/model_utils_completion.py
(calc_cd :69-80, calc_cd_percent :83-100, calc_cd_percent_aligned :103-117, loss_view :157-166).
These are the un-fused, differentiable forms (one Chamfer launch + torch.topk each); the optimisation
loop itself never calls them -- it runs inside houv_solve_iterate."""
import torch

from .metrics import cd


def _f1(dist1, dist2, threshold=0.0001):
    """F-score of two clouds from their SQUARED nearest-neighbour distances (utils/metrics/CD/fscore.py:3-16): the harmonic mean of
    the two fractions of points closer than `threshold`; 0 where both fractions are 0."""
    near1 = (dist1 < threshold).to(dist1.dtype).mean(dim=1)
    near2 = (dist2 < threshold).to(dist2.dtype).mean(dim=1)
    both = near1 + near2
    return torch.where(both > 0, 2 * near1 * near2 / both.clamp_min(torch.finfo(dist1.dtype).tiny), torch.zeros_like(both))


def calc_cd(output, gt, calc_f1=False):
    dist1, dist2, _, _ = cd()(gt, output)
    cd_p = (torch.sqrt(dist1).mean(1) + torch.sqrt(dist2).mean(1)) / 2
    cd_t = dist1.mean(1) + dist2.mean(1)
    return (cd_p, cd_t, _f1(dist1, dist2)) if calc_f1 else (cd_p, cd_t)


def calc_cd_percent(output, gt, calc_f1=False, percent=1):
    k = int(output.shape[1] * percent)
    dist1, dist2, _, _ = cd()(gt, output)
    dist1, _ = dist1.topk(k, dim=1, largest=False, sorted=True)
    dist2, _ = dist2.topk(k, dim=1, largest=False, sorted=True)
    cd_p, cd_t = torch.sqrt(dist1).mean(1), torch.sqrt(dist2).mean(1)
    return (cd_p, cd_t, _f1(dist1, dist2)) if calc_f1 else (cd_p, cd_t)      # the F-score of the KEPT distances, as :96-98 has it


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
