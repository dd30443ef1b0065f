"""ICP refinement on top of the HOUV solve (BASELINE configs[3] "HOUV + ICP refine (train_ICP.py path)").

``icp_refine`` mirrors the Open3D call of registration/train_ICP.py:148-151 (threshold 0.02, point-to-point,
max_iteration 500) as one batched HIP launch; ``solve_model_icp`` chains it after ``solve_model`` -- the reference
runs ICP from a fixed tutorial initialisation (:138-141) and leaves the HOUV->ICP chaining commented out (:119)."""
import torch

from . import ops
from .models.houv import solve_model
from .train_utils import rotation_error, translation_error

ICP_THRESHOLD = 0.02          # train_ICP.py:136
ICP_MAX_ITERATION = 500       # train_ICP.py:151


def icp_refine(src, tgt, init=None, threshold=ICP_THRESHOLD, max_iteration=ICP_MAX_ITERATION):
    """src[B,N,3], tgt[B,M,3], init[B,4,4]|None -> T[B,4,4] mapping src into tgt's frame (bottom row 0,0,0,1)."""
    if init is not None:
        init = init.to(src.device).float().contiguous().clone()
        init[:, 3, :] = torch.tensor([0.0, 0.0, 0.0, 1.0], device=src.device)     # HOUV's ans keeps row 3 all-zero
    return ops.icp_refine(src.contiguous().float(), tgt.contiguous().float(), init, threshold, max_iteration)["T"]


def solve_model_icp(net, src, src_rotated, pose=None, kernel=64, num_epochs=200, threshold=ICP_THRESHOLD,
                    max_iteration=ICP_MAX_ITERATION, prefix='train'):
    """HOUV solve, then ICP refinement started from HOUV's answer."""
    if prefix == 'test':
        ans = solve_model(net, src, src_rotated, None, kernel=kernel, num_epochs=num_epochs, prefix='test').to(src.device)
    else:
        _, _, ans = solve_model(net, src, src_rotated, pose, kernel=kernel, num_epochs=num_epochs)
    T = icp_refine(src, src_rotated, ans, threshold, max_iteration)
    if prefix == 'test':
        return T.cpu()
    return rotation_error(T[:, :3, :3], pose[:, :3, :3]), translation_error(T[:, :3, 3], pose[:, :3, 3]), T
