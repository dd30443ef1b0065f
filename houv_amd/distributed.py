"""Multi-GPU sharding of registration pairs: one process per GPU, contiguous pair ranges, local solve, and ONE
all-gather of the per-pair (R, t) -- 12 floats per pair -- over RCCL/xGMI (torch.distributed backend "nccl" is
RCCL on ROCm; "gloo" is used by the CPU tests).  Replaces the reference's 4 shell processes + ``{l}_{r}.npy`` files
+ ``sleep 600`` + ``--combine`` (registration/run_test.sh:6-23, test_mult.py:70-81, dataset.py:369-372).

Pairs are independent, so there is no data-path collective during the solve."""
import torch
import torch.distributed as dist


def shard_range(n_pairs, rank, world_size):
    """Contiguous [l, r) with r - l = ceil(n_pairs / world_size), like the reference's -l/-r slices
    (run_test.sh:6: 2000 pairs -> 4 x 500)."""
    per = -(-n_pairs // world_size)
    l = min(rank * per, n_pairs)
    return l, min(l + per, n_pairs)


def gather_transforms(ans_local, n_pairs, group=None):
    """ans_local [n_local,4,4] (this rank's contiguous shard) -> [n_pairs,4,4] on every rank, by a single
    all_gather of a [per,12] fp32 block per rank (rows beyond a short last shard are padding)."""
    if not (dist.is_available() and dist.is_initialized()):
        return ans_local
    world = dist.get_world_size(group)
    per = -(-n_pairs // world)
    dev = ans_local.device
    block = torch.zeros((per, 12), dtype=torch.float32, device=dev)
    n_local = ans_local.shape[0]
    if n_local:
        block[:n_local] = ans_local[:, :3, :].reshape(n_local, 12)      # R|t rows; row 3 is all-zero by construction
    out = torch.empty((world * per, 12), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(out, block, group=group) if dev.type == "cuda" else _gather_cpu(out, block, world, group)
    full = torch.zeros((n_pairs, 4, 4), dtype=torch.float32, device=dev)
    full[:, :3, :] = out[:n_pairs].reshape(n_pairs, 3, 4)
    return full


def _gather_cpu(out, block, world, group):
    parts = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(parts, block, group=group)
    out.copy_(torch.cat(parts, 0))


def solve_sharded(solve_fn, src, tgt, group=None):
    """Run ``solve_fn(src_shard, tgt_shard) -> ans[n,4,4]`` on this rank's shard of the pair list and gather the
    transforms of all shards.  ``src``/``tgt`` hold ALL pairs (or anything sliceable by [l:r])."""
    n = src.shape[0]
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    l, r = shard_range(n, rank, world)
    if r > l:
        ans = solve_fn(src[l:r], tgt[l:r])
    else:
        ans = torch.zeros((0, 4, 4), dtype=torch.float32, device=src.device)
    return gather_transforms(ans, n, group)
