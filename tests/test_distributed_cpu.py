"""CPU, world_size 2, gloo: the sharding + single all-gather path of houv_amd.distributed (the N>1 path of bench.py)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, n_pairs, q, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from houv_amd import distributed as hd
    src = torch.arange(n_pairs, dtype=torch.float32).view(n_pairs, 1, 1).expand(n_pairs, 4, 3).contiguous()

    def fake_solve(s, t):
        # a deterministic stand-in for the per-shard solve: encodes the pair id into the transform
        n = s.shape[0]
        ans = torch.zeros((n, 4, 4))
        ans[:, :3, :3] = torch.eye(3) * (1 + s[:, 0, 0]).view(n, 1, 1)
        ans[:, :3, 3] = s[:, 0, :]
        return ans

    full = hd.solve_sharded(fake_solve, src, src, mode=mode)
    q.put((rank, hd.shard_indices(n_pairs, rank, world, mode).tolist(), full.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["interleaved", "contiguous"])
@pytest.mark.parametrize("n_pairs", [8, 7, 1])
def test_shard_and_allgather_world2(n_pairs, mode):
    from houv_amd import distributed as hd
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + n_pairs + (20 if mode == 'contiguous' else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(got[0][1] + got[1][1]) == list(range(n_pairs))                               # every pair exactly once
    if mode == "contiguous":
        ranges = sorted((g[1][0], g[1][-1] + 1) if g[1] else (n_pairs, n_pairs) for g in got)
        assert ranges[0][0] == 0 and ranges[-1][1] == n_pairs and ranges[0][1] == ranges[1][0]
    want = np.zeros((n_pairs, 4, 4), np.float32)
    for i in range(n_pairs):
        want[i, :3, :3] = np.eye(3) * (1 + i)
        want[i, :3, 3] = i
    for _, _, full in got:
        np.testing.assert_array_equal(full, want)          # every rank holds all transforms; row 3 stays zero


def test_shard_range_matches_reference_slices():
    from houv_amd.distributed import shard_range
    assert [shard_range(2000, r, 4) for r in range(4)] == [(0, 500), (500, 1000), (1000, 1500), (1500, 2000)]   # run_test.sh:6
    assert [shard_range(1200, r, 8) for r in range(8)][-1] == (1050, 1200)
    assert shard_range(3, 7, 8) == (3, 3)
    from houv_amd.distributed import shard_indices
    assert shard_indices(10, 1, 4).tolist() == [1, 5, 9] and shard_indices(10, 3, 4, "contiguous").tolist() == [9]
    assert shard_indices(2, 3, 4).tolist() == []
