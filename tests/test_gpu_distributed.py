"""Multi-GPU readiness without a multi-GPU node (VERDICT r1 #4a): world_size 2 over gloo, BOTH ranks on cuda:0, each
running the REAL solve_model on its contiguous shard; the transforms gathered by houv_amd.distributed must equal the
single-process result bit for bit.  (The production path differs only in the backend: RCCL's all_gather_into_tensor on
device tensors instead of gloo's all_gather on host tensors; the sharding, padding and re-assembly code is the same.)"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rank_main(rank, world, port, q, src_np, tgt_np, K, iters, mode):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from houv_amd import distributed as hd
        from houv_amd.models.houv import HOUV, solve_model
        dev = torch.device("cuda:0")

        def solve_fn(s, t):            # prefix='test' returns the [n,4,4] answer on the host (houv.py:199-200)
            return solve_model(HOUV(s.shape[0] * K, 0), s.to(dev), t.to(dev), None, kernel=K, num_epochs=iters, prefix='test')

        full = hd.solve_sharded(solve_fn, torch.from_numpy(src_np), torch.from_numpy(tgt_np), mode=mode)
        q.put((rank, hd.shard_indices(len(src_np), rank, world, mode).tolist(), full.numpy(), None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover - reported by the parent
        import traceback
        q.put((rank, None, None, traceback.format_exc() + repr(e)))


@pytest.mark.parametrize("n_pairs,mode", [(6, "interleaved"), (5, "interleaved"), (5, "contiguous")])
def test_world2_real_solve_gathers_the_single_process_result(n_pairs, mode):
    import multiprocessing as mp
    from houv_amd import distributed as hd
    from houv_amd import synthetic
    from houv_amd.models.houv import HOUV, solve_model
    K, iters, world = 26, 40, 2
    src, tgt, _ = synthetic.make_pairs(n_pairs, 192, seed=515)
    ctx = mp.get_context("forkserver")              # started in conftest.py before this process touched the GPU
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 1500) + n_pairs + (40 if mode == 'contiguous' else 0)
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, q, src.numpy(), tgt.numpy(), K, iters, mode)) for r in range(world)]
    for p in procs:
        p.start()
    # the single-process result: the same per-shard batches, solved one after the other (the loss scale 1/(B K) of
    # houv.py:124 is per batch, exactly as in the reference's one-process-per-shard runs)
    dev = torch.device("cuda:0")
    want = np.zeros((n_pairs, 4, 4), np.float32)
    for r in range(world):
        idx = hd.shard_indices(n_pairs, r, world, mode)
        if idx.numel():
            want[idx.numpy()] = solve_model(HOUV(len(idx) * K, 0), src[idx].to(dev), tgt[idx].to(dev), None, kernel=K,
                                            num_epochs=iters, prefix='test').numpy()
    got = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, rng, full, err in got:
        assert err is None, err
        np.testing.assert_array_equal(full, want)          # every rank holds ALL transforms, bit for bit
    assert sorted(i for g in got for i in g[1]) == list(range(n_pairs))
    assert all(p.exitcode == 0 for p in procs)
    assert np.all(want[:, 3, :] == 0) and np.abs(want[:, :3, :3]).max() <= 1.0 + 1e-5


def _rccl_main(q, port, src_np, tgt_np, K, iters):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch.distributed as dist
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)        # "nccl" IS RCCL on ROCm
        from houv_amd import distributed as hd
        from houv_amd.models.houv import HOUV, solve_model
        res = {}
        # (1) the collective itself on HBM tensors, both row orders, ragged shard
        ans = torch.zeros((5, 4, 4), device=dev)
        ans[:, :3, :] = torch.arange(60, dtype=torch.float32, device=dev).reshape(5, 3, 4)
        for mode in ("contiguous", "interleaved"):
            full = hd.gather_transforms(ans, 5, mode=mode)
            res["gather_" + mode] = bool(full.is_cuda and torch.equal(full, ans))
        # (2) solve_sharded: real solve_model on device tensors, transforms stay in HBM through all_gather_into_tensor
        src, tgt = torch.from_numpy(src_np).to(dev), torch.from_numpy(tgt_np).to(dev)

        def solve_fn(s, t):
            return solve_model(HOUV(s.shape[0] * K, 0), s, t, None, kernel=K, num_epochs=iters, prefix='test').to(dev)
        want = solve_fn(src, tgt)
        for mode in ("contiguous", "interleaved"):
            full = hd.solve_sharded(solve_fn, src, tgt, mode=mode)
            res["solve_" + mode] = bool(full.is_cuda and torch.equal(full, want))
        # (3) the other two collectives bench.py uses on device tensors
        t = torch.tensor([3.5], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        res["all_reduce"] = float(t.item()) == 3.5
        res["backend"] = dist.get_backend()
        torch.cuda.synchronize()
        dist.destroy_process_group()
        q.put((res, None))
    except Exception as e:  # pragma: no cover - reported by the parent
        import traceback
        q.put((None, traceback.format_exc() + repr(e)))


def test_rccl_process_group_on_one_gpu():
    """until a multi-GPU node runs it, execute the RCCL path at least at world size 1 on the one GPU there is:
    init_process_group("nccl", device_id=cuda:0), all_gather_into_tensor of device-resident transforms through
    gather_transforms / solve_sharded (both sharding modes), barrier and all_reduce -- in a child of the forkserver, so that a
    hung collective cannot hang the suite."""
    import multiprocessing as mp
    import socket
    from houv_amd import synthetic
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    src, tgt, _ = synthetic.make_pairs(5, 192, seed=77)
    ctx = mp.get_context("forkserver")
    q = ctx.Queue()
    proc = ctx.Process(target=_rccl_main, args=(q, port, src.numpy(), tgt.numpy(), 26, 30))
    proc.start()
    res, err = q.get(timeout=300)
    proc.join(timeout=60)
    assert err is None, err
    assert res["backend"] == "nccl"
    for k in ("gather_contiguous", "gather_interleaved", "solve_contiguous", "solve_interleaved", "all_reduce"):
        assert res[k] is True, (k, res)
    assert proc.exitcode == 0
