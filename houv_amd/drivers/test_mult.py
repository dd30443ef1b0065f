"""``python -m houv_amd.drivers.test_mult -c cfg.yaml [-l L -r R] [--combine True]`` -- the sharded test driver
(registration/test_mult.py:83-125 for the flags, test_mult_modelnet.py:27-54 for the intact loop, run_test.sh for the job).
The shard [l, r) of MVP_ExtraTest_RG.h5 comes through MVP_RG_rotated_bound("test") (dataset.py:354-372).

Two ways to shard, same results layout ([N,4,4] float32, row 3 all-zero):
  * reference style: one process per GPU with explicit ``-l/-r``; each writes ``{l}_{r}.npy``; ``--combine True``
    concatenates them into results.h5 (test_mult.py:70-81);
  * MI355X style: launch under ``torch.distributed.run`` -- the ranks deal the pairs of [l, r) out interleaved, solve them, and
    ONE RCCL all-gather of [n,12] assembles the result on every rank; rank 0 writes results.h5 (no files, no ``sleep 600``)."""
import argparse
import logging
import os

import numpy as np
import torch
import torch.distributed as dist

from .. import distributed as hd
from .. import io as hio
from ..config import load_config
from ..train_utils import solve
from . import _common


def solve_range(args, l, r, device, kernel, iters, take=None):
    """test_mult_modelnet.py:38-48 over the pairs [l, r) (``take``: index tensor into that range, for interleaved shards)."""
    if r <= l or (take is not None and len(take) == 0):
        return np.zeros((0, 4, 4), np.float32)
    dataloader_test = _common.loader("test", args, l, r, take=take)
    print("Test Size:{}".format(len(dataloader_test)))
    result_list = []
    for i, data in enumerate(dataloader_test):
        src, tgt, label = data
        result = solve(src.float().to(device), tgt.float().to(device), prefix='test', kernel=kernel, _iters=iters)
        result_list.append(result.detach().numpy())
        print('Solve step:{}'.format(i))
    return np.concatenate(result_list, axis=0)


def main(argv=None):
    ap = argparse.ArgumentParser(description='Train config file')
    ap.add_argument('-c', '--config', required=True)
    # the reference's own defaults, which always override the config's l / r (test_mult.py:87-88,96-97): without -l/-r a
    # process solves [0, 500) and writes 0_500.npy, exactly what --combine's defaults (step 500, num 4) look for
    ap.add_argument('-l', '--left', default=0, help='solve the left index')
    ap.add_argument('-r', '--right', default=500, help='solve the right index')
    ap.add_argument('--combine', default=False, type=str)     # a *string* in the reference: any non-empty value is truthy
    ap.add_argument('--step', type=int, default=500, help='shard size --combine expects (test_mult.py:70: 500)')
    ap.add_argument('--num', type=int, default=4, help='shards --combine expects (test_mult.py:70: 4)')
    ap.add_argument('--kernel', type=int, default=64)
    ap.add_argument('--iters', type=int, default=500, help="the reference hard-codes 500 (train_utils.py:488)")
    a = ap.parse_args(argv)
    args = load_config(a.config)
    args.l = int(a.left)
    args.r = int(a.right)
    args.combine = a.combine
    # shards and --combine must meet in ONE directory: no time stamp (the reference gets that from load_model's dirname)
    log_dir = _common.make_log_dir(args, stamp=False)
    _common.setup_logging(log_dir)
    if args.combine:
        res = hio.combine_shards(log_dir, step=a.step, num=a.num)
        print(res.shape)
        print("saved", hio.save_results(log_dir, res))
        return res
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    l, r = int(args.l), int(args.r)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
        _common.seed_everything(args)
        n = r - l
        # interleaved shards of [l, r): retry-heavy pairs cluster in pair order (houv_amd/distributed.py)
        take = hd.shard_indices(n, dist.get_rank(), world)
        mine = torch.from_numpy(solve_range(args, l, r, dev, a.kernel, a.iters, take=take)).to(dev)
        full = hd.gather_transforms(mine, n, mode=hd.DEFAULT_SHARDING)
        if dist.get_rank() == 0:
            print("saved", hio.save_results(log_dir, full.cpu().numpy()))
        dist.barrier()
        dist.destroy_process_group()
        return full
    _common.seed_everything(args)
    res = solve_range(args, l, r, dev, a.kernel, a.iters)
    print(res.shape, l, r)
    logging.info("saved %s", hio.save_shard(log_dir, l, r, res))
    return res


if __name__ == "__main__":
    main()
