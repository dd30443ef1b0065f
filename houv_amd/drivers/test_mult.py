"""``python -m houv_amd.drivers.test_mult -c cfg.yaml [-l L -r R] [--combine True]`` -- the sharded test driver
(registration/test_mult.py:83-125, test_mult_modelnet.py:27-54, run_test.sh).

Two ways to shard, same results layout ([N,4,4] float32, row 3 all-zero):
  * reference style: one process per GPU with explicit ``-l/-r``; each writes ``{l}_{r}.npy``; ``--combine True``
    concatenates them into results.(h5|npy);
  * MI355X style: launch under ``torch.distributed.run`` -- ranks take contiguous shards and ONE RCCL all-gather of
    [n,12] assembles the result on every rank; rank 0 writes it (no files, no ``sleep 600``)."""
import argparse
import logging
import os
import sys

import torch
import torch.distributed as dist

from .. import distributed as hd
from .. import io as hio
from .. import synthetic
from ..config import load_config
from ..train_utils import solve


def main(argv=None):
    ap = argparse.ArgumentParser(description='Train config file')
    ap.add_argument('-c', '--config', required=True)
    ap.add_argument('-l', default=0, type=int)
    ap.add_argument('-r', default=4, type=int)
    ap.add_argument('--combine', default=False, type=str)     # a *string* in the reference: any non-empty value is truthy
    ap.add_argument('--pairs', type=int, default=None, help='total synthetic pairs (distributed mode)')
    ap.add_argument('--kernel', type=int, default=64)
    ap.add_argument('--iters', type=int, default=500, help="the reference hard-codes 500 (train_utils.py:488)")
    a = ap.parse_args(argv)
    args = load_config(a.config)
    args.l, args.r, args.combine = a.l, a.r, a.combine
    log_dir = os.path.join(args.work_dir, args.model_name)
    os.makedirs(log_dir, exist_ok=True)
    logging.basicConfig(level=logging.INFO, handlers=[logging.StreamHandler(sys.stdout)])
    if args.combine:
        step = a.r - a.l if a.r > a.l else 500
        num = max(1, len([f for f in os.listdir(log_dir) if f.endswith('.npy') and '_' in f and f != 'results.npy']))
        res = hio.combine_shards(log_dir, step=step, num=num)
        print(res.shape)
        print("saved", hio.save_results(log_dir, res))
        return res
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    bs = int(args.batch_size)

    def solve_range(l, r):
        out = []
        for first in range(l, r, bs):
            n = min(bs, r - first)
            s, t, _ = synthetic.make_pairs(n, int(args.num_points), seed=int(args.manual_seed or 2021), first_id=first)
            out.append(solve(s.float().to(dev), t.float().to(dev), prefix='test', kernel=a.kernel, _iters=a.iters))
            print('Solve step:{}'.format(first // bs))
        return torch.cat(out, 0) if out else torch.zeros((0, 4, 4))

    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
        n = a.pairs or (a.r - a.l)
        l, r = hd.shard_range(n, dist.get_rank(), world)
        full = hd.gather_transforms(solve_range(l, r).to(dev), n)
        if dist.get_rank() == 0:
            print("saved", hio.save_results(log_dir, full.cpu().numpy()))
        dist.barrier()
        dist.destroy_process_group()
        return full
    res = solve_range(a.l, a.r).numpy()
    print(res.shape, a.l, a.r)
    print("saved", hio.save_shard(log_dir, a.l, a.r, res))
    return res


if __name__ == "__main__":
    main()
