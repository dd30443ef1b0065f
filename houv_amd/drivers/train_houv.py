"""``python -m houv_amd.drivers.train_houv -c cfgs/houv.yaml`` -- the north-star entry of the reference
(registration/train_HOUV.py:40-153): iterate the validation pairs in batches, ``solve_model`` each batch on the GPU,
and meter RotE / transE / MSE.  Without the MVP h5 files (not shipped, no h5py) it runs on synthetic MVP-shaped pairs."""
import argparse
import datetime
import logging
import os
import random
import sys

import numpy as np
import torch

from .. import synthetic
from ..config import load_config
from ..models.houv import HOUV, solve_model
from ..train_utils import AverageValueMeter, rmse_loss


def iterate_pairs(args, n_pairs, device):
    """Yields (src_rotated, tgt_rotated, transform) batches like the val loader of train_HOUV.py:90-104."""
    bs = int(args.batch_size)
    for first in range(0, n_pairs, bs):
        n = min(bs, n_pairs - first)
        s, t, pose = synthetic.make_pairs(n, int(args.num_points), seed=int(args.manual_seed or 2021), first_id=first)
        yield s.float().to(device), t.float().to(device), pose.float().to(device)


def train(args, n_pairs, kernel=64, num_epochs=200):
    logging.info(str(args))
    meters = {m: AverageValueMeter() for m in ('RotE', 'transE', 'MSE')}
    seed = int(args.manual_seed) if args.manual_seed else random.randint(1, 10000)      # train_HOUV.py:74-82
    logging.info('Random Seed: %d' % seed)
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    device = torch.device("cuda", torch.cuda.current_device())
    # train_HOUV.py:84 -- ctor sizes are throw-away (reset_weight re-creates the parameters per stage)
    net = HOUV(int(args.batch_size) * int(args.kernel), 0).to(device)
    for i, (src_rotated, tgt_rotated, transform) in enumerate(iterate_pairs(args, n_pairs, device)):
        r_err, t_err, ans = solve_model(net, src_rotated, tgt_rotated, transform, kernel=kernel, num_epochs=num_epochs)
        mse = rmse_loss(src_rotated, ans, transform)
        meters['RotE'].update(r_err.mean().item())
        meters['transE'].update(t_err.mean().item())
        meters['MSE'].update(mse.mean().item())
        if i % 10 == 0:
            logging.info('RotE:{} TransE:{} MSE:{}'.format(meters['RotE'].avg, meters['transE'].avg, meters['MSE'].avg))
    print(meters['RotE'].avg)
    print(meters['transE'].avg)
    print(meters['MSE'].avg)
    return {k: v.avg for k, v in meters.items()}


def main(argv=None):
    ap = argparse.ArgumentParser(description='Train config file')
    ap.add_argument('-c', '--config', help='path to config file', required=True)
    ap.add_argument('--pairs', type=int, default=100, help='number of synthetic validation pairs')
    ap.add_argument('--kernel', type=int, default=64)       # solve_model's default (houv.py:142), NOT cfg.kernel
    ap.add_argument('--iters', type=int, default=200)
    a = ap.parse_args(argv)
    args = load_config(a.config)
    time = datetime.datetime.now().isoformat()[:19]
    log_dir = os.path.join(args.work_dir, args.model_name + '_' + args.benchmark + '_' + args.flag + '_' + time)
    os.makedirs(log_dir, exist_ok=True)
    logging.basicConfig(level=logging.INFO, handlers=[logging.FileHandler(os.path.join(log_dir, 'train.log')),
                                                      logging.StreamHandler(sys.stdout)])
    return train(args, a.pairs, a.kernel, a.iters)


if __name__ == "__main__":
    main()
