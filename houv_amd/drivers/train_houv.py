"""``python -m houv_amd.drivers.train_houv -c cfgs/houv.yaml`` -- the north-star entry of the reference
(registration/train_HOUV.py:40-153): iterate the validation split (MVP_Test_RG.h5 through MVP_RG_rotated("val"), as
:41-69 does) in batches, ``solve_model`` each batch on the GPU, and meter RotE / transE / MSE.  When the h5 file is not
there (it is not shipped) the same loop runs on MVP-shaped synthetic pairs and says so in the log."""
import argparse
import logging

import torch

from ..config import load_config
from ..models.houv import HOUV, solve_model
from ..train_utils import AverageValueMeter, rmse_loss
from . import _common


def train(args, n_synthetic=100, kernel=64, num_epochs=200):
    logging.info(str(args))
    meters = {m: AverageValueMeter() for m in ('RotE', 'transE', 'MSE')}
    dataloader = _common.loader("val", args, n_synthetic=n_synthetic)             # train_HOUV.py:66-69 (the train loader is overwritten)
    _common.seed_everything(args)
    device = torch.device("cuda", torch.cuda.current_device())
    # train_HOUV.py:84 -- ctor sizes are throw-away (reset_weight re-creates the parameters per stage)
    net = HOUV(int(args.batch_size) * int(args.kernel), 0).to(device)
    for i, data in enumerate(dataloader, 0):
        # 17-tuple of dataset.py:346; HOUV uses slots 2, 3, 4 (train_HOUV.py:92-112)
        src_rotated = data[2].float().to(device)
        tgt_rotated = data[3].float().to(device)
        transform = data[4].float().to(device)
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
    ap.add_argument('--pairs', type=int, default=100, help='number of synthetic validation pairs when the h5 is absent')
    ap.add_argument('--kernel', type=int, default=64)       # solve_model's default (houv.py:142), NOT cfg.kernel
    ap.add_argument('--iters', type=int, default=200)
    a = ap.parse_args(argv)
    args = load_config(a.config)
    log_dir = _common.make_log_dir(args)
    _common.setup_logging(log_dir)
    return train(args, a.pairs, a.kernel, a.iters)


if __name__ == "__main__":
    main()
