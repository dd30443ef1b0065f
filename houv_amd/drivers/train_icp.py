"""``python -m houv_amd.drivers.train_icp -c cfgs/houv.yaml [--init houv]`` -- mirror of registration/train_ICP.py:100-200, the ICP
baseline driver: iterate the validation split (17-tuples of dataset.py:346), run point-to-point ICP (threshold 0.02,
<= 500 iterations: :136,:148-151) on every (src_rotated, tgt_rotated) pair and meter RotE / transE / MSE.

The reference calls Open3D per pair in a Python loop, always from the fixed tutorial rotation of :138-141 (translation 0), and
leaves the HOUV -> ICP chaining commented out (:119).  Here the whole batch is ONE launch of houv_icp_refine;
``--init tutorial`` (default) reproduces the reference's initialisation, ``--init identity`` starts from I, ``--init houv``
runs solve_model first and refines its answer (BASELINE configs[3]).  Open3D is not a dependency (parity unpinned: DESIGN 9.1)."""
import argparse
import logging

import numpy as np
import torch

from ..config import load_config
from ..icp import ICP_MAX_ITERATION, ICP_THRESHOLD, icp_refine
from ..models.houv import HOUV, solve_model
from ..train_utils import AverageValueMeter, rmse_loss, rotation_error, translation_error
from . import _common

# train_ICP.py:138-141 (the Open3D tutorial's initial guess with its translation zeroed)
TUTORIAL_INIT = np.asarray([[0.862, 0.011, -0.507, 0.0], [-0.139, 0.967, -0.215, 0.0], [0.487, 0.255, 0.835, 0.0],
                            [0.0, 0.0, 0.0, 1.0]], dtype=np.float32)


def train(args, init="tutorial", n_synthetic=100, kernel=64, num_epochs=200):
    logging.info(str(args))
    meters = {m: AverageValueMeter() for m in ('RotE', 'transE', 'MSE')}
    dataloader = _common.loader("val", args, n_synthetic=n_synthetic)
    _common.seed_everything(args)
    device = torch.device("cuda", torch.cuda.current_device())
    net = HOUV(int(args.batch_size) * int(args.kernel), 0).to(device) if init == "houv" else None
    for i, data in enumerate(dataloader, 0):
        src_rotated = data[2].float().to(device)
        tgt_rotated = data[3].float().to(device)
        transform = data[4].float().to(device)
        B = src_rotated.shape[0]
        if init == "houv":
            _, _, start = solve_model(net, src_rotated, tgt_rotated, transform, kernel=kernel, num_epochs=num_epochs)
        elif init == "identity":
            start = None
        else:
            start = torch.from_numpy(TUTORIAL_INIT).to(device).expand(B, 4, 4).contiguous()
        ans = icp_refine(src_rotated, tgt_rotated, start, ICP_THRESHOLD, ICP_MAX_ITERATION)        # train_ICP.py:148-151, batched
        r_err = rotation_error(ans[:, :3, :3], transform[:, :3, :3])
        t_err = translation_error(ans[:, :3, 3], transform[:, :3, 3])
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
    ap.add_argument('--init', choices=["tutorial", "identity", "houv"], default="tutorial")
    ap.add_argument('--pairs', type=int, default=100, help='number of synthetic validation pairs when the h5 is absent')
    ap.add_argument('--kernel', type=int, default=64)
    ap.add_argument('--iters', type=int, default=200)
    a = ap.parse_args(argv)
    args = load_config(a.config)
    log_dir = _common.make_log_dir(args)
    _common.setup_logging(log_dir)
    return train(args, a.init, a.pairs, a.kernel, a.iters)


if __name__ == "__main__":
    main()
