"""``python -m houv_amd.drivers.test -c cfgs/houv.yaml`` -- the test-set driver north_star names
(registration/test.py:27-98): MVP_ExtraTest_RG.h5 through MVP_RG_rotated("test") -> per batch
``solve(src, tgt, prefix='test')`` (train_utils.py:467) -> ``results.h5`` (dataset 'results', float32 [N,4,4], :70-71)
-> ``submission.zip`` (:73-76).

The reference's own test.py cannot run HOUV as shipped: it instantiates ``models.houv.Model`` which does not exist
(:35-36) and its dataset evaluates ``self.transforms`` for the test split (dataset.py:334).  This mirrors the loop it
evidently intends (the intact copy is test_mult_modelnet.py:38-52); the unused DataParallel model is not built."""
import argparse
import logging

import numpy as np
import torch

from .. import io as hio
from ..config import load_config
from ..train_utils import solve
from . import _common


def test(args, log_dir, n_synthetic=16, kernel=64, iters=500):
    logging.info(str(args))
    dataloader_test = _common.loader("test", args, n_synthetic=n_synthetic)
    logging.info('Testing...')
    print("Test Size:{}".format(len(dataloader_test)))
    device = torch.device("cuda", torch.cuda.current_device())
    result_list = []
    for _, data in enumerate(dataloader_test):
        src, tgt, _ = data
        result = solve(src.float().to(device), tgt.float().to(device), prefix='test', kernel=kernel, _iters=iters)
        result_list.append(result.detach().numpy())
    all_results = np.concatenate(result_list, axis=0)
    print(all_results.shape)
    out = hio.save_results(log_dir, all_results)                    # results.h5 + submission.zip
    print("Submission file has been saved to %s/submission.zip" % (log_dir))
    return all_results, out


def main(argv=None):
    ap = argparse.ArgumentParser(description='Train config file')
    ap.add_argument('-c', '--config', help='path to config file', required=True)
    ap.add_argument('--pairs', type=int, default=16, help='number of synthetic test pairs when the h5 is absent')
    ap.add_argument('--kernel', type=int, default=64)
    ap.add_argument('--iters', type=int, default=500, help="the reference hard-codes 500 (train_utils.py:488)")
    a = ap.parse_args(argv)
    args = load_config(a.config)
    log_dir = _common.make_log_dir(args)
    _common.setup_logging(log_dir)
    _common.seed_everything(args)       # test.py does not seed (solve draws from the global numpy RNG); seeding makes runs repeatable
    return test(args, log_dir, a.pairs, a.kernel, a.iters)


if __name__ == "__main__":
    main()
