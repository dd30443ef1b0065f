"""Shared plumbing of the driver mirrors: the reference's log-dir rule, seeding and data loaders."""
import datetime
import logging
import os
import random
import sys

import numpy as np
import torch

from .. import io as hio
from ..dataset import open_pairs


def make_log_dir(args, stamp=True):
    """registration/test.py:87-96 / train_HOUV.py:142-150: dirname(load_model) if a checkpoint is named, else
    work_dir/{model}_{benchmark}_{flag}_{time}.  ``stamp=False`` (or cfg ``log_dir``) gives a fixed directory, which the
    shard -> combine hand-over needs (the reference's run_test.sh relies on load_model for that)."""
    if getattr(args, "log_dir", None):
        log_dir = args.log_dir
    elif getattr(args, "load_model", None):
        log_dir = os.path.dirname(args.load_model)
    else:
        name = '%s_%s_%s' % (args.model_name, args.benchmark, args.flag)
        if stamp:
            name += '_' + datetime.datetime.now().isoformat()[:19]
        log_dir = os.path.join(args.work_dir, name)
    os.makedirs(log_dir, exist_ok=True)
    return log_dir


def setup_logging(log_dir):
    logging.basicConfig(level=logging.INFO, force=True,
                        handlers=[logging.FileHandler(os.path.join(log_dir, 'train.log')), logging.StreamHandler(sys.stdout)])


def seed_everything(args):
    """train_HOUV.py:74-82 / test_mult_modelnet.py:84-92."""
    seed = int(args.manual_seed) if getattr(args, "manual_seed", None) else random.randint(1, 10000)
    logging.info('Random Seed: %d' % seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    return seed


def loader(prefix, args, l=None, r=None, n_synthetic=None, take=None):
    """DataLoader(batch_size=args.batch_size, shuffle=False, num_workers=args.workers) over the MVP split when its h5
    file exists (./data or cfg data_dir), over MVP-shaped synthetic pairs otherwise."""
    ds, path = open_pairs(prefix, args, l, r, n_synthetic)
    if path:
        logging.info('%s split from %s', prefix, path)
    else:
        logging.info('%s not found: running on %d MVP-shaped SYNTHETIC pairs', hio.mvp_path(prefix, args), len(ds))
    if take is not None:
        ds = torch.utils.data.Subset(ds, [int(i) for i in take])
    dl = torch.utils.data.DataLoader(ds, batch_size=int(args.batch_size), shuffle=False, num_workers=int(args.workers))
    logging.info('Length of %s dataset:%d', prefix, len(ds))
    return dl
