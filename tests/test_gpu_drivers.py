"""GPU smoke of the driver mirrors (train_HOUV.py / test_mult.py flows) on tiny synthetic workloads."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_houv_driver(tmp_path, monkeypatch):
    from houv_amd.drivers import train_houv
    cfg = tmp_path / "houv.yaml"
    cfg.write_text(f"batch_size: 4\nworkers: 0\nmodel_name: houv\nload_model: null\nwork_dir: {tmp_path}/log/\nflag: t\n"
                   "manual_seed: 2021\nnum_points: 256\nbenchmark: mvp\nkernel: 32\nlr: 0.01\n")
    res = train_houv.main(["-c", str(cfg), "--pairs", "8", "--kernel", "26", "--iters", "60"])
    assert set(res) == {"RotE", "transE", "MSE"} and np.isfinite(list(res.values())).all()
    assert res["RotE"] < 60.0          # 8 easy-ish 256-pt pairs, 60 iterations: far better than chance (~90 deg)


def test_test_mult_driver_shards_and_combine(tmp_path):
    from houv_amd.drivers import test_mult
    cfg = tmp_path / "houv.yaml"
    cfg.write_text(f"batch_size: 2\nmodel_name: houv\nwork_dir: {tmp_path}/log/\nmanual_seed: 2021\nnum_points: 128\n")
    np.random.seed(0)
    a = test_mult.main(["-c", str(cfg), "-l", "0", "-r", "2", "--kernel", "26", "--iters", "15"])
    b = test_mult.main(["-c", str(cfg), "-l", "2", "-r", "4", "--kernel", "26", "--iters", "15"])
    assert a.shape == (2, 4, 4) and np.all(a[:, 3, :] == 0)
    full = test_mult.main(["-c", str(cfg), "-l", "0", "-r", "2", "--combine", "True"])
    np.testing.assert_array_equal(full, np.concatenate([a, b], 0))
    assert os.path.exists(os.path.join(str(tmp_path), "log", "houv", "results.npy"))
