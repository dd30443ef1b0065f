"""CPU tests of the real-data path: the MVP h5 layouts as the reference's datasets read them
(registration/dataset.py:189-238, :354-478) -- including ``match_id`` as a GROUP of ragged per-pair lists -- through
io.load_mvp_rg / dataset.MVP_RG_rotated(_bound), whole and sharded, with h5py absent (hdf5_min) ; and the three driver
mirrors (train_HOUV.py, test.py, test_mult*.py) driven from those files with a stubbed solver (no GPU).
The fixtures (tests/golden/mvp_mini/*.h5) were written by the real libhdf5: tests/golden/make_golden_mvp_mini.py."""
import os
import zipfile

import numpy as np
import pytest
import torch

from houv_amd import dataset, hdf5_min
from houv_amd import io as hio
from houv_amd.config import Config

MINI = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mvp_mini")
E = np.load(os.path.join(MINI, "expected.npz"))
N_PAIRS = 12


def _args(**kw):
    base = dict(data_dir=MINI, l=0, r=4, category=None, batch_size=5, workers=0, manual_seed=2021, model_name="houv",
                benchmark="mvp", flag="t", load_model=None, num_points=128, kernel=32)
    base.update(kw)
    return Config(base)


@pytest.mark.parametrize("lr", [(None, None), (0, 5), (3, 9), (10, 20), (4, 4)])
def test_val_file_whole_and_sharded_with_match_id_group(lr):
    l, r = lr
    d = hio.load_mvp_rg(os.path.join(MINI, "MVP_Test_RG.h5"), l, r, match_id=True)
    sl = slice(l, r)
    for k in hio.MVP_KEYS:
        assert np.array_equal(d[k], E["val__" + k][sl]), k
        assert d[k].dtype == E["val__" + k].dtype
    ids = list(range(N_PAIRS))[sl]
    assert len(d["match_id"]) == len(ids)
    for got, i in zip(d["match_id"], ids):                       # ragged per-pair lists, dataset.py:211-215
        assert np.array_equal(got, E["match_id__%d" % i]) and got.dtype == np.int32
    with hdf5_min.H5File(os.path.join(MINI, "MVP_Test_RG.h5")) as f:       # the access pattern of the reference itself
        assert len(f["match_id"].keys()) == N_PAIRS
        assert np.array_equal(f["match_id"][str(7)][:], E["match_id__7"])
    assert "match_id" not in hio.load_mvp_rg(os.path.join(MINI, "MVP_Test_RG.h5"), l, r)      # HOUV never reads it


def test_h5py_absent_path_is_the_one_tested():
    assert hio.h5py is None, "these tests pin the hdf5_min path; with h5py installed run them again with it masked"


def test_dataset_tuple_layouts_from_the_files():
    val = dataset.MVP_RG_rotated("val", _args())
    assert len(val) == N_PAIRS
    it = val[3]
    assert len(it) == 17                                                      # dataset.py:346
    np.testing.assert_array_equal(it[0].numpy(), E["val__complete"][3])        # :321-323 val serves `complete` as src
    np.testing.assert_array_equal(it[1].numpy(), E["val__tgt"][3])
    np.testing.assert_array_equal(it[2].numpy(), E["val__rotated_src"][3])     # what HOUV solves on (train_HOUV.py:92-112)
    np.testing.assert_array_equal(it[3].numpy(), E["val__rotated_tgt"][3])
    np.testing.assert_array_equal(it[4].numpy(), E["val__transforms"][3])
    assert it[5] == int(E["val__match_level"][3]) and it[6] == int(E["val__rot_level"][3])
    np.testing.assert_array_equal(it[7].numpy(), E["val__pose_src"][3])
    assert int(it[10]) == int(E["val__cat_labels"][3])
    ang = dataset.rotation_angle_deg(E["val__transforms"][3][:3, :3].astype(np.float64))
    assert abs(it[16] - ang) < 1e-3 and float(it[15]) == float(ang > 45)       # add_ps / angle via translation_back
    b = dataset.MVP_RG_rotated_bound("val", _args(l=2, r=7))
    assert len(b) == 5 and len(b[0]) == 8                                      # dataset.py:476
    np.testing.assert_array_equal(b[1][2].numpy(), E["val__rotated_src"][3])
    t = dataset.MVP_RG_rotated_bound("test", _args(l=8, r=20))                 # r beyond the end clips like h5py slicing
    assert len(t) == 4 and len(t[0]) == 3                                      # dataset.py:478
    np.testing.assert_array_equal(t[1][0].numpy(), E["test__rotated_src"][9])
    np.testing.assert_array_equal(t[1][1].numpy(), E["test__rotated_tgt"][9])
    lab = int(E["val__cat_labels"][0])
    c = dataset.MVP_RG_rotated("val", _args(category=lab)) if lab else None    # dataset.py:240-251 (category 0 is falsy there too)
    if c is not None:
        assert len(c) == int((E["val__cat_labels"] == lab).sum())
    with pytest.raises(RuntimeError):
        dataset.MVP_RG_rotated("test", _args(data_dir="/nonexistent"))
    ds, path = dataset.open_pairs("test", _args(data_dir="/nonexistent", num_points=32), 3, 7)
    assert path is None and len(ds) == 4 and len(ds[0]) == 3                   # synthetic fallback keeps the shard arithmetic


def _fake_solve_factory(calls):
    """Stands in for train_utils.solve / models.houv.solve_model: a deterministic function of the clouds it was given."""
    def fake_solve(src, tgt, pose=None, *a, kernel=64, prefix='train', **kw):
        calls.append((tuple(src.shape), kernel))
        ans = torch.zeros((src.shape[0], 4, 4))
        ans[:, :3, :3] = torch.eye(3)
        ans[:, :3, 3] = src.mean(dim=1).cpu() - tgt.mean(dim=1).cpu()
        return ans
    return fake_solve


@pytest.fixture
def cpu_as_cuda(monkeypatch):
    """The drivers address cuda:<local rank>; on the CPU box route that to the CPU so that their control flow can run."""
    real_device = torch.device
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(torch.cuda, "set_device", lambda *_: None)
    monkeypatch.setattr(torch, "device", lambda *a, **k: real_device("cpu") if a and a[0] == "cuda" else real_device(*a, **k))


def _cfg(tmp_path, bs=5):
    cfg = tmp_path / "houv.yaml"
    cfg.write_text(f"batch_size: {bs}\nworkers: 0\nmodel_name: houv\nload_model: null\nwork_dir: {tmp_path}/log/\nflag: t\n"
                   f"manual_seed: 2021\nnum_points: 128\nbenchmark: mvp\nkernel: 32\ndata_dir: {MINI}\n")
    return str(cfg)


def test_test_driver_reads_the_file_and_writes_results_h5(tmp_path, monkeypatch, cpu_as_cuda):
    from houv_amd.drivers import test as drv
    calls = []
    monkeypatch.setattr(drv, "solve", _fake_solve_factory(calls))
    res, out = drv.main(["-c", _cfg(tmp_path), "--kernel", "26"])
    assert [c[0] for c in calls] == [(5, 128, 3), (5, 128, 3), (2, 128, 3)] and calls[0][1] == 26     # 12 pairs in batches of 5
    want = (E["test__rotated_src"].mean(1) - E["test__rotated_tgt"].mean(1))
    np.testing.assert_allclose(res[:, :3, 3], want, atol=1e-6)
    assert out.endswith("results.h5")
    back = hio.load_results(out)
    assert back.dtype == np.float32 and np.array_equal(back, res)              # test.py:70-71
    with zipfile.ZipFile(os.path.join(os.path.dirname(out), "submission.zip")) as z:
        assert z.namelist() == ["results.h5"]                                  # test.py:73-76


def test_test_mult_shards_from_the_file_and_combines(tmp_path, monkeypatch, cpu_as_cuda):
    from houv_amd.drivers import test_mult as drv
    calls = []
    monkeypatch.setattr(drv, "solve", _fake_solve_factory(calls))
    cfg = _cfg(tmp_path, bs=4)
    parts = [drv.main(["-c", cfg, "-l", str(l), "-r", str(l + 6)]) for l in (0, 6)]
    assert [c[0][0] for c in calls] == [4, 2, 4, 2]
    log_dir = os.path.join(str(tmp_path), "log", "houv_mvp_t")
    assert sorted(f for f in os.listdir(log_dir) if f.endswith(".npy")) == ["0_6.npy", "6_12.npy"]      # test_mult_modelnet.py:51-52
    full = drv.main(["-c", cfg, "--combine", "True", "--step", "6", "--num", "2"])
    assert np.array_equal(full, np.concatenate(parts, 0)) and full.shape == (12, 4, 4)
    want = (E["test__rotated_src"].mean(1) - E["test__rotated_tgt"].mean(1))
    np.testing.assert_allclose(full[:, :3, 3], want, atol=1e-6)                # every pair once, in file order
    assert np.array_equal(hio.load_results(os.path.join(log_dir, "results.h5")), full)


def test_train_houv_driver_reads_the_val_file(tmp_path, monkeypatch, cpu_as_cuda):
    from houv_amd.drivers import train_houv as drv
    seen = []

    class FakeNet(torch.nn.Module):
        def __init__(self, *a):
            super().__init__()

    def fake_solve_model(net, src, tgt, pose, kernel=64, num_epochs=200, **kw):
        seen.append((src.clone(), tgt.clone(), pose.clone(), kernel, num_epochs))
        ans = pose.clone()
        ans[:, 3, :] = 0
        return torch.zeros(src.shape[0]), torch.zeros(src.shape[0]), ans
    monkeypatch.setattr(drv, "HOUV", FakeNet)
    monkeypatch.setattr(drv, "solve_model", fake_solve_model)
    res = drv.main(["-c", _cfg(tmp_path), "--kernel", "26", "--iters", "7"])
    assert [s[0].shape[0] for s in seen] == [5, 5, 2] and seen[0][3:] == (26, 7)
    np.testing.assert_array_equal(torch.cat([s[0] for s in seen]).numpy(), E["val__rotated_src"])    # slots 2/3/4 of the 17-tuple
    np.testing.assert_array_equal(torch.cat([s[1] for s in seen]).numpy(), E["val__rotated_tgt"])
    np.testing.assert_array_equal(torch.cat([s[2] for s in seen]).numpy(), E["val__transforms"])
    assert res["RotE"] == 0 and res["MSE"] < 1e-6            # the ground-truth transform has zero rmse_loss against itself


def test_train_icp_driver_reads_the_val_file(tmp_path, monkeypatch, cpu_as_cuda):
    from houv_amd.drivers import train_icp as drv
    seen = []

    def fake_icp(src, tgt, init, thr, max_it):
        seen.append((src.clone(), None if init is None else init.clone(), thr, max_it))
        out = torch.eye(4).expand(src.shape[0], 4, 4).clone()
        return out
    monkeypatch.setattr(drv, "icp_refine", fake_icp)
    res = drv.main(["-c", _cfg(tmp_path), "--init", "tutorial"])
    assert [s[0].shape[0] for s in seen] == [5, 5, 2] and seen[0][2:] == (0.02, 500)                 # train_ICP.py:136,151
    np.testing.assert_allclose(seen[0][1][0].numpy(), drv.TUTORIAL_INIT)                              # :138-141
    np.testing.assert_array_equal(torch.cat([s[0] for s in seen]).numpy(), E["val__rotated_src"])
    assert set(res) == {"RotE", "transE", "MSE"}
    seen.clear()
    drv.main(["-c", _cfg(tmp_path), "--init", "identity"])
    assert seen[0][1] is None
