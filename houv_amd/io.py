"""On-disk formats at the edge of the path: per-shard ``{l}_{r}.npy`` files and the combined ``results`` array
[N,4,4] float32 (registration/test_mult_modelnet.py:51-52, test_mult.py:70-81, test.py:70-71).  ``results.h5`` and
the MVP ``*.h5`` inputs go through h5py when it is installed and through the dependency-free ``houv_amd.hdf5_min``
otherwise (same files either way: checked against libhdf5 in tests/test_hdf5_min.py)."""
import os

import numpy as np

from . import hdf5_min

try:  # optional
    import h5py
except Exception:  # pragma: no cover - not installed in the build image
    h5py = None


def save_shard(log_dir, l, r, results):
    os.makedirs(log_dir, exist_ok=True)
    path = os.path.join(log_dir, "{}_{}.npy".format(l, r))
    np.save(path, np.asarray(results, dtype=np.float32))
    return path


def combine_shards(log_dir, step=500, num=4):
    """test_mult.py:70-81: concatenate ``{step*i}_{step*i+step}.npy`` for i < num."""
    parts = [np.load(os.path.join(log_dir, "{}_{}.npy".format(step * i, step * i + step))) for i in range(num)]
    return np.concatenate(parts, axis=0)


def save_results(log_dir, results):
    """Write the [N,4,4] float32 result array as ``results.h5`` (dataset 'results', test.py:70-71), ``results.npy``
    beside it, and ``submission.zip`` holding the .h5 (test.py:73-76)."""
    os.makedirs(log_dir, exist_ok=True)
    arr = np.asarray(results, dtype=np.float32)
    np.save(os.path.join(log_dir, "results.npy"), arr)
    out = os.path.join(log_dir, "results.h5")
    if h5py is not None:
        with h5py.File(out, "w") as f:
            f.create_dataset("results", data=arr)
    else:
        hdf5_min.write_h5(out, {"results": arr})
    # test.py:73-76 shells out to `zip -r submission.zip results.h5`; same archive, without the subprocess
    import zipfile
    with zipfile.ZipFile(os.path.join(log_dir, "submission.zip"), "w", zipfile.ZIP_DEFLATED) as z:
        z.write(out, os.path.basename(out))
    return out


def load_results(path):
    """The [N,4,4] array back from ``results.h5`` / ``results.npy``."""
    if path.endswith(".npy"):
        return np.load(path)
    if h5py is not None:
        with h5py.File(path, "r") as f:
            return np.array(f["results"])
    with hdf5_min.H5File(path) as f:
        return np.array(f["results"])


MVP_KEYS = ("src", "tgt", "complete", "transforms", "rotated_src", "rotated_tgt", "pose_src", "pose_tgt", "rot_level",
            "match_level", "match_id", "cat_labels")


def load_mvp_rg(path, l=None, r=None):
    """MVP_*_RG.h5 reader (registration/dataset.py:205-238, :369-372): returns dict(src, tgt[, transforms, ...]) of the
    arrays present, optionally sliced [l:r] like MVP_RG_rotated_bound."""
    if not os.path.exists(path):
        raise RuntimeError("%s not found: the MVP registration files are not shipped; use houv_amd.synthetic / "
                           "dataset.SyntheticRG for MVP-shaped pairs" % path)
    out = {}
    opener = (lambda p: h5py.File(p, "r")) if h5py is not None else hdf5_min.H5File
    with opener(path) as f:
        for k in MVP_KEYS:
            if k in f:
                a = f[k]
                out[k] = np.array(a[l:r] if (l is not None or r is not None) else a)
    return out
