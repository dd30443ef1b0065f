"""On-disk formats at the edge of the path: per-shard ``{l}_{r}.npy`` files and the combined ``results`` array
[N,4,4] float32 (registration/test_mult_modelnet.py:51-52, test_mult.py:70-81, test.py:70-71).  ``results.h5`` and
the MVP ``*.h5`` inputs need h5py, which is optional: without it the arrays are kept as .npy and a clear error says so."""
import os

import numpy as np

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
    """Write the [N,4,4] float32 result array: ``results.h5`` (dataset 'results', test.py:70-71) when h5py is
    available, and always ``results.npy`` beside it."""
    os.makedirs(log_dir, exist_ok=True)
    arr = np.asarray(results, dtype=np.float32)
    np.save(os.path.join(log_dir, "results.npy"), arr)
    out = os.path.join(log_dir, "results.npy")
    if h5py is not None:
        out = os.path.join(log_dir, "results.h5")
        with h5py.File(out, "w") as f:
            f.create_dataset("results", data=arr)
    # test.py:73-76 shells out to `zip -r submission.zip results.h5`; same archive, without the subprocess
    import zipfile
    with zipfile.ZipFile(os.path.join(log_dir, "submission.zip"), "w", zipfile.ZIP_DEFLATED) as z:
        z.write(out, os.path.basename(out))
    return out


def load_mvp_rg(path, l=None, r=None):
    """MVP_*_RG.h5 reader (registration/dataset.py:205-238, :369-372): returns dict(src, tgt[, transforms, ...]) of the
    arrays present, optionally sliced [l:r] like MVP_RG_rotated_bound."""
    if h5py is None:
        raise RuntimeError("h5py is not installed: MVP .h5 files cannot be read here; use houv_amd.synthetic instead")
    out = {}
    with h5py.File(path, "r") as f:
        for k in ("src", "tgt", "complete", "transforms", "rotated_src", "rotated_tgt", "pose_src", "pose_tgt",
                  "rot_level", "match_level", "match_id", "cat_labels"):
            if k in f:
                a = f[k]
                out[k] = np.array(a[l:r] if (l is not None or r is not None) else a)
    return out
