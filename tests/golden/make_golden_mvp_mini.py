"""Writes miniature MVP registration files with the REAL libhdf5 (ctypes, as make_golden_h5.py does), in the layout the
reference's datasets read (registration/dataset.py:189-238, :354-402):

  tests/golden/mvp_mini/MVP_Test_RG.h5       "val":  src tgt complete transforms rotated_src rotated_tgt pose_src pose_tgt
                                             rot_level match_level cat_labels + GROUP match_id/{"0".."n-1"} of RAGGED int32
                                             lists (dataset.py:211-215 reads f['match_id'][str(i)][:])
  tests/golden/mvp_mini/MVP_ExtraTest_RG.h5  "test": rotated_src rotated_tgt cat_labels
  tests/golden/mvp_mini/expected.npz         the arrays that were written

12 pairs x 128 points from houv_amd.synthetic (seed 4321): real partial-overlap geometry, so the drivers' RotE/transE mean
something.  Run once here: ``python tests/golden/make_golden_mvp_mini.py`` (needs /opt/conda/lib/libhdf5.so)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden_h5 as h5w  # noqa: E402  (libhdf5 through ctypes)


def write(path, arrays, groups=None):
    lib = h5w.lib
    fapl = lib.H5Pcreate(h5w.g("H5P_CLS_FILE_ACCESS_ID_g"))
    fcpl = lib.H5Pcreate(h5w.g("H5P_CLS_FILE_CREATE_ID_g"))
    f = lib.H5Fcreate(path.encode(), h5w.H5F_ACC_TRUNC, fcpl, fapl)
    assert f >= 0
    for k, a in arrays.items():
        # h5py's defaults for create_dataset(data=...) are contiguous; the big clouds get gzip chunks like MVP's files
        if a.ndim == 3 and a.shape[1] >= 64:
            h5w.put(f, k, a, chunks=(1, a.shape[1], 3), gzip=4)
        else:
            h5w.put(f, k, a)
    for gname, children in (groups or {}).items():
        grp = lib.H5Gcreate2(f, gname.encode(), h5w.H5P_DEFAULT, h5w.H5P_DEFAULT, h5w.H5P_DEFAULT)
        for cname, a in children.items():
            h5w.put(grp, cname, a)
        lib.H5Gclose(grp)
    lib.H5Pclose(fapl)
    lib.H5Pclose(fcpl)
    assert lib.H5Fclose(f) >= 0


def main():
    from houv_amd import synthetic
    n, N = 12, 128
    rng = np.random.default_rng(4321)
    out = os.path.join(HERE, "mvp_mini")
    os.makedirs(out, exist_ok=True)
    rs, rt, T = synthetic.make_pairs(n, N, seed=4321)
    rs, rt, T = rs.numpy(), rt.numpy(), T.numpy()
    # un-rotated clouds + the poses that produced the rotated ones (dataset.py:297-301: rotated = cloud @ pose^T + t)
    pose_src = np.stack([np.eye(4, dtype=np.float32)] * n)
    for i in range(n):
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] *= -1
        pose_src[i, :3, :3] = q
        pose_src[i, :3, 3] = rng.standard_normal(3) * 0.1
    pose_tgt = (T @ pose_src).astype(np.float32)
    src = np.einsum("bij,bnj->bni", np.transpose(pose_src[:, :3, :3], (0, 2, 1)), rs - pose_src[:, None, :3, 3]).astype(np.float32)
    tgt = np.einsum("bij,bnj->bni", np.transpose(pose_tgt[:, :3, :3], (0, 2, 1)), rt - pose_tgt[:, None, :3, 3]).astype(np.float32)
    val = {
        "src": src, "tgt": tgt, "complete": np.concatenate([src, tgt], 1)[:, ::2].copy(),
        "transforms": T.astype(np.float32), "rotated_src": rs, "rotated_tgt": rt, "pose_src": pose_src, "pose_tgt": pose_tgt,
        "rot_level": (np.arange(n) % 2).astype(np.int32), "match_level": (np.arange(n) % 3).astype(np.int32),
        "cat_labels": rng.integers(0, 16, n).astype(np.int64),
    }
    match_id = {str(i): rng.integers(0, N, size=int(rng.integers(3, 40))).astype(np.int32) for i in range(n)}
    write(os.path.join(out, "MVP_Test_RG.h5"), val, {"match_id": match_id})
    ts, tt, _ = synthetic.make_pairs(n, N, seed=8765)
    test = {"rotated_src": ts.numpy(), "rotated_tgt": tt.numpy(), "cat_labels": rng.integers(0, 16, n).astype(np.int64)}
    write(os.path.join(out, "MVP_ExtraTest_RG.h5"), test)
    np.savez_compressed(os.path.join(out, "expected.npz"), **{"val__" + k: v for k, v in val.items()},
                        **{"test__" + k: v for k, v in test.items()}, **{"match_id__" + k: v for k, v in match_id.items()})
    for p in os.listdir(out):
        print(p, os.path.getsize(os.path.join(out, p)), "bytes")


if __name__ == "__main__":
    main()
