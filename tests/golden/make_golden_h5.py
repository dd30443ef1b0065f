"""Writes the HDF5 fixtures of tests/test_hdf5_min.py with the REAL libhdf5 (1.10.6, found in this image under
/opt/conda/lib; driven through ctypes -- h5py is not installed):

  g10_mvp_like.h5      libver 'earliest' (superblock v0, v1 object headers, symbol-table groups, B-tree v1 chunk index):
                       the shapes/dtypes of MVP_*_RG.h5 (registration/dataset.py:205-238) in miniature, one dataset per
                       storage flavour: contiguous, chunked, chunked+gzip, chunked+shuffle+gzip, +fletcher32, compact,
                       big-endian, a nested group, an edge-chunked 2-level B-tree, > 1024 chunks.
  g11_latest.h5        libver 'latest' (superblock v3, v2 object headers, link messages, v4 layouts).
  g10_g11_expected.npz the arrays that were written.

Run once here: ``python tests/golden/make_golden_h5.py``; the .h5 files are data (a few KB each)."""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.environ.get("HOUV_HDF5_LIB", "/opt/conda/lib/libhdf5.so")
lib = ctypes.CDLL(LIB)
hid_t = ctypes.c_int64
lib.H5open()
for fn, res, args in (
        ("H5Fcreate", hid_t, [ctypes.c_char_p, ctypes.c_uint, hid_t, hid_t]),
        ("H5Fclose", ctypes.c_int, [hid_t]),
        ("H5Pcreate", hid_t, [hid_t]),
        ("H5Pclose", ctypes.c_int, [hid_t]),
        ("H5Pset_chunk", ctypes.c_int, [hid_t, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]),
        ("H5Pset_deflate", ctypes.c_int, [hid_t, ctypes.c_uint]),
        ("H5Pset_shuffle", ctypes.c_int, [hid_t]),
        ("H5Pset_fletcher32", ctypes.c_int, [hid_t]),
        ("H5Pset_layout", ctypes.c_int, [hid_t, ctypes.c_int]),
        ("H5Pset_libver_bounds", ctypes.c_int, [hid_t, ctypes.c_int, ctypes.c_int]),
        ("H5Pset_istore_k", ctypes.c_int, [hid_t, ctypes.c_uint]),
        ("H5Screate_simple", hid_t, [ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p]),
        ("H5Sclose", ctypes.c_int, [hid_t]),
        ("H5Dcreate2", hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        ("H5Dwrite", ctypes.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, ctypes.c_void_p]),
        ("H5Dclose", ctypes.c_int, [hid_t]),
        ("H5Gcreate2", hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t]),
        ("H5Gclose", ctypes.c_int, [hid_t])):
    f = getattr(lib, fn)
    f.restype, f.argtypes = res, args


def g(name):
    return hid_t.in_dll(lib, name).value


MEM = {"float32": "H5T_NATIVE_FLOAT_g", "float64": "H5T_NATIVE_DOUBLE_g", "int64": "H5T_NATIVE_INT64_g",
       "int32": "H5T_NATIVE_INT32_g", "uint8": "H5T_NATIVE_UINT8_g", "int16": "H5T_NATIVE_INT16_g"}
H5F_ACC_TRUNC, H5P_DEFAULT, H5S_ALL = 2, 0, 0
H5D_COMPACT, H5D_CONTIGUOUS, H5D_CHUNKED = 0, 1, 2


def put(loc, name, a, chunks=None, gzip=None, shuffle=False, fletcher=False, layout=None, filetype=None):
    a = np.ascontiguousarray(a)
    dims = (ctypes.c_uint64 * a.ndim)(*a.shape)
    space = lib.H5Screate_simple(a.ndim, dims, None)
    dcpl = lib.H5Pcreate(g("H5P_CLS_DATASET_CREATE_ID_g"))
    if chunks:
        assert lib.H5Pset_chunk(dcpl, len(chunks), (ctypes.c_uint64 * len(chunks))(*chunks)) >= 0
    if shuffle:
        assert lib.H5Pset_shuffle(dcpl) >= 0
    if gzip is not None:
        assert lib.H5Pset_deflate(dcpl, gzip) >= 0
    if fletcher:
        assert lib.H5Pset_fletcher32(dcpl) >= 0
    if layout is not None:
        assert lib.H5Pset_layout(dcpl, layout) >= 0
    mem = g(MEM[a.dtype.name])
    ds = lib.H5Dcreate2(loc, name.encode(), g(filetype) if filetype else mem, space, H5P_DEFAULT, dcpl, H5P_DEFAULT)
    assert ds >= 0, name
    assert lib.H5Dwrite(ds, mem, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(ctypes.c_void_p)) >= 0, name
    lib.H5Dclose(ds)
    lib.H5Pclose(dcpl)
    lib.H5Sclose(space)


def build(path, latest, arrays):
    fapl = lib.H5Pcreate(g("H5P_CLS_FILE_ACCESS_ID_g"))
    fcpl = lib.H5Pcreate(g("H5P_CLS_FILE_CREATE_ID_g"))
    if latest:
        assert lib.H5Pset_libver_bounds(fapl, 2, 2) >= 0     # H5F_LIBVER_V110 = "latest" in 1.10
    else:
        assert lib.H5Pset_istore_k(fcpl, 2) >= 0             # tiny chunk B-tree nodes: forces a 2-level tree
    f = lib.H5Fcreate(path.encode(), H5F_ACC_TRUNC, fcpl, fapl)
    assert f >= 0
    A = arrays
    put(f, "src", A["src"], chunks=(2, 64, 3), gzip=4, shuffle=True)
    put(f, "tgt", A["tgt"], chunks=(4, 64, 3), gzip=1)
    put(f, "rotated_src", A["rotated_src"], chunks=(5, 16, 3), gzip=6, shuffle=True, fletcher=True)
    put(f, "rotated_tgt", A["rotated_tgt"], chunks=(1, 64, 3))
    put(f, "transforms", A["transforms"])
    put(f, "pose_src", A["pose_src"], layout=H5D_COMPACT)
    put(f, "cat_labels", A["cat_labels"])
    put(f, "match_level", A["match_level"], chunks=(4,))
    put(f, "match_id", A["match_id"], filetype="H5T_STD_I32BE_g")
    put(f, "rot_level", A["rot_level"])
    put(f, "edge", A["edge"], chunks=(3, 5), gzip=2)          # 7x3 chunks with ragged edges
    put(f, "many", A["many"], chunks=(1,))                    # > 1024 chunks: paged fixed-array index in 'latest'
    put(f, "manyz", A["manyz"], chunks=(1, 8), gzip=1)
    grp = lib.H5Gcreate2(f, b"extra", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
    put(grp, "complete", A["extra/complete"], chunks=(1, 32, 3), shuffle=True)
    lib.H5Gclose(grp)
    lib.H5Pclose(fapl)
    lib.H5Pclose(fcpl)
    assert lib.H5Fclose(f) >= 0


def main():
    rng = np.random.default_rng(20211)
    n = 6
    arrays = {
        "src": rng.standard_normal((n, 64, 3)).astype(np.float32),
        "tgt": rng.standard_normal((n, 64, 3)).astype(np.float32),
        "rotated_src": rng.standard_normal((n, 64, 3)).astype(np.float32),
        "rotated_tgt": rng.standard_normal((n, 64, 3)).astype(np.float32),
        "transforms": rng.standard_normal((n, 4, 4)),
        "pose_src": rng.standard_normal((n, 4, 4)).astype(np.float32),
        "cat_labels": rng.integers(0, 16, n).astype(np.int64),
        "match_level": rng.integers(0, 3, n).astype(np.int32),
        "match_id": rng.integers(-1000, 1000, (n, 2)).astype(np.int32),
        "rot_level": rng.integers(0, 2, n).astype(np.uint8),
        "edge": rng.integers(-30000, 30000, (19, 13)).astype(np.int16),
        "extra/complete": rng.standard_normal((2, 32, 3)).astype(np.float32),
        "many": rng.integers(-30000, 30000, 1100).astype(np.int16),
        "manyz": rng.integers(0, 4, (1030, 8)).astype(np.int32),
    }
    build(os.path.join(HERE, "g10_mvp_like.h5"), False, arrays)
    build(os.path.join(HERE, "g11_latest.h5"), True, arrays)
    np.savez(os.path.join(HERE, "g10_g11_expected.npz"), **{k.replace("/", "__"): v for k, v in arrays.items()})
    for p in ("g10_mvp_like.h5", "g11_latest.h5", "g10_g11_expected.npz"):
        print(p, os.path.getsize(os.path.join(HERE, p)), "bytes")


if __name__ == "__main__":
    sys.exit(main())
