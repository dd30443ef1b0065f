"""CPU tests of the host-side mirrors that need no GPU: config surface, import aliasing, shard files, parameter
initialisation quirks, synthetic data generator."""
import os
import sys

import numpy as np
import pytest
import torch


def test_config_accepts_reference_yaml_surface(tmp_path):
    from houv_amd.config import load_config
    p = tmp_path / "houv.yaml"
    # the key set of registration/cfgs/houv.yaml (values abridged)
    p.write_text("batch_size: 100\nworkers: 0\nnepoch: 100\nmodel_name: houv\nload_model: null\nwork_dir: log/\n"
                 "flag: debug\nloss: cd\nmanual_seed: 2021\nnum_points: 2048\nmax_angle: 180\nmax_trans: 0.5\n"
                 "benchmark: mvp\nkernel: 32\nlr: 0.01\nbetas: 0.9, 0.999\neval_emd: False\nl: 0\nr: 4\ncombine: False\n")
    a = load_config(str(p))
    assert a.batch_size == 100 and a["kernel"] == 32 and a.load_model is None and a.combine is False
    assert a.betas == "0.9, 0.999" and "manual_seed" in str(a)
    a.l = 7
    assert a["l"] == 7
    with pytest.raises(AttributeError):
        a.nope


def test_compat_aliases_resolve_reference_imports():
    from houv_amd import compat
    saved = {k: sys.modules.get(k) for k in ("metrics", "models", "models.houv", "models.dcp", "mm3d_pn2", "train_utils",
                                             "model_utils", "model_utils_completion")}
    for k in saved:
        sys.modules.pop(k, None)
    try:
        compat.install()
        from metrics import cd                                                    # noqa: F401
        from models.houv import HOUV, Predict_loss, predict_model, solve_model    # noqa: F401
        from train_utils import (AverageValueMeter, rmse_loss, rotation_error, solve,   # noqa: F401
                                 translation_error)
        from model_utils import SVDHead                                            # noqa: F401
        from models.dcp import Model                                               # noqa: F401
        from mm3d_pn2 import furthest_point_sample, gather_points                  # noqa: F401
        import houv_amd.models.houv as mine
        assert predict_model is mine.predict_model
    finally:
        compat.uninstall()
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v


def test_signatures_match_reference():
    """Argument names and defaults of the drop-in entry points (houv.py:14,40,94,106,142,209; train_utils.py:359,467)."""
    import inspect
    from houv_amd.models import houv
    from houv_amd import train_utils, model_utils
    sig = lambda f: str(inspect.signature(f))
    assert sig(houv.predict_model) == ("(net, src, src_rotated, pose=None, src_ori=None, tgt_ori=None, angle_t=None, "
                                       "label=None, kernel=64, num_epochs=500, angle_base=0, device='cuda', seed=2021)")
    assert sig(houv.solve_model) == ("(net, src, src_rotated, pose=None, src_ori=None, tgt_ori=None, angle_t=None, "
                                     "label=None, kernel=64, num_epochs=200, prefix='train')")
    assert sig(houv.Predict_loss) == "(src, src_rotated, alpha=0.5)"
    assert sig(houv.HOUV.reset_weight) == "(self, batch_size, angle_base, seed=2021)"
    assert sig(houv.HOUV.__init__) == "(self, batch_size, angle_base)"
    assert sig(train_utils.solve).startswith("(src, src_rotated, pose=None, src_ori=None, tgt_ori=None, angle_t=None, "
                                             "label=None, kernel=64, num_epochs=500, prefix='train'")
    assert sig(train_utils.getPredict_angle) == ("(src, src_rotated, pose=None, src_ori=None, tgt_ori=None, "
                                                 "angle_t=None, label=None, kernel=64, num_epochs=1000, angle_base=0)")
    assert sig(model_utils.SVDHead.forward) == "(self, src, src_corr, weights=None)"


def test_param_init_matches_golden_and_quirks(golden):
    from houv_amd import solver
    g = golden("g3_g4_params_forward.npz")
    p = solver.houv_init_params(32, 2021).astype(np.float32)
    want = np.concatenate([g["V"], g["angle"], g["tran_c"], g["tran_s"]], 1)
    np.testing.assert_array_equal(p, want)
    with pytest.raises(IndexError):
        solver.houv_init_params(25)
    # the `solve` twin draws from the global RNG in the reference's order (and burns the unused angle_XYZ draw)
    np.random.seed(3)
    a = solver.solve_twin_init_params(5)
    np.random.seed(3)
    V = np.random.randn(5, 3); an = np.random.randn(5, 1); c = np.random.randn(5, 3); s = np.random.randn(5, 1)
    np.random.randn(5, 3)
    nxt = np.random.randn()
    np.testing.assert_array_equal(a, np.concatenate([V, an, c, s], 1))
    np.random.seed(3); solver.solve_twin_init_params(5)
    assert np.random.randn() == nxt


def test_houv_module_parameters_mirror_reference():
    from houv_amd.models.houv import HOUV
    net = HOUV(40, 0)
    names = [n for n, _ in net.named_parameters()]
    assert names == ["V_c", "angle_c", "tran_c", "tran_s_cpu"]          # quirk A.5(5): tran_s only after reset_weight
    net.reset_weight(40, 1)
    assert "tran_s" in dict(net.named_parameters()) and net.angle_base == 1
    assert tuple(net.packed_params().shape) == (40, 8)
    small = HOUV(5, 0)                                                  # __init__ (unlike reset_weight) bounds the lattice fill
    assert tuple(small.V_c.shape) == (5, 3)


def test_shard_files_and_combine(tmp_path):
    from houv_amd import io as hio
    rng = np.random.default_rng(0)
    parts = [rng.standard_normal((5, 4, 4)).astype(np.float32) for _ in range(4)]
    for i, p in enumerate(parts):
        hio.save_shard(str(tmp_path), 5 * i, 5 * i + 5, p)
    res = hio.combine_shards(str(tmp_path), step=5, num=4)
    np.testing.assert_array_equal(res, np.concatenate(parts, 0))
    out = hio.save_results(str(tmp_path), res)
    assert os.path.exists(out)
    import zipfile
    with zipfile.ZipFile(os.path.join(str(tmp_path), "submission.zip")) as z:          # test.py:73-76
        assert z.namelist() == [os.path.basename(out)]
    np.testing.assert_array_equal(np.load(os.path.join(str(tmp_path), "results.npy")), res)


def test_synthetic_pairs_are_mvp_shaped():
    from houv_amd import synthetic
    s, t, T = synthetic.make_pairs(5, 256, seed=1)
    assert s.shape == (5, 256, 3) and t.shape == (5, 256, 3) and T.shape == (5, 4, 4) and s.dtype == torch.float32
    R = T[:, :3, :3]
    np.testing.assert_allclose((R @ R.transpose(1, 2)).numpy(), np.broadcast_to(np.eye(3), (5, 3, 3)), atol=1e-5)
    assert float(T[:, :3, 3].norm(dim=1).max()) <= 0.25 + 1e-6          # random_translation(0.25)
    s2, _, _ = synthetic.make_pairs(5, 256, seed=1)
    assert torch.equal(s, s2)                                            # deterministic per (seed, pair id)


def test_dataset_layouts_and_pose_samplers():
    from houv_amd import dataset
    from houv_amd.config import Config
    args = Config(num_points=64, manual_seed=3)
    val = dataset.SyntheticRG("val", args, n_pairs=3)
    item = val[1]
    assert len(item) == 17 and item[2].shape == (64, 3) and item[4].shape == (4, 4)        # dataset.py:346
    test = dataset.SyntheticRG("test", args, n_pairs=3)
    assert len(test[0]) == 3 and len(test) == 3                                            # dataset.py:348
    np.random.seed(0)
    P, ang = dataset.random_pose(np.pi / 4, 0.25)
    assert P.shape == (4, 4) and 0 <= ang <= np.pi / 4 and np.linalg.norm(P[:3, 3]) <= 0.25 + 1e-12
    np.testing.assert_allclose(P[:3, :3] @ P[:3, :3].T, np.eye(3), atol=1e-12)
    assert abs(dataset.rotation_angle_deg(P[:3, :3]) - np.degrees(ang)) < 1e-6
    with pytest.raises(RuntimeError):
        dataset.MVP_RG_rotated("test", args)                 # no h5py / no MVP files here: a clear error, not a crash


def test_every_solve_kernel_variant_is_oracle_compared():
    """The variant table of the fused loop (houv_solve_variant, houv_amd/csrc/solve.hip) enumerated over every cloud
    size: each solve_kernel<BLOCK, Q, NMET> instantiation must have a size in tests/solve_cases.py at which the GPU
    tests compare it with the CPU oracle (ORACLE_CASES), and each pruned instantiation a size at which it is compared
    bit for bit with its brute-force twin (PRUNED_CASES).  Host-only: no GPU work is launched."""
    from houv_amd import _lib
    from solve_cases import ORACLE_CASES, PRUNED_CASES
    table = {_lib.solve_variant(n, n) for n in range(1, 4097)}
    assert table == {(64, 1), (128, 1), (256, 1), (256, 2), (256, 3), (256, 4), (512, 3), (512, 4), (1024, 3), (1024, 4)}
    assert _lib.solve_variant(2048, 2048) == (512, 4)            # the kernels bench.py times (BASELINE configs[1]) ...
    assert _lib.solve_variant(2048, 2048, pruned=True, with_mode=True) == (512, 4, 2)     # ... the pruned one with the balanced walk
    assert _lib.solve_variant(2048, 2048, with_mode=True) == (512, 4, 0) and _lib.solve_variant(400, 400, True, True) == (256, 2, 2) and _lib.solve_variant(200, 200, True, True) == (256, 1, 0)
    from houv_amd import solver                                   # the host switch mirrors the library's prune-mode table
    assert all(solver.uses_pruned(n, n, True) == (_lib.solve_variant(n, n, True, True)[2] != 0) for n in range(1, 4097))
    assert solver.uses_pruned(3000, 100, True) and not solver.uses_pruned(5000, 100, True)
    # brute force and pruned search use the SAME (block, points per lane) at every size: same summation order, same bits
    assert all(_lib.solve_variant(n, n) == _lib.solve_variant(n, n, pruned=True) for n in range(1, 4097))
    assert _lib.solve_variant(100, 3000) == (1024, 3)            # the larger cloud decides
    covered = {(_lib.solve_variant(N, M), 4 if mode == "houv" else 1) for N, M, _, mode in ORACLE_CASES}
    missing = {(v, nmet) for v in table for nmet in (4, 1)} - covered
    assert not missing, f"solve_kernel variants never compared with the oracle: {sorted(missing)}"
    # pruned instantiations = the sizes where the library reports a prune mode other than 0
    ptable = {_lib.solve_variant(n, n, pruned=True) for n in range(1, 4097) if _lib.solve_variant(n, n, True, True)[2] != 0}
    assert ptable == {(256, 2), (256, 3), (256, 4), (512, 3), (512, 4), (1024, 3), (1024, 4)}
    assert _lib.solve_variant(4096, 4096, True, True) == (1024, 4, 3) and _lib.solve_variant(2049, 100, True, True) == (1024, 3, 3)
    pcovered = {(_lib.solve_variant(N, M, pruned=True), 4 if views else 1) for N, M, views, _, _ in PRUNED_CASES}
    pmissing = {(v, nmet) for v in ptable for nmet in (4, 1)} - pcovered
    assert not pmissing, f"pruned solve_kernel variants never compared with brute force: {sorted(pmissing)}"
    for bad in ((4097, 10, False), (10, 4097, True), (0, 5, False)):
        with pytest.raises(_lib.HouvHipError):
            _lib.solve_variant(*bad)


def test_kd_sort_is_canonical_and_compact():
    """solver.kd_sort (the point order of the pruned search, round 3): a permutation of the points; a function of the point
    SET -- sorting twice, or sorting any permutation of the cloud, gives the same order (bench.py and the bit-identity tests
    rely on it); every run of 32 points is a k-d leaf whose box is tighter than a Morton run's; ragged sizes split at multiples
    of 32; exact duplicates are harmless."""
    import torch
    from houv_amd import solver, synthetic
    src, _, _ = synthetic.make_pairs(3, 2048, seed=9)
    a = solver.kd_sort(src.clone())
    assert torch.equal(a.sort(dim=1)[0], src.sort(dim=1)[0])                          # same multiset per coordinate ...
    key = lambda x: x[:, :, 0].double() * 7 + x[:, :, 1].double() * 13 + x[:, :, 2].double() * 29
    assert torch.equal(key(a).sort(dim=1)[0], key(src).sort(dim=1)[0])                # ... and of whole points
    assert torch.equal(solver.kd_sort(a.clone()), a)                                  # idempotent
    perm = torch.randperm(2048, generator=torch.Generator().manual_seed(3))
    assert torch.equal(solver.kd_sort(src[:, perm].clone()), a)                       # permutation invariant

    def extent_sum(x):
        t = x.reshape(x.shape[0], -1, 32, 3)
        return float((t.max(2)[0] - t.min(2)[0]).sum(-1).mean())
    assert extent_sum(a) < 0.8 * extent_sum(solver.morton_sort(src))                  # measured 0.283 vs 0.415
    rag = solver.kd_sort(src[:, :1800].clone())                                       # 57 sub-tiles, the last one ragged
    assert rag.shape == (3, 1800, 3) and torch.equal(solver.kd_sort(rag.clone()), rag)
    tiny = solver.kd_sort(src[:, :20].clone())
    assert torch.equal(tiny.sort(dim=1)[0], src[:, :20].sort(dim=1)[0])
    dup = src[:, :64].clone()
    dup[:, 10] = dup[:, 40]                                                            # an exact duplicate
    d1 = solver.kd_sort(dup.clone())
    assert torch.equal(solver.kd_sort(dup[:, torch.randperm(64, generator=torch.Generator().manual_seed(4))].clone()), d1)
    assert solver.SPATIAL_SORT == "kd" and torch.equal(solver.spatial_sort(src.clone()), a)


def test_kd_sort_batched_levels_equal_the_per_segment_recursion():
    """solver.kd_sort splits all same-shaped segments of a tree level in one batched call; the result must be the permutation the
    plain per-segment recursion gives (restated here), for both split rules, power-of-two and ragged sizes, leaf 32 and 64."""
    import torch
    from houv_amd import solver, synthetic

    def plain(cloud, leaf, rule):
        for ax in (2, 1, 0):
            cloud = torch.gather(cloud, 1, torch.argsort(cloud[..., ax], dim=1, stable=True).unsqueeze(2).expand(-1, -1, 3))

        def rec(a, b):
            tiles = -(-(b - a) // leaf)
            if tiles <= 1:
                return
            mid = a + (tiles - tiles // 2) * leaf
            cloud[:, a:b] = solver._split_segments(cloud[:, a:b].clone(), mid - a, rule)
            rec(a, mid)
            rec(mid, b)
        rec(0, cloud.shape[1])
        return cloud

    src, _, _ = synthetic.make_pairs(2, 4096, seed=11)
    for n, leaf in ((2048, 32), (1800, 32), (640, 32), (4096, 64), (3000, 64), (33, 32)):
        for rule in ("extent", "area"):
            x = src[:, :n].clone()
            assert torch.equal(solver.kd_sort(x.clone(), leaf, rule), plain(x.clone(), leaf, rule)), (n, leaf, rule)
