"""Data side of the path: the pose samplers of registration/dataset.py:16-52 and Dataset classes with the tuple layouts the
HOUV drivers unpack (val: the 17-tuple of dataset.py:346, test: (src, tgt, label) of :348; sharded [l:r] slices of
MVP_RG_rotated_bound, :369-372).  Real MVP ``*.h5`` files are read by houv_amd.io (h5py or hdf5_min); ``SyntheticRG`` serves MVP-shaped
synthetic pairs with the same layouts when they are absent."""
import numpy as np
import torch
from torch.utils.data import Dataset

from . import io as hio
from . import synthetic


def random_rotation(max_angle):
    """dataset.py:22-30."""
    axis = np.random.randn(3)
    axis /= np.linalg.norm(axis)
    angle = np.random.rand() * max_angle
    A = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * A + (1 - np.cos(angle)) * np.dot(A, A), angle


def random_translation(max_dist):
    """dataset.py:33-37."""
    t = np.random.randn(3)
    t /= np.linalg.norm(t)
    t *= np.random.rand() * max_dist
    return np.expand_dims(t, 1)


def random_pose(max_angle, max_trans):
    """dataset.py:16-19: 4x4 pose and its rotation angle."""
    R, angle = random_rotation(max_angle)
    t = random_translation(max_trans)
    return np.concatenate([np.concatenate([R, t], 1), [[0, 0, 0, 1]]], 0), angle


def rotation_angle_deg(R):
    """Angle of a rotation matrix in degrees (trace form; translation_back below is the reference's asin form)."""
    return float(np.degrees(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))))


def translation_back(R):
    """Rotation angle in degrees and axis of a rotation matrix, the way the reference's datasets derive `add_ps`
    (registration/train_utils.py:1019-1032: asin of the antisymmetric part, quadrant from the symmetric part)."""
    tran = 0.5 * (R - R.T)
    theta_sin = np.sqrt(tran[0][1] ** 2 + tran[0][2] ** 2 + tran[1][2] ** 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        axis = np.array([[-tran[1][2] / theta_sin, tran[0][2] / theta_sin, tran[0][1] / theta_sin]])
    axis_matrix = np.dot(axis.T, axis)
    np_matrix = R - tran - axis_matrix
    cos = np.sqrt(max(0.0, 1. - theta_sin ** 2)) * (np.ones((3, 3)) - axis_matrix)
    result1 = np.square(np_matrix - cos).sum()
    result2 = np.square(np_matrix + cos).sum()
    theta = (np.pi - np.arcsin(min(theta_sin, 1.0))) if result1 >= result2 else np.arcsin(min(theta_sin, 1.0))
    return theta * 180 / np.pi, axis


class _PairsBase(Dataset):
    """Shared item logic.  ``bound`` selects the tuple layout of MVP_RG_rotated_bound (8 slots, dataset.py:476) instead
    of MVP_RG_rotated's 17 (dataset.py:346)."""
    bound = False

    def __init__(self, prefix, src, tgt, transforms=None, labels=None, rotated=None, extra=None):
        self.prefix = prefix
        self.src, self.tgt, self.transforms = src, tgt, transforms
        self.label = labels if labels is not None else np.zeros(len(src), np.int32)
        # val: (rotated_src, rotated_tgt) are what the solver sees; src/tgt are the un-rotated clouds (dataset.py:222-226)
        self.src_rotated, self.tgt_rotated = rotated if rotated is not None else (src, tgt)
        self.extra = extra or {}

    def __len__(self):
        return self.src.shape[0]

    def __getitem__(self, index):
        s = torch.from_numpy(np.asarray(self.src[index], np.float32))
        t = torch.from_numpy(np.asarray(self.tgt[index], np.float32))
        label = torch.from_numpy(np.array([self.label[index]]))
        if self.prefix == "test":
            return s, t, label                                                   # dataset.py:348 / :478
        sr = torch.from_numpy(np.asarray(self.src_rotated[index], np.float32))
        tr = torch.from_numpy(np.asarray(self.tgt_rotated[index], np.float32))
        T = torch.from_numpy(np.asarray(self.transforms[index], np.float32))
        a, _ = translation_back(np.asarray(self.transforms[index], np.float64)[:3, :3])
        add_ps = torch.ones(1) if a > 45 else torch.zeros(1)
        if self.bound:
            return s, t, sr, tr, T, label, add_ps, a                             # dataset.py:476
        ex = self.extra

        def pick(name, default):
            return torch.from_numpy(np.asarray(ex[name][index])) if name in ex else default
        if "complete" in ex:                                                     # dataset.py:321-323: val serves `complete` as src
            s = torch.from_numpy(np.asarray(ex["complete"][index], np.float32))
        z = torch.zeros(1)
        match_level = int(ex["match_level"][index]) if "match_level" in ex else 0
        rot_level = int(ex["rot_level"][index]) if "rot_level" in ex else int(a > 45)
        # (src, tgt, src_rotated, tgt_rotated, transform, match_level, rot_level, pose1, pose2, angle_t, label,
        #  src_vox, tgt_vox, src_vox_len, tgt_vox_len, add_ps, angle)  -- HOUV reads slots 2, 3, 4 (train_HOUV.py:92-112);
        #  the four voxel slots feed other models (dataset.py:264-288) and are placeholders here.
        return (s, t, sr, tr, T, match_level, rot_level, pick("pose_src", torch.eye(4)), pick("pose_tgt", T),
                torch.from_numpy(np.array([-1])), label, z, z, z, z, add_ps, a)


class SyntheticRG(_PairsBase):
    """MVP-shaped synthetic pairs (houv_amd.synthetic) behind the reference's Dataset layouts."""

    def __init__(self, prefix, args, n_pairs=100, first_id=0):
        s, t, T = synthetic.make_pairs(n_pairs, int(getattr(args, "num_points", 2048)),
                                       seed=int(getattr(args, "manual_seed", 2021) or 2021), first_id=first_id)
        super().__init__(prefix, s.numpy(), t.numpy(), T.numpy())


class MVP_RG_rotated(_PairsBase):
    """registration/dataset.py:189-348 for prefix in {"val", "test"} (the splits HOUV's drivers read; "train" applies
    random poses on the fly for the learned baselines and is out of scope).  val: MVP_Test_RG.h5 -- the solver sees
    rotated_src / rotated_tgt, ground truth = transforms (:222-232, :312-323); test: MVP_ExtraTest_RG.h5 -- rotated_src /
    rotated_tgt only (:205-207).  The files are opened where the reference opens them (./data, or cfg ``data_dir``)."""

    def __init__(self, prefix, args, l=None, r=None):
        if prefix not in ("val", "test"):
            raise ValueError("houv_amd serves the 'val' and 'test' splits (HOUV is training-free)")
        d = hio.load_mvp_rg(hio.mvp_path(prefix, args), l, r)
        labels = d["cat_labels"].astype("int32") if "cat_labels" in d else None
        f32 = lambda k: d[k].astype("float32")           # noqa: E731
        if prefix == "test":
            super().__init__(prefix, f32("rotated_src"), f32("rotated_tgt"), None, labels)
        else:
            extra = {k: d[k] for k in ("complete", "pose_src", "pose_tgt", "match_level", "rot_level") if k in d}
            super().__init__(prefix, f32("src"), f32("tgt"), f32("transforms"), labels,
                             rotated=(f32("rotated_src"), f32("rotated_tgt")), extra=extra)
        category = getattr(args, "category", None)
        if category:                                      # dataset.py:240-251
            keep = self.label == category
            self.src, self.tgt, self.label = self.src[keep], self.tgt[keep], self.label[keep]
            self.src_rotated, self.tgt_rotated = self.src_rotated[keep], self.tgt_rotated[keep]
            if self.transforms is not None:
                self.transforms = self.transforms[keep]
            self.extra = {k: v[keep] for k, v in self.extra.items()}


class MVP_RG_rotated_bound(MVP_RG_rotated):
    """dataset.py:354-478: the [l : r] shard of the set (run_test.sh:6); 8-slot val tuples (:476).  l / r default to the
    config's (test_mult.py:98-99)."""
    bound = True

    def __init__(self, prefix, args, l=None, r=None):
        super().__init__(prefix, args, int(args.l) if l is None else l, int(args.r) if r is None else r)


def open_pairs(prefix, args, l=None, r=None, n_synthetic=None):
    """What the drivers iterate: the MVP split when its h5 file exists, MVP-shaped synthetic pairs otherwise (said in the
    return value so that the driver can log it)."""
    import os
    path = hio.mvp_path(prefix, args)
    if os.path.exists(path):
        ds = MVP_RG_rotated_bound(prefix, args, l, r) if (l is not None or r is not None) else MVP_RG_rotated(prefix, args)
        return ds, path
    lo = 0 if l is None else int(l)
    n = (int(r) - lo) if r is not None else int(n_synthetic or 100)
    ds = SyntheticRG(prefix, args, n_pairs=n, first_id=lo)
    if l is not None or r is not None:
        ds.bound = True
    return ds, None
