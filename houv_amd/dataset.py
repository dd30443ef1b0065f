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
    """Angle of a rotation matrix in degrees (what train_utils.translation_back feeds `add_ps`, dataset.py:334-339)."""
    return float(np.degrees(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))))


class _PairsBase(Dataset):
    def __init__(self, prefix, src, tgt, transforms=None, labels=None):
        self.prefix = prefix
        self.src, self.tgt, self.transforms = src, tgt, transforms
        self.label = labels if labels is not None else np.zeros(len(src), np.int32)

    def __len__(self):
        return self.src.shape[0]

    def __getitem__(self, index):
        s = torch.from_numpy(np.asarray(self.src[index], np.float32))
        t = torch.from_numpy(np.asarray(self.tgt[index], np.float32))
        label = torch.from_numpy(np.array([self.label[index]]))
        if self.prefix == "test":
            return s, t, label                                                   # dataset.py:348
        T = torch.from_numpy(np.asarray(self.transforms[index], np.float32))
        a = rotation_angle_deg(np.asarray(self.transforms[index])[:3, :3])
        add_ps = torch.ones(1) if a > 45 else torch.zeros(1)
        z = torch.zeros(1)
        eye = torch.eye(4)
        # (src, tgt, src_rotated, tgt_rotated, transform, match_level, rot_level, pose1, pose2, angle_t, label,
        #  src_vox, tgt_vox, src_vox_len, tgt_vox_len, add_ps, angle)  -- HOUV reads slots 2, 3, 4 (train_HOUV.py:92-112);
        #  the voxel slots belong to other models and are placeholders here.
        return (s, t, s, t, T, 0, int(a > 45), eye, T, torch.tensor([-1.0]), label, z, z, z, z, add_ps, a)


class SyntheticRG(_PairsBase):
    """MVP-shaped synthetic pairs (houv_amd.synthetic) behind the reference's Dataset layouts."""

    def __init__(self, prefix, args, n_pairs=100, first_id=0):
        s, t, T = synthetic.make_pairs(n_pairs, int(getattr(args, "num_points", 2048)),
                                       seed=int(getattr(args, "manual_seed", 2021) or 2021), first_id=first_id)
        super().__init__(prefix, s.numpy(), t.numpy(), T.numpy())


class MVP_RG_rotated(_PairsBase):
    """registration/dataset.py:189-348 for prefix in {"val", "test"} (what HOUV's drivers use): needs the MVP h5 files
    (not shipped).  val serves the stored rotated clouds + transforms (:312-323), test the rotated test clouds (:205-207)."""
    FILES = {"train": "./data/MVP_Train_RG.h5", "val": "./data/MVP_Test_RG.h5", "test": "./data/MVP_ExtraTest_RG.h5"}

    def __init__(self, prefix, args, l=None, r=None):
        d = hio.load_mvp_rg(self.FILES[prefix], l, r)
        labels = d.get("cat_labels")
        if prefix == "test":
            super().__init__(prefix, d["rotated_src"], d["rotated_tgt"], None, labels)
        else:
            super().__init__(prefix, d["rotated_src"], d["rotated_tgt"], d["transforms"], labels)


class MVP_RG_rotated_bound(MVP_RG_rotated):
    """dataset.py:354-478: the [args.l : args.r] shard of the set (run_test.sh:6)."""

    def __init__(self, prefix, args):
        super().__init__(prefix, args, int(args.l), int(args.r))
