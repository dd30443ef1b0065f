"""Synthetic "MVP-shaped" registration pairs (the real MVP_*_RG.h5 files are not available offline).

Follows SURVEY.md 8(d): per pair a closed surface inside the radius-0.5 ball; the two partial views are the
points with the largest projection on two view directions at least 30 degrees apart (so the overlap is partial,
like MVP's match levels); poses follow the reference's sampler semantics (registration/dataset.py:16-37):
axis = normalised randn, angle = U[0,max], translation = normalised randn * U[0,max];
pose2 = transform @ pose1 and the ground truth is `transform` (dataset.py:297-301).  80 % of pairs draw the
relative rotation from [0,45] degrees and 20 % from [0,180] (registration/README.md:56-57)."""
import numpy as np
import torch


def _pose(rng, max_angle, max_trans):
    axis = rng.standard_normal(3)
    axis /= np.linalg.norm(axis)
    angle = rng.random() * max_angle
    A = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + np.sin(angle) * A + (1 - np.cos(angle)) * (A @ A)
    t = rng.standard_normal(3)
    t /= np.linalg.norm(t)
    t *= rng.random() * max_trans
    P = np.eye(4)
    P[:3, :3] = R
    P[:3, 3] = t
    return P


def make_pair(pair_id, n_points, seed=2021, dense_factor=4):
    rng = np.random.default_rng(seed + pair_id)
    m = max(dense_factor * n_points, 64)
    d = rng.standard_normal((m, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # a bumpy star-shaped closed surface: radius in [0.25, 0.5]
    k = rng.integers(1, 4, size=3)
    ph = rng.random(3) * 2 * np.pi
    r = 0.375 + 0.125 * np.sin(k[0] * np.arctan2(d[:, 1], d[:, 0]) + ph[0]) * np.cos(k[1] * np.arccos(np.clip(d[:, 2], -1, 1)) + ph[1])
    pts = d * r[:, None]
    v1 = rng.standard_normal(3)
    v1 /= np.linalg.norm(v1)
    while True:
        v2 = rng.standard_normal(3)
        v2 /= np.linalg.norm(v2)
        ang = np.degrees(np.arccos(np.clip(v1 @ v2, -1, 1)))
        if 30.0 <= ang <= 100.0:
            break
    a = pts[np.argsort(-(pts @ v1))[:n_points]]
    b = pts[np.argsort(-(pts @ v2))[:n_points]]
    a = a[rng.permutation(n_points)]
    b = b[rng.permutation(n_points)]
    max_angle = np.pi / 4 if (pair_id % 5) != 4 else np.pi
    pose1 = _pose(rng, np.pi, 0.5)
    transform = _pose(rng, max_angle, 0.25)
    pose2 = transform @ pose1
    src = a @ pose1[:3, :3].T + pose1[:3, 3]
    tgt = b @ pose2[:3, :3].T + pose2[:3, 3]
    return src.astype(np.float32), tgt.astype(np.float32), transform.astype(np.float32)


def make_pairs(n_pairs, n_points, seed=2021, first_id=0):
    """-> src[P,N,3], tgt[P,N,3], transform[P,4,4] (CPU float32 tensors)."""
    out = [make_pair(first_id + i, n_points, seed) for i in range(n_pairs)]
    return (torch.from_numpy(np.stack([o[0] for o in out])), torch.from_numpy(np.stack([o[1] for o in out])),
            torch.from_numpy(np.stack([o[2] for o in out])))
