"""Synthetic "MVP-shaped" registration pairs (the real MVP_*_RG.h5 files are not available offline).

Follows SURVEY.md 8(d).  Per pair (seeded by ``seed + pair_id``):
  * an asymmetric man-made-looking object: the union of 3-6 randomly posed boxes / plates / cylinders, surface
    sampled uniformly by area and scaled into the radius-0.5 ball (MVP objects are CAD models normalised alike);
  * two partial scans = the points visible from two view directions 30-100 degrees apart (orthographic z-buffer
    hidden-point removal on a 56x56 grid), each re-sampled to exactly ``n_points`` -- partial overlap, like MVP's
    26-view partial clouds and their "match levels";
  * poses with the reference's sampler semantics (registration/dataset.py:16-37): axis = normalised randn,
    angle = U[0,max], translation = normalised randn * U[0,max];  pose1 = random_pose(pi, 0.5),
    transform = random_pose(max_angle, 0.25), pose2 = transform @ pose1, ground truth = transform
    (dataset.py:297-301); 80 % of pairs use max_angle = 45 deg and 20 % use 180 deg (registration/README.md:56-57).
"""
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


def _rand_rot(rng):
    q, r = np.linalg.qr(rng.standard_normal((3, 3)))
    q = q * np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] *= -1
    return q


def _box_surface(rng, n, half):
    """n points uniform on the surface of the axis-aligned box with half-extents `half`."""
    a, b, c = half
    areas = np.array([b * c, b * c, a * c, a * c, a * b, a * b])
    face = rng.choice(6, size=n, p=areas / areas.sum())
    u = rng.uniform(-1, 1, n)
    v = rng.uniform(-1, 1, n)
    p = np.empty((n, 3))
    ax = face // 2
    sgn = np.where(face % 2 == 0, 1.0, -1.0)
    for k in range(3):
        m = ax == k
        o = [i for i in range(3) if i != k]
        p[m, k] = sgn[m] * half[k]
        p[m, o[0]] = u[m] * half[o[0]]
        p[m, o[1]] = v[m] * half[o[1]]
    return p


def _cyl_surface(rng, n, radius, half_h):
    side = 2 * np.pi * radius * 2 * half_h
    cap = np.pi * radius ** 2
    kind = rng.choice(3, size=n, p=np.array([side, cap, cap]) / (side + 2 * cap))
    th = rng.uniform(0, 2 * np.pi, n)
    r = np.where(kind == 0, radius, radius * np.sqrt(rng.random(n)))
    z = np.where(kind == 0, rng.uniform(-half_h, half_h, n), np.where(kind == 1, half_h, -half_h))
    return np.stack([r * np.cos(th), r * np.sin(th), z], 1)


def make_object(rng, n_surface=16384):
    """A random composite object: [n_surface,3] float64 surface samples inside the radius-0.5 ball."""
    n_parts = int(rng.integers(3, 7))
    parts, areas = [], []
    for _ in range(n_parts):
        kind = rng.integers(0, 3)
        R = _rand_rot(rng)
        c = rng.uniform(-0.25, 0.25, 3)
        if kind == 0:      # box
            half = rng.uniform(0.05, 0.3, 3)
            gen = lambda n, half=half: _box_surface(rng, n, half)
            area = 8 * (half[0] * half[1] + half[1] * half[2] + half[0] * half[2])
        elif kind == 1:    # thin plate
            half = np.array([rng.uniform(0.15, 0.4), rng.uniform(0.1, 0.35), rng.uniform(0.008, 0.025)])
            gen = lambda n, half=half: _box_surface(rng, n, half)
            area = 8 * (half[0] * half[1] + half[1] * half[2] + half[0] * half[2])
        else:              # cylinder / leg
            rad, hh = rng.uniform(0.02, 0.12), rng.uniform(0.1, 0.35)
            gen = lambda n, rad=rad, hh=hh: _cyl_surface(rng, n, rad, hh)
            area = 2 * np.pi * rad * 2 * hh + 2 * np.pi * rad ** 2
        parts.append((gen, R, c))
        areas.append(area)
    areas = np.array(areas)
    counts = rng.multinomial(n_surface, areas / areas.sum())
    pts = np.concatenate([gen(k) @ R.T + c for (gen, R, c), k in zip(parts, counts) if k > 0], 0)
    pts -= (pts.max(0) + pts.min(0)) / 2
    pts *= 0.5 / np.linalg.norm(pts, axis=1).max()
    return pts


def partial_view(rng, pts, view_dir, n_points, res=56, tol=0.015):
    """Orthographic z-buffer visibility from `view_dir` (unit), re-sampled to n_points."""
    w = view_dir / np.linalg.norm(view_dir)
    u = np.cross(w, [1.0, 0, 0] if abs(w[0]) < 0.9 else [0, 1.0, 0])
    u /= np.linalg.norm(u)
    v = np.cross(w, u)
    a, b, depth = pts @ u, pts @ v, pts @ w
    ia = np.clip(((a + 0.5) * res).astype(int), 0, res - 1)
    ib = np.clip(((b + 0.5) * res).astype(int), 0, res - 1)
    pix = ia * res + ib
    zmax = np.full(res * res, -np.inf)
    np.maximum.at(zmax, pix, depth)
    vis = np.nonzero(depth >= zmax[pix] - tol)[0]
    pick = rng.choice(vis, size=n_points, replace=len(vis) < n_points)
    return pts[pick]


def make_pair(pair_id, n_points, seed=2021):
    rng = np.random.default_rng(seed + pair_id)
    pts = make_object(rng, max(8 * n_points, 4096))
    v1 = rng.standard_normal(3)
    v1 /= np.linalg.norm(v1)
    while True:
        v2 = rng.standard_normal(3)
        v2 /= np.linalg.norm(v2)
        ang = np.degrees(np.arccos(np.clip(v1 @ v2, -1, 1)))
        if 30.0 <= ang <= 100.0:
            break
    a = partial_view(rng, pts, v1, n_points)
    b = partial_view(rng, pts, v2, n_points)
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
