"""CPU oracle for the ICP refinement row -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The reference's ICP baseline (registration/train_ICP.py:137-153) calls Open3D 0.9.0
(`o3d.registration.registration_icp`, point-to-point, max_correspondence_distance 0.02, max_iteration 500), a
third-party dependency that is neither vendored under /root/reference nor installed here, and the reference holds
no fixture of its output.  This file restates Open3D's PUBLISHED algorithm (Registration.cpp: RegistrationICP,
GetRegistrationResultAndCorrespondences; TransformationEstimationPointToPoint = Eigen::umeyama without scaling) in
numpy float64; the HIP kernel is checked against it, which pins the kernel to this restatement, not to Open3D."""
import numpy as np


def _correspondences(p, tgt, max_dist):
    d2 = ((p[:, None, :] - tgt[None, :, :]) ** 2).sum(-1)
    j = d2.argmin(1)
    dm = d2[np.arange(len(p)), j]
    ok = dm < max_dist * max_dist            # radius search keeps neighbours strictly inside the radius
    fitness = ok.sum() / len(p)
    rmse = float(np.sqrt(dm[ok].sum() / ok.sum())) if ok.any() else 0.0
    return ok, j, fitness, rmse


def _umeyama_no_scale(a, b):
    """Rigid (R, t) minimising |R a + t - b| (Eigen::umeyama(src, dst, false))."""
    ma, mb = a.mean(0), b.mean(0)
    S = (b - mb).T @ (a - ma) / len(a)
    U, _, Vt = np.linalg.svd(S)
    D = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        D[2, 2] = -1
    R = U @ D @ Vt
    return R, mb - R @ ma


def icp_point_to_point(src, tgt, init=None, max_correspondence_distance=0.02, max_iteration=500,
                       relative_fitness=1e-6, relative_rmse=1e-6):
    """src[N,3], tgt[M,3] float -> (T[4,4], fitness, inlier_rmse, iterations)."""
    src = np.asarray(src, np.float64)
    tgt = np.asarray(tgt, np.float64)
    T = np.eye(4) if init is None else np.asarray(init, np.float64).copy()
    p = src @ T[:3, :3].T + T[:3, 3]
    ok, j, fit, rmse = _correspondences(p, tgt, max_correspondence_distance)
    it = 0
    for it in range(1, max_iteration + 1):
        if not ok.any():
            it -= 1
            break
        R, t = _umeyama_no_scale(p[ok], tgt[j[ok]])
        U = np.eye(4)
        U[:3, :3], U[:3, 3] = R, t
        T = U @ T
        p = p @ R.T + t
        pf, pr = fit, rmse
        ok, j, fit, rmse = _correspondences(p, tgt, max_correspondence_distance)
        if abs(pf - fit) < relative_fitness and abs(pr - rmse) < relative_rmse:
            break
    return T, fit, rmse, it
