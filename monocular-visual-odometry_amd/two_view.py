"""Two-view geometry on the host without OpenCV: essential matrix from matched pixels, pose
recovery by cheirality, and DLT triangulation.

The reference takes these three steps from OpenCV (cv2.findEssentialMat / cv2.recoverPose at
src/ProjectiveICP.py:68-69, cv2.triangulatePoints at src/geometry_utils.py:75).  OpenCV is not
part of this package's environment, so they are implemented here directly.  They run once per
frame on at most a few hundred points and are not part of the GPU hot path.

Conventions are OpenCV's, so the call sites read like the reference's: for matched pixels
x0 (first view) and x1 (second view), x1^T K^-T E K^-1 x0 = 0, and (R, t) maps first-camera
coordinates to second-camera coordinates, X1 = R X0 + t with |t| = 1.
"""
from __future__ import annotations

import numpy as np


def _normalized(points, K):
    """Pixels -> normalised image coordinates (homogeneous, third component 1)."""
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 2)
    Kinv = np.linalg.inv(np.asarray(K, dtype=np.float64))
    h = np.column_stack([pts, np.ones(len(pts))])
    return (Kinv @ h.T).T


def _eight_point(x0, x1):
    """Least-squares essential matrix of normalised correspondences, projected onto the essential
    manifold (two equal singular values, one zero)."""
    def hartley(x):
        c = x[:, :2].mean(axis=0)
        s = np.sqrt(2.0) / max(np.mean(np.linalg.norm(x[:, :2] - c, axis=1)), 1e-300)
        return np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.0]])
    T0, T1 = hartley(x0), hartley(x1)
    a, b = (T0 @ x0.T).T, (T1 @ x1.T).T
    A = np.column_stack([b[:, 0] * a[:, 0], b[:, 0] * a[:, 1], b[:, 0], b[:, 1] * a[:, 0], b[:, 1] * a[:, 1],
                         b[:, 1], a[:, 0], a[:, 1], np.ones(len(a))])
    _, _, Vt = np.linalg.svd(A)
    E = T1.T @ Vt[-1].reshape(3, 3) @ T0
    U, S, Vt = np.linalg.svd(E)
    s = 0.5 * (S[0] + S[1])
    return U @ np.diag([s, s, 0.0]) @ Vt


def _sampson(E, x0, x1):
    Ex0 = (E @ x0.T).T
    Etx1 = (E.T @ x1.T).T
    num = np.sum(x1 * Ex0, axis=1) ** 2
    den = Ex0[:, 0] ** 2 + Ex0[:, 1] ** 2 + Etx1[:, 0] ** 2 + Etx1[:, 1] ** 2
    return num / np.maximum(den, 1e-300)


def essential_from_matches(points_0, points_1, K, threshold=1.0, rounds=3):
    """Essential matrix and inlier mask of matched pixels.

    Stands where the reference calls cv2.findEssentialMat(points_0, points_1, K, RANSAC, 0.999, 1.0)
    (src/ProjectiveICP.py:68).  The bundled matches come from descriptor equality and contain no
    wrong pairs, so instead of random sampling this fits all pairs, drops those whose Sampson
    distance exceeds ``threshold`` pixels, and refits, a fixed number of rounds; the result is
    deterministic."""
    K = np.asarray(K, dtype=np.float64)
    x0, x1 = _normalized(points_0, K), _normalized(points_1, K)
    if len(x0) < 8:
        raise ValueError(f"need at least 8 matched points for the essential matrix, got {len(x0)}")
    focal = 0.5 * (K[0, 0] + K[1, 1])
    mask = np.ones(len(x0), dtype=bool)
    E = _eight_point(x0, x1)
    for _ in range(rounds):
        d = np.sqrt(_sampson(E, x0, x1)) * focal
        new_mask = d <= threshold
        if new_mask.sum() < 8 or np.array_equal(new_mask, mask):
            break
        mask = new_mask
        E = _eight_point(x0[mask], x1[mask])
    return E, mask


def _triangulate_normalized(P0, P1, x0, x1):
    """Homogeneous DLT triangulation, one 4x4 null-space problem per point."""
    X = np.empty((len(x0), 4))
    for k in range(len(x0)):
        A = np.stack([x0[k, 0] * P0[2] - P0[0], x0[k, 1] * P0[2] - P0[1],
                      x1[k, 0] * P1[2] - P1[0], x1[k, 1] * P1[2] - P1[1]])
        _, _, Vt = np.linalg.svd(A)
        X[k] = Vt[-1]
    return X


def recover_pose(E, points_0, points_1, K, mask=None, distance_threshold=50.0):
    """(R, t, mask): the decomposition of E that puts the triangulated points in front of both
    cameras  (the role of cv2.recoverPose at src/ProjectiveICP.py:69).  t has unit norm and is
    returned as a (3, 1) column like OpenCV's."""
    x0, x1 = _normalized(points_0, K), _normalized(points_1, K)
    mask = np.ones(len(x0), dtype=bool) if mask is None else np.asarray(mask, dtype=bool).reshape(-1)
    U, _, Vt = np.linalg.svd(E)
    if np.linalg.det(U) < 0:
        U = -U
    if np.linalg.det(Vt) < 0:
        Vt = -Vt
    W = np.array([[0.0, 1.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    Ra, Rb, t = U @ W @ Vt, U @ W.T @ Vt, U[:, 2]
    P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
    best = None
    for R, tt in ((Ra, t), (Rb, t), (Ra, -t), (Rb, -t)):
        P1 = np.hstack([R, tt.reshape(3, 1)])
        X = _triangulate_normalized(P0, P1, x0, x1)
        with np.errstate(divide="ignore", invalid="ignore"):
            X0 = X[:, :3] / X[:, 3:4]
        X1 = (R @ X0.T).T + tt
        good = mask & (X0[:, 2] > 0) & (X1[:, 2] > 0) & (X0[:, 2] < distance_threshold) & (X1[:, 2] < distance_threshold)
        if best is None or good.sum() > best[0]:
            best = (good.sum(), R, tt, good)
    _, R, tt, good = best
    return R, tt.reshape(3, 1) / np.linalg.norm(tt), good


def triangulate_points(points_0, points_1, w_T_c0, w_T_c1, K, threshold=30):
    """Triangulates matched pixels of two posed views  (src/geometry_utils.py:64-91).

    The points are triangulated in the first view's camera frame by homogeneous DLT (what
    cv2.triangulatePoints computes, :75), kept when they lie in front of the camera (:78) and
    closer than ``threshold`` (:80-81), and then moved to the map frame with w_T_c0 (:88-89).
    Returns (points[N_kept, 3], mask[N])."""
    K = np.asarray(K, dtype=np.float64)
    points_0 = np.asarray(points_0, dtype=np.float64).reshape(-1, 2)
    points_1 = np.asarray(points_1, dtype=np.float64).reshape(-1, 2)
    T = np.linalg.inv(w_T_c1) @ w_T_c0
    P0 = K @ np.hstack([np.eye(3), np.zeros((3, 1))])
    P1 = K @ np.hstack([T[:3, :3], T[:3, 3].reshape(3, 1)])
    if len(points_0) == 0:
        return np.zeros((0, 3)), np.zeros(0, dtype=bool)
    X = _triangulate_normalized(P0, P1, points_0, points_1)
    with np.errstate(divide="ignore", invalid="ignore"):
        p = X[:, :3] / X[:, 3:4]
        mask = (p[:, 2] > 0) & (np.linalg.norm(p, axis=1) < threshold)
    kept = np.column_stack([p[mask], np.ones(int(mask.sum()))])
    world = (w_T_c0 @ kept.T)
    return (world[:3] / world[3]).T, mask
