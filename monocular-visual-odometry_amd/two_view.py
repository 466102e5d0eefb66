"""Two-view geometry on the host without OpenCV: essential matrix from matched pixels, pose
recovery by cheirality, and DLT triangulation.

The reference takes these three steps from OpenCV (cv2.findEssentialMat / cv2.recoverPose at
src/ProjectiveICP.py:68-69, cv2.triangulatePoints at src/geometry_utils.py:75; opencv_python
4.10.0.84, requirements.txt:4).  OpenCV is not part of this package's environment, so they are
implemented here directly.  They run once per frame on at most a few hundred points and are not
part of the GPU hot path.

cv2.findEssentialMat(..., RANSAC, 0.999, 1.0) draws five-point samples with OpenCV's own generator
from a fixed seed, so its result is a deterministic function of the matches.  `find_essential_mat`
restates that published procedure (generator, sampling without repetition, five-point solutions,
float32 Sampson distances, adaptive iteration count); on the bundled sequence it returns the model
of the same first sample, and the bootstrap pose agrees with the one recoverable from the reference's
published outputs/final_results/estimated_trajectory.dat to 3e-10 (tests/test_host_logic.py).

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


class OpenCVRng:
    """cv::RNG, a multiply-with-carry generator: state <- (low 32 bits) * 4164903690 + (high 32
    bits); an integer in [a, b) is next() mod (b - a) + a.  The RANSAC of cv2.findEssentialMat seeds
    it with 2^64 - 1 on every call."""

    def __init__(self, state=0xFFFFFFFFFFFFFFFF):
        self.state = state if state else 0xFFFFFFFF

    def next(self):
        self.state = ((self.state & 0xFFFFFFFF) * 4164903690 + (self.state >> 32)) & 0xFFFFFFFFFFFFFFFF
        return self.state & 0xFFFFFFFF

    def uniform(self, a, b):
        return a if a == b else self.next() % (b - a) + a


# monomials of degree <= 3 in (x, y, z): the ten cubic ones, then the basis of the quotient ring
_MONOMIALS = [(3, 0, 0), (2, 1, 0), (2, 0, 1), (1, 2, 0), (1, 1, 1), (1, 0, 2), (0, 3, 0), (0, 2, 1), (0, 1, 2),
              (0, 0, 3), (2, 0, 0), (1, 1, 0), (1, 0, 1), (0, 2, 0), (0, 1, 1), (0, 0, 2), (1, 0, 0), (0, 1, 0),
              (0, 0, 1), (0, 0, 0)]
_MONOMIAL_INDEX = {m: i for i, m in enumerate(_MONOMIALS)}


def _poly_mul(a, b):
    out = {}
    for ma, ca in a.items():
        for mb, cb in b.items():
            m = (ma[0] + mb[0], ma[1] + mb[1], ma[2] + mb[2])
            out[m] = out.get(m, 0.0) + ca * cb
    return out


def _poly_add(a, b, scale=1.0):
    out = dict(a)
    for m, c in b.items():
        out[m] = out.get(m, 0.0) + scale * c
    return out


def five_point(x0, x1):
    """All real essential matrices (unit Frobenius norm) of five normalised correspondences,
    x1^T E x0 = 0.  E is sought in the four-dimensional null space of the five epipolar constraints,
    E = x E0 + y E1 + z E2 + E3; det E = 0 and 2 E E^T E - tr(E E^T) E = 0 are ten cubic equations in
    (x, y, z), whose (up to ten) solutions are the eigenvectors of the multiplication-by-x matrix
    of the quotient ring (Nister's five-point problem in Stewenius' eigenvalue form; the roots are
    those of Nister's tenth-degree polynomial, which OpenCV solves)."""
    x0 = np.asarray(x0, dtype=np.float64)[:, :2]
    x1 = np.asarray(x1, dtype=np.float64)[:, :2]
    A = np.column_stack([x1[:, 0] * x0[:, 0], x1[:, 0] * x0[:, 1], x1[:, 0], x1[:, 1] * x0[:, 0], x1[:, 1] * x0[:, 1],
                         x1[:, 1], x0[:, 0], x0[:, 1], np.ones(len(x0))])
    _, _, Vt = np.linalg.svd(A)
    basis = [Vt[k].reshape(3, 3) for k in range(5, 9)]
    E = [[{(1, 0, 0): basis[0][r, c], (0, 1, 0): basis[1][r, c], (0, 0, 1): basis[2][r, c], (0, 0, 0): basis[3][r, c]}
          for c in range(3)] for r in range(3)]

    def matmul(P, Q):
        out = [[{} for _ in range(3)] for _ in range(3)]
        for r in range(3):
            for c in range(3):
                for k in range(3):
                    out[r][c] = _poly_add(out[r][c], _poly_mul(P[r][k], Q[k][c]))
        return out

    Et = [[E[c][r] for c in range(3)] for r in range(3)]
    EEt = matmul(E, Et)
    trace = _poly_add(_poly_add(EEt[0][0], EEt[1][1]), EEt[2][2])
    EEtE = matmul(EEt, E)
    cubics = [_poly_add({m: 2.0 * v for m, v in EEtE[r][c].items()}, _poly_mul(trace, E[r][c]), -1.0)
              for r in range(3) for c in range(3)]
    det = {}
    for (a, b, c), sign in (((0, 1, 2), 1.0), ((1, 2, 0), 1.0), ((2, 0, 1), 1.0), ((2, 1, 0), -1.0), ((1, 0, 2), -1.0),
                            ((0, 2, 1), -1.0)):
        det = _poly_add(det, _poly_mul(_poly_mul(E[0][a], E[1][b]), E[2][c]), sign)
    cubics.append(det)
    M = np.zeros((10, 20))
    for i, poly in enumerate(cubics):
        for m, v in poly.items():
            M[i, _MONOMIAL_INDEX[m]] = v
    try:
        B = np.linalg.solve(M[:, :10], M[:, 10:])          # rows: cubic monomial = -B . (quotient basis)
    except np.linalg.LinAlgError:
        return []
    action = np.zeros((10, 10))                             # x * (x^2, xy, xz, y^2, yz, z^2, x, y, z, 1)
    action[0:6] = -B[0:6]
    action[6, 0] = action[7, 1] = action[8, 2] = action[9, 6] = 1.0
    w, V = np.linalg.eig(action)
    out = []
    for k in range(10):
        if abs(w[k].imag) > 1e-10:
            continue
        v = V[:, k].real
        if abs(v[9]) < 1e-14 * max(np.max(np.abs(v)), 1e-300):
            continue
        Ek = (v[6] * basis[0] + v[7] * basis[1] + v[8] * basis[2] + v[9] * basis[3])
        out.append(Ek / np.linalg.norm(Ek))
    return out


def _ransac_update_iterations(confidence, outlier_ratio, model_points, max_iters):
    """Iterations still needed to have drawn an all-inlier sample with the given confidence."""
    p = min(max(confidence, 0.0), 1.0)
    ep = min(max(outlier_ratio, 0.0), 1.0)
    tiny = np.finfo(np.float64).tiny
    num = max(1.0 - p, tiny)
    den = 1.0 - (1.0 - ep) ** model_points
    if den < tiny:
        return 0
    num, den = np.log(num), np.log(den)
    return max_iters if (den >= 0 or -num >= max_iters * (-den)) else int(np.rint(num / den))


def find_essential_mat(points_0, points_1, K, prob=0.999, threshold=1.0, max_iters=1000):
    """Essential matrix and inlier mask of matched pixels as cv2.findEssentialMat(points_0, points_1,
    K, method=RANSAC, prob=0.999, threshold=1.0) determines them (src/ProjectiveICP.py:68):

    * pixels are normalised with the focal lengths and the principal point, the threshold is
      divided by the mean focal length;
    * every iteration draws five distinct indices with OpenCVRng (seeded anew per call, a
      repeated index is drawn again), solves the five-point problem, and scores each solution by
      the number of pairs whose Sampson distance, rounded to float32, is <= threshold^2 (float32);
    * a solution replaces the best one only with strictly more inliers (and at least five), and
      the iteration limit is then lowered to log(1 - prob) / log(1 - inlier_ratio^5), rounded;
    * the best five-point solution is returned as it is (no refit on the inliers).
    """
    K = np.asarray(K, dtype=np.float64)
    p0 = np.asarray(points_0, dtype=np.float64).reshape(-1, 2)
    p1 = np.asarray(points_1, dtype=np.float64).reshape(-1, 2)
    count = len(p0)
    if count < 5:
        raise ValueError(f"need at least 5 matched points for the essential matrix, got {count}")
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    x0 = np.column_stack([(p0[:, 0] - cx) / fx, (p0[:, 1] - cy) / fy, np.ones(count)])
    x1 = np.column_stack([(p1[:, 0] - cx) / fx, (p1[:, 1] - cy) / fy, np.ones(count)])
    thr = threshold / ((fx + fy) / 2.0)
    limit = np.float32(thr * thr)
    rng = OpenCVRng()
    best_E, best_mask, best_count = None, np.zeros(count, dtype=bool), 0
    iterations, it = max(int(max_iters), 1), 0
    while it < iterations:
        if count > 5:
            idx = []
            for _ in range(5):
                k = rng.uniform(0, count)
                while k in idx:
                    k = rng.uniform(0, count)
                idx.append(k)
        else:
            idx = list(range(5))
        for E in five_point(x0[idx], x1[idx]):
            with np.errstate(divide="ignore", invalid="ignore"):
                err = _sampson(E, x0, x1).astype(np.float32)
            mask = err <= limit
            good = int(mask.sum())
            if good > max(best_count, 4):
                best_E, best_mask, best_count = E, mask, good
                iterations = _ransac_update_iterations(prob, (count - good) / count, 5, iterations)
        it += 1
    if best_E is None:
        raise ValueError("no essential matrix found")
    return best_E, best_mask


def essential_from_matches(points_0, points_1, K, threshold=1.0, rounds=3):
    """Least-squares alternative to find_essential_mat (not what the reference computes): fits all
    pairs with the normalised eight-point method, drops those whose Sampson distance exceeds
    ``threshold`` pixels, and refits, a fixed number of rounds."""
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
    """Homogeneous DLT triangulation: per point the right singular vector of the smallest singular
    value of the 4x4 matrix of the two views' constraints (one batched SVD)."""
    x0 = np.asarray(x0, dtype=np.float64)
    x1 = np.asarray(x1, dtype=np.float64)
    if len(x0) == 0:
        return np.zeros((0, 4))
    A = np.stack([x0[:, 0, None] * P0[2] - P0[0], x0[:, 1, None] * P0[2] - P0[1],
                  x1[:, 0, None] * P1[2] - P1[0], x1[:, 1, None] * P1[2] - P1[1]], axis=1)
    return np.linalg.svd(A)[2][:, -1, :]


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
