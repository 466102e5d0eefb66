"""Seeded synthetic correspondence sets shared by the tests, smoke() and bench.py.

World points are spread over the frustum of the identity pose (depth 1..20 m, pixel uniform over
the 640x480 image, K of the bundled camera); measurements are their projections under the true
pose T* = v2T([0.05,-0.03,0.2, 0.01,-0.02,0.015]) plus N(0, 0.5 px) noise; a tenth of the
measurements is replaced by uniform pixels (gross outliers) so both inlier stages have work.
"""
import numpy as np

K = np.array([[180.0, 0.0, 320.0], [0.0, 180.0, 240.0], [0.0, 0.0, 1.0]])
TRUE_V = (0.05, -0.03, 0.2, 0.01, -0.02, 0.015)


def true_pose(v=TRUE_V):
    a = np.asarray(v[3:], dtype=np.float64)
    cx, sx, cy, sy, cz, sz = np.cos(a[0]), np.sin(a[0]), np.cos(a[1]), np.sin(a[1]), np.cos(a[2]), np.sin(a[2])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rx @ Ry @ Rz
    T[:3, 3] = v[:3]
    return T


def make_problem(M, seed=0, outlier_frac=0.1, noise=0.5, v=TRUE_V, dtype=np.float64):
    """Returns (uv[M,2], pw[M,3], T_true)."""
    rng = np.random.default_rng(seed)
    T = true_pose(v)
    z = rng.uniform(1.0, 20.0, M)
    px = rng.uniform(0.0, 640.0, M)
    py = rng.uniform(0.0, 480.0, M)
    pw = np.stack([(px - 320.0) / 180.0 * z, (py - 240.0) / 180.0 * z, z], axis=1)
    c_T_w = np.linalg.inv(T)
    pc = pw @ c_T_w[:3, :3].T + c_T_w[:3, 3]
    h = pc @ K.T
    uv = h[:, :2] / h[:, 2:3] + rng.normal(0.0, noise, (M, 2))
    n_out = int(M * outlier_frac)
    if n_out:
        idx = rng.choice(M, n_out, replace=False)
        uv[idx, 0] = rng.uniform(0.0, 640.0, n_out)
        uv[idx, 1] = rng.uniform(0.0, 480.0, n_out)
    return uv.astype(dtype), pw.astype(dtype), T
