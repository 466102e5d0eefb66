"""Small pose utilities on the host (numpy): the counterparts of the reference's
src/geometry_utils.py that the drivers around the hot path need.

Inside the LM loop these are evaluated on the GPU (csrc/picp_device.h); the functions here serve
the code around it: ground-truth poses, evaluation transforms, two-view bootstrap.
"""
from __future__ import annotations

import numpy as np


def _rot(axis: int, theta) -> np.ndarray:
    """Elementary rotation about x (0), y (1) or z (2)  (src/geometry_utils.py:4-17).

    The result is float64; ``theta`` may be a float32 scalar, in which case its sine and cosine
    are float32 values widened on assignment -- the behaviour v2T depends on."""
    c, s = np.cos(theta), np.sin(theta)
    R = np.eye(3)
    i, j = [(1, 2), (2, 0), (0, 1)][axis]
    R[i, i] = c
    R[j, j] = c
    R[i, j] = -s
    R[j, i] = s
    return R


def Rx(theta): return _rot(0, theta)
def Ry(theta): return _rot(1, theta)
def Rz(theta): return _rot(2, theta)


def euler2R(euler) -> np.ndarray:
    """R = Rx Ry Rz  (src/geometry_utils.py:19-23)."""
    return Rx(euler[0]) @ Ry(euler[1]) @ Rz(euler[2])


def v2T(v) -> np.ndarray:
    """6-vector (translation, Euler angles) -> 4x4  (src/geometry_utils.py:38-45).

    The angles are cast to float32 before the trigonometry (:41); this is part of the reference's
    numerics (it is visible in the published rotation ratios) and is kept."""
    v = np.asarray(v, dtype=np.float64)
    T = np.eye(4)
    T[:3, :3] = euler2R(v[3:6].astype(np.float32))
    T[:3, 3] = v[:3]
    return T


def Rt2T(R, t) -> np.ndarray:
    """src/geometry_utils.py:53-57."""
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = np.asarray(t, dtype=np.float64).reshape(3)
    return T


def skew(w) -> np.ndarray:
    """src/geometry_utils.py:59-62."""
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def transform(poses, T=None, scale=1, are_points=False):
    """Scale the translation of every pose (or point), then left-multiply by T
    (src/geometry_utils.py:93-109)."""
    T = np.eye(4) if T is None else np.asarray(T, dtype=np.float64)
    out = []
    for pose in poses:
        if are_points:
            P = np.eye(4)
            P[:3, 3] = np.asarray(pose, dtype=np.float64)[:3]
        else:
            P = np.array(pose, dtype=np.float64)
        P[:3, 3] = P[:3, 3] * scale
        P = T @ P
        out.append(P[:3, 3] if are_points else P)
    return out


def poses2positions(poses):
    """src/geometry_utils.py:111-115."""
    return [np.asarray(p)[:3, 3] for p in poses]


# src/geometry_utils.py:64-91 is reachable from this module in the reference (its ProjectiveICP takes it from
# `from src.geometry_utils import *`, src/ProjectiveICP.py:2, used at :74 and :109).  The DLT itself lives in
# two_view.py, where it replaces cv2.triangulatePoints.
from .two_view import triangulate_points  # noqa: E402,F401
