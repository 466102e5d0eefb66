"""Appearance data association on the GPU, behind the reference's function signature
(src/data_association.py:3-57).

The matching itself -- all pairs of descriptors that are equal component by component, in
query-major order -- and the optional in-image test run as HIP kernels (csrc/picp_assoc.hip);
this module converts the reference's dict-of-lists arguments to device arrays and the result
back to the dict-of-lists the callers expect.
"""
from __future__ import annotations

import numpy as np

from ._lib import camera_struct, get_context


def _as_matrix(rows):
    """list of equal-length float lists -> (n, D) float64 array (D = 0 for an empty list)."""
    if len(rows) == 0:
        return np.zeros((0, 0))
    a = np.asarray(rows, dtype=np.float64)
    return a.reshape(len(rows), -1)


def match_indices(appearance_1, appearance_2, ctx=None):
    """Index form of the association: int64 arrays (idx1, idx2) of all equal descriptor pairs in
    the order of the reference's double loop (src/data_association.py:29-31)."""
    a1, a2 = _as_matrix(appearance_1), _as_matrix(appearance_2)
    if a1.shape[0] == 0 or a2.shape[0] == 0 or a1.shape[1] != a2.shape[1] or a1.shape[1] == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    ctx = ctx or get_context()
    i1, i2 = ctx.associate(ctx.to_device(a1), ctx.to_device(a2))
    return i1.cpu().numpy(), i2.cpu().numpy()


def data_association_on_appearance(set_1, set_2, projection=0, camera=None):
    """Same arguments and return value as the reference function:

    ``set_1`` / ``set_2`` are dicts with 'position' and 'appearance' lists.  Returns a dict of
    lists, one entry per matched pair: 'points_1', 'points_2', 'appearance' and, when
    ``projection`` is 1 or 2, 'projected_points_1' / 'projected_points_2' -- in that case pairs
    whose set_1 / set_2 position does not project inside ``camera``'s image are dropped
    (src/data_association.py:36-50)."""
    points_1, appearance_1 = set_1["position"], set_1["appearance"]
    points_2, appearance_2 = set_2["position"], set_2["appearance"]
    ctx = get_context()
    idx1, idx2 = match_indices(appearance_1, appearance_2, ctx)
    if projection in (1, 2):
        key = "projected_points_1" if projection == 1 else "projected_points_2"
        src, idx = (points_1, idx1) if projection == 1 else (points_2, idx2)
        projected = []
        if len(idx):
            pts = np.asarray([src[k] for k in idx], dtype=np.float64).reshape(len(idx), 3)
            W, H = camera.get_camera_resolution()
            cam = camera_struct(camera.get_camera_matrix(), W, H)
            inside, u, v = ctx.project(ctx.to_device(np.ascontiguousarray(pts.T)), camera.get_c_T_w(), cam)
            keep = np.nonzero(inside.cpu().numpy())[0]
            uv = np.stack([u.cpu().numpy(), v.cpu().numpy()], axis=1)
            projected = [uv[k].copy() for k in keep]
            idx1, idx2 = idx1[keep], idx2[keep]
        matches = {"points_1": [points_1[i] for i in idx1], key: projected,
                   "points_2": [points_2[j] for j in idx2], "appearance": [appearance_1[i] for i in idx1]}
        if projection == 2:   # the reference's key order (src/data_association.py:26)
            matches = {k: matches[k] for k in ("points_1", "points_2", key, "appearance")}
        return matches
    return {"points_1": [points_1[i] for i in idx1], "points_2": [points_2[j] for j in idx2],
            "appearance": [appearance_1[i] for i in idx1]}
