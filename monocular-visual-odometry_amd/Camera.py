"""Pinhole camera with a mutable world-in-camera pose: the host-side counterpart of the
reference's src/Camera.py (same constructor behaviour, getters and projection semantics).

The LM loop and the association evaluate the projection on the GPU from the values held here
(camera matrix, resolution, c_T_w); the host methods below exist for callers outside the loop.
"""
from __future__ import annotations

import numpy as np


def parse_camera_file(path):
    """Reads the bundled camera description (format of data/camera.dat): a 3x3 integer camera
    matrix after the line 'camera matrix:', a 4x4 'cam_transform:', and the scalars z_near,
    z_far, width, height  (src/Camera.py:8-57).  The camera matrix is parsed with int(), as the
    reference does (:23), so the returned K has an integer dtype."""
    with open(path, "r") as fh:
        rows = [ln.split() for ln in fh.read().splitlines()]
    K, C, z_range, resolution = [], [], [], []
    n = 0
    while n < len(rows):
        tok = rows[n]
        head = " ".join(tok[:2]) if tok else ""
        if head.startswith("camera matrix"):
            K = [[int(x) for x in rows[n + k]] for k in (1, 2, 3)]
            n += 3
        elif tok and tok[0].startswith("cam_transform"):
            C = [[float(x) for x in rows[n + k]] for k in (1, 2, 3, 4)]
            n += 4
        elif tok and tok[0].startswith(("z_near", "z_far")):
            z_range.append(float(tok[1]))
        elif tok and tok[0].startswith(("width", "height")):
            resolution.append(int(tok[1]))
        n += 1
    return np.array(K), np.array(C), z_range, resolution


class Camera:
    def __init__(self, path="data/camera.dat", camera_matrix=None, camera_transform=None, camera_range=None,
                 camera_resolution=None):
        """With no arguments: reads ``data/camera.dat`` relative to the working directory, like the
        reference (src/Camera.py:15); a missing file prints the reference's message and leaves the
        camera empty (:52).  The keyword arguments build a camera without a file."""
        if camera_matrix is not None:
            self.__camera_matrix = np.array(camera_matrix)
            self.__camera_transform = np.array(camera_transform if camera_transform is not None else np.eye(4))
            self.__camera_range = list(camera_range if camera_range is not None else [0.0, 5.0])
            self.__camera_resolution = list(camera_resolution if camera_resolution is not None else [640, 480])
        else:
            try:
                (self.__camera_matrix, self.__camera_transform, self.__camera_range,
                 self.__camera_resolution) = parse_camera_file(path)
            except OSError:
                print("Error: Could not find the camera.dat file")
                self.__camera_matrix, self.__camera_transform = np.array([]), np.array([])
                self.__camera_range, self.__camera_resolution = [], []
        self.__c_T_w = np.eye(4)

    def get_camera_matrix(self): return self.__camera_matrix

    def get_intrinsic_camera_matrix(self):
        P = np.zeros((3, 4))
        P[:3, :3] = self.__camera_matrix
        return P

    def get_camera_transform(self): return self.__camera_transform
    def get_camera_range(self): return self.__camera_range
    def get_camera_resolution(self): return self.__camera_resolution
    def set_c_T_w(self, c_T_w): self.__c_T_w = c_T_w
    def get_c_T_w(self): return self.__c_T_w

    def project_point(self, world_point):
        """(is_inside, pixel | None) for one point  (src/Camera.py:82-121): transform by c_T_w,
        perspective divide, then reject z <= 0 (:111) and pixels outside [0,W) x [0,H) (:117-119).
        The z range of the camera file is not tested (the reference has that test commented out)."""
        ph = self.__c_T_w @ np.append(world_point, 1)
        pc = ph[:3] / ph[3]
        with np.errstate(divide="ignore", invalid="ignore"):
            h = self.__camera_matrix @ pc
            uv = h[:2] / h[2]
        if pc[2] <= 0:
            return False, None
        W, H = self.__camera_resolution
        if uv[0] < 0 or uv[0] >= W or uv[1] < 0 or uv[1] >= H:
            return False, None
        return True, uv

    def project_points(self, world_points):
        """Pixels of the points that project inside the image  (src/Camera.py:123-137)."""
        out = []
        for p in world_points:
            ok, uv = self.project_point(p)
            if ok:
                out.append(uv)
        return out
