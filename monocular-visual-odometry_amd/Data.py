"""Readers for the bundled sequence (trajectory.dat, world.dat, meas-%05d.dat): host-side
counterpart of the reference's src/Data.py with the same getters and return shapes.
"""
from __future__ import annotations

import os

from .geometry_utils import v2T


def read_trajectory(path):
    """Rows 'id odom_x odom_y odom_theta gt_x gt_y gt_theta' -> list of [gt_x, gt_y, gt_theta]
    (the reference keeps tokens[4:], src/Data.py:19)."""
    with open(path, "r") as fh:
        return [[float(x) for x in line.split()[4:]] for line in fh if line.strip()]


def read_world(path):
    """Rows 'id x y z a0..a9' -> {'position': [[x,y,z]..], 'appearance': [[a0..a9]..]}  (src/Data.py:33-39)."""
    world = {"position": [], "appearance": []}
    with open(path, "r") as fh:
        for line in fh:
            tok = line.split()
            if not tok:
                continue
            world["position"].append([float(x) for x in tok[1:4]])
            world["appearance"].append([float(x) for x in tok[4:]])
    return world


def read_measurement(path):
    """Lines 'point <id in frame> <id in world> u v a0..a9' -> dict of lists  (src/Data.py:55-69)."""
    m = {"current_point_id": [], "actual_point_id": [], "position": [], "appearance": []}
    with open(path, "r") as fh:
        for line in fh:
            if not line.startswith("point"):
                continue
            tok = line.split()
            m["current_point_id"].append(int(tok[1]))
            m["actual_point_id"].append(int(tok[2]))
            m["position"].append([float(x) for x in tok[3:5]])
            m["appearance"].append([float(x) for x in tok[5:]])
    return m


class Data:
    def __init__(self, directory="data", num_measurements=121):
        """Reads ``data/`` relative to the working directory like the reference (src/Data.py:14,30,52);
        the reference's count of 121 measurement files (:48) is the default.  A missing file prints
        the reference's message and is skipped."""
        try:
            self.__trajectory_data = read_trajectory(os.path.join(directory, "trajectory.dat"))
        except OSError:
            print("Error: Could not find the trajectory.dat file")
            self.__trajectory_data = []
        try:
            self.__world_data = read_world(os.path.join(directory, "world.dat"))
        except OSError:
            print("Error: Could not find the world.dat file")
            self.__world_data = {"position": [], "appearance": []}
        self.__measurements_data = []
        for i in range(num_measurements):
            name = f"meas-{i:05d}.dat"
            try:
                self.__measurements_data.append(read_measurement(os.path.join(directory, name)))
            except OSError:
                print(f"Error: Could not find the {name} file")

    def get_trajectory_data(self): return self.__trajectory_data

    def get_trajectory_data_poses(self):
        """Ground-truth poses as 4x4: planar (x, y, theta) -> v2T([x, y, 0, 0, 0, theta])  (src/Data.py:93-98)."""
        return [v2T([p[0], p[1], 0, 0, 0, p[2]]) for p in self.__trajectory_data]

    def get_world_data(self): return self.__world_data
    def get_measurements_data(self, index): return self.__measurements_data[index]
