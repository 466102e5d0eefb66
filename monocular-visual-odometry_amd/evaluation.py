"""Trajectory and map evaluation against ground truth: the arithmetic of the reference's
VisualOdometry.evaluate (src/VisualOdometry.py:97-157), kept free of any device dependency.
"""
from __future__ import annotations

import numpy as np

from .geometry_utils import Rt2T, transform


def trajectory_errors(estimated_trajectory, estimated_map, gt_trajectory, cam_transform, initial_frame=0):
    """Per-frame errors of an estimated camera trajectory.

    estimated_trajectory: list of 4x4 camera poses in the first camera's frame (monocular scale).
    estimated_map:        {'position': [...], 'appearance': [...]} in the same frame.
    gt_trajectory:        list of 4x4 robot poses (Data.get_trajectory_data_poses()).
    cam_transform:        camera pose in the robot frame (Camera.get_camera_transform()).

    Follows the reference step by step: move the estimate into the robot-world frame (:104-108);
    for consecutive pose pairs compare relative motions -- rotation error from the trace of the
    relative rotation rounded to 5 decimals (:124-125), Frobenius rotation ratio (:128), translation
    ratio |t_gt| / |t_est| (:129) and the error left after applying that ratio (:135); the global
    scale is the mean ratio (:154).  Returns a dict with the four per-frame lists, 'scale', and the
    scaled 'trajectory' / 'world_points' in the robot-world frame (:156-157).
    """
    C = np.asarray(cam_transform, dtype=np.float64)
    T = gt_trajectory[initial_frame] @ Rt2T(R=C[:3, :3], t=np.zeros(3)) if initial_frame > 0 else C
    est = transform(estimated_trajectory, T)
    pts = transform(estimated_map['position'], T, are_points=True)
    rot_err, rot_ratio, tr_err, tr_ratio = [], [], [], []
    for i in range(len(est) - 1):
        rel_est = np.linalg.inv(est[i]) @ est[i + 1]
        rel_gt = np.linalg.inv(gt_trajectory[i]) @ gt_trajectory[i + 1]
        error_T = np.linalg.inv(rel_est) @ rel_gt
        error_R = np.round(error_T[:3, :3], 5)                                              # :124
        with np.errstate(invalid="ignore", divide="ignore"):
            rot_err.append(np.arccos((np.trace(error_R) - 1) / 2))                          # :125
            rot_ratio.append(np.linalg.norm(est[i][:3, :3], 'fro') / np.linalg.norm(gt_trajectory[i][:3, :3], 'fro'))
            ratio = np.linalg.norm(gt_trajectory[i][:3, 3]) / np.linalg.norm(est[i][:3, 3])  # :129
            tr_ratio.append(ratio)
            tr_err.append(np.linalg.norm(ratio * est[i][:3, 3] - gt_trajectory[i][:3, 3]))   # :135
    scale = np.mean(tr_ratio) if tr_ratio else 1.0                                          # :154
    return {'rotation_error': rot_err, 'rotation_ratio': rot_ratio, 'translation_error': tr_err,
            'translation_ratio': tr_ratio, 'scale': scale,
            'trajectory': transform(est, scale=scale),
            'world_points': transform(pts, scale=scale, are_points=True),
            'world_appearance': estimated_map['appearance']}
