"""Frame-loop driver and evaluation with the reference's interface (src/VisualOdometry.py:10-224).
Plotting is not reproduced; the three result files are.
"""
from __future__ import annotations

import os
import time

import numpy as np

from .Camera import Camera
from .Data import Data
from .ProjectiveICP import ProjectiveICP
from .data_association import data_association_on_appearance
from .geometry_utils import Rt2T, transform


class VisualOdometry:
    def __init__(self, initial_frame=0,
                 final_frame=120,
                 verbose=False,
                 save_results=True,
                 save_icp_plots=False,
                 save_icp_plots_indices=[],
                 base_kernel_threshold=1500,
                 min_kernel_threshold=100,
                 max_kernel_threshold=3 * 1e3,
                 base_dumping_factor=1e4,
                 min_dumping_factor=1e3,
                 max_dumping_factor=1e9,
                 min_inliners=10,
                 num_iterations=300,
                 camera=None,
                 data=None):
        self.__initial_frame = initial_frame
        self.__final_frame = final_frame
        self.__save_results = save_results
        self.__num_of_frames = 0
        self.__camera = camera if camera is not None else Camera()
        self.__data = data if data is not None else Data()
        # positional, in the reference's order (src/VisualOdometry.py:54-67)
        self.ProjectiveICP = ProjectiveICP(self.__camera, self.__data, num_iterations, min_inliners,
                                           base_kernel_threshold, min_kernel_threshold, max_kernel_threshold,
                                           base_dumping_factor, min_dumping_factor, max_dumping_factor,
                                           verbose, save_results, save_icp_plots, save_icp_plots_indices)

    def run(self):
        """src/VisualOdometry.py:70-95.  Returns (total_time, mean_time_per_frame)."""
        start = time.time()
        frame_times = []
        for i in range(self.__initial_frame, self.__final_frame):
            t0 = time.time()
            res = self.ProjectiveICP.initialize(i) if i == self.__initial_frame else self.ProjectiveICP.update(i)
            frame_times.append(time.time() - t0)
            if not res:
                print(f'Not valid transform found at frame {i}. Break.')
                break
            self.__num_of_frames += 1
        total_time = time.time() - start
        mean_time_per_frame = np.mean(frame_times)
        print(f'Mean time per frame: {mean_time_per_frame} [s]')
        print(f'Total time:          {total_time} [s]\n')
        return total_time, mean_time_per_frame

    def compute_errors(self):
        """The numbers behind errors.dat  (src/VisualOdometry.py:97-157).

        Returns a dict with per-frame 'rotation_error', 'rotation_ratio', 'translation_error',
        'translation_ratio', the global 'scale', and the scaled 'trajectory' / 'world_points' in the
        robot-world frame."""
        gt_trajectory = self.ProjectiveICP.get_data().get_trajectory_data_poses()
        estimated_trajectory = self.ProjectiveICP.get_trajectory()
        estimated_world_points = self.ProjectiveICP.get_map()
        C = np.asarray(self.ProjectiveICP.get_camera().get_camera_transform(), dtype=np.float64)
        T = gt_trajectory[self.__initial_frame] @ Rt2T(R=C[:3, :3], t=np.zeros(3)) if self.__initial_frame > 0 else C
        est = transform(estimated_trajectory, T)
        pts = transform(estimated_world_points['position'], T, are_points=True)
        rot_err, rot_ratio, tr_err, tr_ratio = [], [], [], []
        for i in range(len(est) - 1):
            rel_est = np.linalg.inv(est[i]) @ est[i + 1]
            rel_gt = np.linalg.inv(gt_trajectory[i]) @ gt_trajectory[i + 1]
            error_T = np.linalg.inv(rel_est) @ rel_gt
            error_R = np.round(error_T[:3, :3], 5)                                   # :124
            with np.errstate(invalid="ignore"):
                rot_err.append(np.arccos((np.trace(error_R) - 1) / 2))               # :125
            rot_ratio.append(np.linalg.norm(est[i][:3, :3], 'fro') / np.linalg.norm(gt_trajectory[i][:3, :3], 'fro'))
            with np.errstate(divide="ignore", invalid="ignore"):
                ratio = np.linalg.norm(gt_trajectory[i][:3, 3]) / np.linalg.norm(est[i][:3, 3])   # :129
                tr_ratio.append(ratio)
                tr_err.append(np.linalg.norm(ratio * est[i][:3, 3] - gt_trajectory[i][:3, 3]))    # :135
        scale = np.mean(tr_ratio)                                                    # :154
        return {'rotation_error': rot_err, 'rotation_ratio': rot_ratio, 'translation_error': tr_err,
                'translation_ratio': tr_ratio, 'scale': scale,
                'trajectory': transform(est, scale=scale),
                'world_points': transform(pts, scale=scale, are_points=True),
                'world_appearance': estimated_world_points['appearance']}

    def evaluate(self, output_dir="outputs/final_results"):
        """Prints the summary and writes errors.dat, estimated_trajectory.dat and
        estimated_world_points.dat in the reference's formats  (src/VisualOdometry.py:158-205)."""
        r = self.compute_errors()
        gt_world_points = self.ProjectiveICP.get_data().get_world_data()
        matches = data_association_on_appearance({'position': r['world_points'], 'appearance': r['world_appearance']},
                                                 gt_world_points)
        a, b = np.array(matches['points_1']), np.array(matches['points_2'])
        rmse = np.sqrt(np.mean(np.linalg.norm(a - b, axis=1) ** 2)) if len(a) else float('nan')
        print(f'Number of frames:            {self.__num_of_frames}')
        print(f'Number of world points:      {len(a)}')
        print(f'RMSE world map [m]:          {np.round(rmse, 3)}\n')
        print(f'scale:                       {np.round(r["scale"], 3)}\n')
        for title, key in (('Rotation error [rad]', 'rotation_error'), ('Translation error', 'translation_error'),
                           ('Rotation ratio', 'rotation_ratio'), ('Translation ratio', 'translation_ratio')):
            v = np.asarray(r[key])
            print(f'Max {title}: {np.round(np.max(v), 3)}  Min: {np.round(np.min(v), 3)}  Mean: {np.round(np.mean(v), 3)}')
        os.makedirs(output_dir, exist_ok=True)
        with open(os.path.join(output_dir, "estimated_trajectory.dat"), "w") as fh:
            for i, pose in enumerate(r['trajectory']):
                x, y, z = pose[:3, 3]
                fh.write(f'{i} {x} {y} {z}\n')
        with open(os.path.join(output_dir, "estimated_world_points.dat"), "w") as fh:
            for i, point in enumerate(r['world_points']):
                x, y, z = point
                fh.write(f'{i} {x} {y} {z} ' + ' '.join(str(c) for c in r['world_appearance'][i]) + '\n')
        with open(os.path.join(output_dir, "errors.dat"), "w") as fh:
            for i in range(len(r['rotation_error'])):
                fh.write(f'{i} {r["rotation_error"][i]} {r["rotation_ratio"][i]} '
                         f'{r["translation_error"][i]} {r["translation_ratio"][i]}\n')
        return r
