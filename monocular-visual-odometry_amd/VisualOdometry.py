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
from .evaluation import trajectory_errors


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
        """The numbers behind errors.dat (see evaluation.trajectory_errors)."""
        picp = self.ProjectiveICP
        return trajectory_errors(picp.get_trajectory(), picp.get_map(), picp.get_data().get_trajectory_data_poses(),
                                 picp.get_camera().get_camera_transform(), self.__initial_frame)

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
