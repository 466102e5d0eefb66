"""Projective ICP with the reference's class interface (src/ProjectiveICP.py:8-381); the pose
loop, its linearisation and the appearance association run as HIP kernels on an MI355X.

Constructor arguments, method names, return values and the side effects on the camera object
are the reference's, so a driver written against the reference class works unchanged.  The
reference's private methods (``__projective_ICP``, ``__one_step``, ``__linearize``) are public
here as ``projective_icp``, ``one_step`` and ``linearize``.
"""
from __future__ import annotations

import os

import numpy as np

from . import _lib
from ._lib import camera_struct, get_context, schedule_struct
from .data_association import data_association_on_appearance, match_indices
from .two_view import find_essential_mat, recover_pose, triangulate_points
from .geometry_utils import Rt2T


def summary_lines(results, is_valid, lo, hi):
    """The per-frame summary the reference prints (src/ProjectiveICP.py:122-138), line for line:
    labels padded to a 38-character column, the error rounded to 5 places, the other means to
    whole numbers, the best and worst iteration's values as they are."""
    def row(label, value, tail=''):
        return f'  - {label}:'.ljust(38) + f'{value}{tail}'
    out = [row('Valid transformation', is_valid), row('Num iterations', len(results['T']), '\n')]
    for title, key in (('Error', 'error'), ('Num inliers', 'num_inliers'),
                       ('Kernel threshold', 'kernel_threshold'), ('Dumping factor', 'dumping_factor')):
        vals = results[key]
        if key == 'error':
            out.append(row(f'{title} best iteration', np.round(vals[lo], 5), f' (index: {lo})'))
            out.append(row(f'{title} worst iteration', np.round(vals[hi], 5), f' (index: {hi})'))
            out.append(row(f'Mean {title.lower()}', np.round(np.mean(vals), 5), '\n'))
        else:
            out.append(row(f'{title} best iteration', vals[lo]))
            out.append(row(f'{title} worst iteration', vals[hi]))
            out.append(row(f'Mean {title.lower()}', np.round(np.mean(vals)), '\n'))
    return out


def results_from_history(hist, base_kernel_threshold, min_kernel_threshold, base_dumping_factor):
    """Device history rows -> the reference's iterations_results dict (src/ProjectiveICP.py:173).

    The reference's threshold and damping values are Python numbers whose type follows its
    arithmetic: the threshold keeps the type it has under ``+= 10`` / ``-= 10`` and takes the type
    of ``min_kernel_threshold`` when every pair is an inlier (:200-205); the damping factor stays
    what the caller passed until the first ``/= 2`` makes it a float for good, while ``*= 2`` keeps
    an int an int (:224-225; a halving and a doubling never fall into one iteration, since the
    former needs ``stuck > 10`` or both counters at zero and the latter ``flicker > 10``, and each
    counter is reset whenever the other one's condition holds).  Integer parameters (the YAML's)
    therefore give ``990`` and ``32.0``, not ``990.0`` and ``32``; the lists here carry the same
    types."""
    def is_int(x):
        return isinstance(x, (int, np.integer)) and not isinstance(x, bool)
    thr_int, min_int, damp_int = is_int(base_kernel_threshold), is_int(min_kernel_threshold), is_int(base_dumping_factor)
    prev_damp = float(base_dumping_factor)
    thresholds, dampings = [], []
    for k in range(hist.shape[0]):
        if hist[k, _lib.H_NUM_INLIERS] == hist[k, _lib.H_N_INSIDE]:      # :200-201
            thr_int = min_int
        t = float(hist[k, _lib.H_KERNEL_THR])
        thresholds.append(int(t) if thr_int and t.is_integer() else t)
        d = float(hist[k, _lib.H_DAMPING])
        if d < prev_damp:                                                # a halving: true division
            damp_int = False
        prev_damp = d
        dampings.append(int(d) if damp_int and d.is_integer() else d)
    return {'T': [hist[k, _lib.H_T:_lib.H_T + 16].reshape(4, 4).copy() for k in range(hist.shape[0])],
            'error': [float(e) for e in hist[:, _lib.H_ERROR]],
            'num_inliers': [int(n) for n in hist[:, _lib.H_NUM_INLIERS]],
            'kernel_threshold': thresholds,
            'dumping_factor': dampings}


class ProjectiveICP:
    def __init__(self, camera,
                 data,
                 num_iterations=300,
                 min_inliers=10,
                 kernel_threshold=1500,
                 min_kernel_threshold=100,
                 max_kernel_threshold=3 * 1e3,
                 dumping_factor=1e4,
                 min_dumping_factor=1e3,
                 max_dumping_factor=1e9,
                 verbose=False,
                 save_results=True,
                 save_icp_plots=False,
                 save_icp_plots_indices=[],
                 device=0):
        self.__data = data
        self.__camera = camera
        self.__num_iterations = num_iterations
        self.__min_inliers = min_inliers
        self.__base_kernel_threshold = kernel_threshold
        self.__min_kernel_threshold = min_kernel_threshold
        self.__max_kernel_threshold = max_kernel_threshold
        self.__base_dumping_factor = dumping_factor
        self.__min_dumping_factor = min_dumping_factor
        self.__max_dumping_factor = max_dumping_factor
        self.__verbose = verbose
        self.__save_results = save_results
        self.__save_icp_plots = save_icp_plots
        self.__save_icp_plots_indices = save_icp_plots_indices
        self.__trajectory = []                                          # list of 4x4 poses w_T_c
        self.__map = {'position': [], 'appearance': [], 'error': []}    # 3-D points in the first camera's frame
        self.__current_pose = np.eye(4)
        self.__ctx = get_context(device)                                # raises without the HIP library / a GPU
        self.last_history = None                                        # raw device history of the last loop

    # ---- helpers ---------------------------------------------------------------------------------
    def __camera_struct(self):
        W, H = self.__camera.get_camera_resolution()
        return camera_struct(self.__camera.get_camera_matrix(), W, H)

    def __schedule(self, num_iterations=None, kernel_threshold=None, dumping_factor=None):
        return schedule_struct(
            self.__num_iterations if num_iterations is None else num_iterations, self.__min_inliers,
            self.__base_kernel_threshold if kernel_threshold is None else kernel_threshold,
            self.__min_kernel_threshold, self.__max_kernel_threshold,
            self.__base_dumping_factor if dumping_factor is None else dumping_factor,
            self.__min_dumping_factor, self.__max_dumping_factor)

    def __run_loop(self, corr, w_T_c0, sched):
        """All iterations on the device.  Returns (result struct, history ndarray [n_it, HIST_STRIDE])."""
        ctx = self.__ctx
        cam = self.__camera_struct()
        return ctx.run(corr, w_T_c0, cam, sched, want_history=True)

    # ---- the reference's public methods ------------------------------------------------------------
    def initialize(self, frame_index):
        """Two-frame bootstrap (src/ProjectiveICP.py:54-86): relative pose from the essential matrix
        of the appearance-matched points, triangulated map, trajectory of two poses."""
        measurement_0 = self.__data.get_measurements_data(frame_index)
        measurement_1 = self.__data.get_measurements_data(frame_index + 1)
        matches = data_association_on_appearance(measurement_0, measurement_1)
        points_0 = np.array(matches['points_1'])
        points_1 = np.array(matches['points_2'])
        appearances = np.array(matches['appearance'])

        w_T_c0 = np.eye(4)
        self.__update_state(w_T_c0, {'position': [], 'appearance': []}, 0.0)

        K = self.__camera.get_camera_matrix()
        E, mask = find_essential_mat(points_0, points_1, K)              # :68
        R, t, mask = recover_pose(E, points_0, points_1, K, mask)       # :69
        c0_T_c1 = Rt2T(R, -t)                                           # :70 (note the sign)
        w_T_c1 = w_T_c0 @ c0_T_c1

        points_3D, mask = triangulate_points(points_0, points_1, w_T_c0, w_T_c1, K)
        map = {'position': points_3D, 'appearance': appearances[mask].tolist()}
        self.__update_state(w_T_c1, map, 0.0)

        print(f'Frame: {frame_index}')
        print(f'Transformation of frame {frame_index}: w_T_c0 - set to identity.')
        print(f'Relative transformation between {frame_index} and {frame_index+1}: c0_T_c1 - estimated using 2D-2D correspondences.')
        print(f'Transformation of frame {frame_index+1}: w_T_c1 = w_T_c0 c0_T_c1')
        print('========================================================================================\n')
        return True

    def update(self, frame_index):
        """One frame (src/ProjectiveICP.py:89-151): ICP of the next measurement against the map,
        then triangulation of the frame-to-frame matches and map growth."""
        os.makedirs(f'outputs/frame_{frame_index:02d}', exist_ok=True)          # unconditionally, as the reference (:90)
        current_measurement = self.__data.get_measurements_data(frame_index)
        next_measurement = self.__data.get_measurements_data(frame_index + 1)

        w_T_c1, is_valid, iterations_results = self.projective_icp(next_measurement, frame_index)

        if is_valid:
            matches = data_association_on_appearance(current_measurement, next_measurement)
            points_0 = np.array(matches['points_1'])
            points_1 = np.array(matches['points_2'])
            appearances = np.array(matches['appearance'])
            points_3D, mask = triangulate_points(points_0, points_1, self.get_current_pose(), w_T_c1,
                                                 self.__camera.get_camera_matrix())

        err = iterations_results['error']
        lo, hi = int(np.argmin(err)), int(np.argmax(err))
        print(f'Frame: {frame_index}\n')
        print(f'Transformation of frame {frame_index}: w_T_c{frame_index} - previous pose')
        print(f'Relative transformation between {frame_index} and {frame_index+1}: c{frame_index}_T_c{frame_index+1} - estimated using PICP')
        print(f'Transformation of frame {frame_index+1}: w_T_c{frame_index+1} = w_T_c{frame_index} c{frame_index}_T_c{frame_index+1}\n')
        for line in self.summary_lines(iterations_results, is_valid, lo, hi):
            print(line)
        print(f'Applied transformation of index {lo} to the camera')
        print('========================================================================================\n')

        if is_valid:
            map = {'position': points_3D, 'appearance': appearances[mask].tolist()}
            self.__update_state(w_T_c1, map, np.mean(iterations_results['error']))
            return True
        return False

    # ---- the loop ----------------------------------------------------------------------------------
    def projective_icp(self, image_points, frame_index):
        """The LM loop of one frame (src/ProjectiveICP.py:154-257).

        Returns ``(T, is_valid, iterations_results)`` with the reference's five history lists.
        The descriptor matching of the per-iteration association call (:182) does not depend on the
        pose, so it is done once; its pose-dependent part (dropping pairs whose map point leaves the
        image) is re-evaluated inside every iteration on the device."""
        ctx = self.__ctx
        w_T_c0 = self.get_current_pose()
        world = self.get_map()
        idx1, idx2 = match_indices(image_points['appearance'], world['appearance'], ctx)
        uv = np.asarray(image_points['position'], dtype=np.float64).reshape(-1, 2)[idx1]
        pw = (np.asarray(world['position'], dtype=np.float64).reshape(-1, 3)[idx2]
              if len(idx2) else np.zeros((0, 3)))
        corr = ctx.soa(uv, pw)
        res, hist = self.__run_loop(corr, w_T_c0, self.__schedule())
        self.last_history = hist
        iterations_results = self.history_to_results(hist)
        if self.__verbose or frame_index in self.__save_icp_plots_indices:
            for k in range(hist.shape[0]):
                print(f'Frame: {frame_index}, Iteration: {k + 1}')
                print(f'  - Error:            {np.round(hist[k, _lib.H_ERROR], 5)}')
                print(f'  - Num inliers:      {int(hist[k, _lib.H_NUM_INLIERS])}')
                print(f'  - Kernel threshold: {hist[k, _lib.H_KERNEL_THR]}')
                print(f'  - Dumping factor:   {np.round(hist[k, _lib.H_DAMPING], 5)}')
                print('------------------------------------------------------------\n')
        if not res.valid:                                               # :249-251
            self.__camera.set_c_T_w(np.linalg.inv(self.get_current_pose()))
            return self.get_current_pose(), False, iterations_results
        T = iterations_results['T'][res.best_index]                     # :254-257
        self.__camera.set_c_T_w(np.linalg.inv(T))
        return T, True, iterations_results

    summary_lines = staticmethod(lambda results, is_valid, lo, hi: summary_lines(results, is_valid, lo, hi))

    def history_to_results(self, hist):
        """Device history rows -> the reference's iterations_results dict (src/ProjectiveICP.py:173)."""
        return results_from_history(hist, self.__base_kernel_threshold, self.__min_kernel_threshold,
                                    self.__base_dumping_factor)

    def one_step(self, reference_image_points, current_world_points, w_T_c0, kernel_threshold, dumping_factor):
        """One damped Gauss-Newton step (src/ProjectiveICP.py:260-273).  The camera state used for
        the projection is inv(w_T_c0), which is what the reference's loop guarantees (:232)."""
        uv = np.asarray(reference_image_points, dtype=np.float64).reshape(-1, 2)
        if len(current_world_points) == 0:                              # :263
            return {'T': w_T_c0, 'num_inliers': 0, 'error': np.inf, 'computation_done': False}
        corr = self.__ctx.soa(uv, current_world_points)
        sched = self.__schedule(num_iterations=1, kernel_threshold=kernel_threshold, dumping_factor=dumping_factor)
        res, hist = self.__ctx.run(corr, w_T_c0, self.__camera_struct(), sched, want_history=True)
        row = hist[0]
        if row[_lib.H_DONE] == 0.0:      # no pair projects inside the image: the step leaves the pose unchanged
            return {'T': w_T_c0, 'num_inliers': 0, 'error': np.inf, 'computation_done': True}
        return {'T': row[_lib.H_T:_lib.H_T + 16].reshape(4, 4).copy(), 'num_inliers': int(row[_lib.H_NUM_INLIERS]),
                'error': float(row[_lib.H_ERROR]), 'computation_done': True}

    def linearize(self, reference_image_points, currrent_world_points, kernel_threshold, w_T_c0):
        """(H, b, num_inliers, error) of the two-stage inlier selection (src/ProjectiveICP.py:276-323)."""
        corr = self.__ctx.soa(reference_image_points, currrent_world_points)
        H, b, n, err, _ = self.__ctx.linearize(corr, w_T_c0, self.__camera_struct(), kernel_threshold,
                                               self.__min_inliers)
        return H, b, n, err

    # reference-style private names, so code that reached into the reference class keeps working
    _ProjectiveICP__projective_ICP = projective_icp
    _ProjectiveICP__one_step = one_step
    _ProjectiveICP__linearize = linearize

    # ---- state -------------------------------------------------------------------------------------
    def __update_state(self, pose, map, error):
        """src/ProjectiveICP.py:356-374: append the pose, add map points whose descriptor is new."""
        self.__trajectory.append(pose)
        known = self.__map['appearance']
        for position, appearance in zip(map['position'], map['appearance']):
            if appearance not in known:
                self.__map['position'].append(position)
                self.__map['appearance'].append(appearance)
                self.__map['error'].append(error)
        self.__current_pose = pose
        self.__camera.set_c_T_w(np.linalg.inv(pose))

    def get_current_pose(self): return self.__current_pose
    def get_trajectory(self): return self.__trajectory
    def get_map(self): return self.__map
    def get_data(self): return self.__data
    def get_camera(self): return self.__camera
