"""Runs the bundled 121-frame sequence end to end on the GPU, like the reference's main.py with the
shipped config/config.yaml values, and compares the result files with the reference's published
outputs (tests/golden/bundled_sequence.npz holds both the sequence and the published files).
The bootstrap restates cv2.findEssentialMat's deterministic RANSAC (two_view.find_essential_mat), so
the whole run is comparable with the published one number by number.

    python run_sequence.py [--workdir DIR] [--final-frame N]
"""
import argparse
import contextlib
import io
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def run(workdir, final_frame=120, quiet=True, published_bootstrap=False):
    """Returns (summary dict, evaluation dict, VisualOdometry)."""
    argv = ["--workdir", workdir, "--final-frame", str(final_frame)] + (["--quiet"] if quiet else [])
    argv += ["--published-bootstrap"] if published_bootstrap else []
    return main(argv)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--workdir", default=None, help="where data/ and outputs/ are written (default: a temporary directory)")
    ap.add_argument("--final-frame", type=int, default=120)
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--published-bootstrap", action="store_true",
                    help="replace the two-view pose estimate of the first frame pair by the translation recoverable "
                         "from the published trajectory (rotation = identity), to compare the rest of the pipeline")
    args = ap.parse_args(argv)
    cwd = os.getcwd()
    from tests.bundled import load, write_data_dir
    if args.workdir is None:
        import tempfile
        args.workdir = tempfile.mkdtemp(prefix="picp_sequence_")
    os.makedirs(args.workdir, exist_ok=True)
    write_data_dir(os.path.join(args.workdir, "data"))
    os.chdir(args.workdir)
    from mvo_amd.VisualOdometry import VisualOdometry
    if args.published_bootstrap:
        import mvo_amd.ProjectiveICP as picp_mod
        bb = load()
        scale = float(np.mean(bb["published_errors"][:, 4]))
        t1 = (np.linalg.inv(bb["cam_transform"]) @ np.append(bb["published_trajectory"][1, 1:4] / scale, 1.0))[:3]
        picp_mod.recover_pose = lambda E, p0, p1, K, mask=None: (np.eye(3), -t1.reshape(3, 1), np.ones(len(p0), bool))
    # config/config.yaml:1-22 of the reference
    vo = VisualOdometry(initial_frame=0, final_frame=args.final_frame, verbose=False, save_results=False,
                        save_icp_plots=False, save_icp_plots_indices=[], base_kernel_threshold=1000,
                        min_kernel_threshold=10, max_kernel_threshold=2000, base_dumping_factor=64,
                        min_dumping_factor=1, max_dumping_factor=64, min_inliners=10, num_iterations=250)
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink if args.quiet else sys.stdout):
        total, per_frame = vo.run()
        r = vo.evaluate()
    b = load()
    pub_err, pub_traj, pub_pts = b["published_errors"], b["published_trajectory"], b["published_world_points"]
    n = len(r["rotation_error"])
    rot = np.nan_to_num(np.array(r["rotation_error"]))
    summary = {
        "frames": n, "total_time_s": total, "mean_time_per_frame_s": float(per_frame),
        "map_points": len(r["world_points"]), "published_map_points": int(len(pub_pts)),
        "scale": float(r["scale"]), "published_scale": float(np.mean(pub_err[:, 4])),
        "max_abs_rotation_error_diff": float(np.max(np.abs(rot - pub_err[:n, 1]))),
        "max_abs_rotation_ratio_diff": float(np.max(np.abs(np.array(r["rotation_ratio"]) - pub_err[:n, 2]))),
        "max_abs_translation_error_diff": float(np.max(np.abs(np.array(r["translation_error"])[1:] - pub_err[1:n, 3]))),
        "max_abs_translation_ratio_diff": float(np.max(np.abs(np.array(r["translation_ratio"])[1:] - pub_err[1:n, 4]))),
        "max_trajectory_position_diff": float(np.max(np.abs(np.array([p[:3, 3] for p in r["trajectory"]])
                                                            - pub_traj[:n + 1, 1:4]))),
    }
    if len(r["world_points"]) == len(pub_pts):
        # the map in insertion order: same points (appearance) at the same places
        summary["map_appearances_identical"] = bool(np.array_equal(np.array(r["world_appearance"]), pub_pts[:, 4:]))
        summary["max_world_point_diff"] = float(np.max(np.abs(np.array(r["world_points"]) - pub_pts[:, 1:4])))
    os.chdir(cwd)
    if argv is None:
        print(json.dumps(summary, indent=1))
    return summary, r, vo


if __name__ == "__main__":
    main()
