"""Re-creates the bundled sequence's text files (camera.dat, trajectory.dat, world.dat,
meas-%05d.dat) from tests/golden/bundled_sequence.npz, in the formats the readers expect.
Floats are written with repr(), so parsing them gives back the stored doubles exactly."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load():
    return np.load(os.path.join(GOLDEN, "bundled_sequence.npz"), allow_pickle=False)


def _row(vals):
    return " ".join(repr(float(v)) for v in vals)


def write_data_dir(directory, frames=121):
    npz = load()
    b = {k: npz[k] for k in npz.files}      # NpzFile decompresses on every [] access
    os.makedirs(directory, exist_ok=True)
    K, C = b["camera_matrix"], b["cam_transform"]
    with open(os.path.join(directory, "camera.dat"), "w") as fh:
        fh.write("camera matrix:\n")
        for r in K:
            fh.write(" ".join(str(int(x)) for x in r) + "\n")
        fh.write("cam_transform:\n")
        for r in C:
            fh.write(_row(r) + "\n")
        fh.write(f"z_near: {repr(float(b['z_range'][0]))}\nz_far:  {repr(float(b['z_range'][1]))}\n")
        fh.write(f"width:  {int(b['resolution'][0])}\nheight: {int(b['resolution'][1])}\n")
    with open(os.path.join(directory, "trajectory.dat"), "w") as fh:
        for r in b["trajectory"]:
            fh.write(f"{int(r[0])} {_row(r[1:])}\n")
    with open(os.path.join(directory, "world.dat"), "w") as fh:
        for r in b["world"]:
            fh.write(f"{int(r[0])} {_row(r[1:])}\n")
    offs = b["meas_offsets"]
    for f in range(frames):
        with open(os.path.join(directory, f"meas-{f:05d}.dat"), "w") as fh:
            fh.write(f"seq: {f}\ngt_pose: {_row(b['meas_gt_pose'][f])}\nodom_pose: {_row(b['meas_odom_pose'][f])}\n")
            for k in range(offs[f], offs[f + 1]):
                fh.write(f"point {int(b['meas_cur_id'][k])} {int(b['meas_actual_id'][k])} "
                         f"{_row(b['meas_uv'][k])} {_row(b['meas_app'][k])}\n")
    return directory
