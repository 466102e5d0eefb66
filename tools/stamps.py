"""Segment times of one iteration of the on-chip single-launch loop, from the diagnostic build with clock
stamps (make -C monocular-visual-odometry_amd/csrc stamps; run with PICP_LIB=ab/libpicp_hip_stamps.so)."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.synthetic import make_problem  # noqa: E402

pkg = importlib.import_module("monocular-visual-odometry_amd")
L = pkg._lib
ctx = pkg.get_context(0)
cam = L.camera_struct([[180, 0, 320], [0, 180, 240], [0, 0, 1]], 640, 480)
sched = L.schedule_struct(20, 10, 1000, 10, 2000, 64, 1, 64, disable_stop=True)
for M in [int(a) for a in sys.argv[1:]] or [1 << 20]:
    uv, pw, _ = make_problem(M, seed=[0, 0])
    corr = ctx.soa(uv, pw)
    for _ in range(3):
        res, hist = ctx.run(corr, np.eye(4), cam, sched)
    t = hist[2:, 74:80]                                   # cumulative microseconds at the six boundaries
    seg = np.diff(np.concatenate([np.zeros((len(t), 1)), t], axis=1), axis=1)
    names = ["stage 1", "workgroup sum 1", "exchange 1", "stage 2", "workgroup sum 2", "exchange 2"]
    med = np.median(seg, axis=0)
    print(f"M={M}: " + ", ".join(f"{n} {m:.2f}" for n, m in zip(names, med)) + f"  | to end of exchange 2: {np.median(t[:, 5]):.2f} us")
