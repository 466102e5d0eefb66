"""Skew between the workgroups of the on-chip single-launch loop: from the diagnostic build's per-workgroup
clock stamps of one iteration (PICP_LIB=ab/libpicp_hip_stamps.so python tools/skew.py 2> /dev/null).  The stamps are
the device's constant-rate clock (10 ns), comparable across workgroups."""
import importlib
import os
import re
import subprocess
import sys

import numpy as np

if os.environ.get("SKEW_CHILD") != "1":
    out = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=dict(os.environ, SKEW_CHILD="1"),
                         capture_output=True, text=True).stdout
    rows = [list(map(int, m.groups())) for m in re.finditer(r"wg (\d+): (\d+) (\d+) (\d+) (\d+) (\d+) (\d+) (\d+) (\d+)", out)]
    n = max(r[0] for r in rows) + 1
    last = np.array(rows[-n:], dtype=np.int64)           # the last recorded run
    last = last[np.argsort(last[:, 0])]
    t = (last[:, 1:] - last[:, 1:].min(axis=0)[0]) * 0.01   # us since the earliest iteration start
    names = ["iteration start", "stage 1 done", "sum 1 done", "exchange 1 done", "stage 2 done", "sum 2 done",
             "exchange 2 done", "solve done"]
    for k, name in enumerate(names):
        c = t[:, k]
        print(f"{name:18s} min {c.min():6.2f}  median {np.median(c):6.2f}  max {c.max():6.2f}  spread {c.max() - c.min():5.2f} us")
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.synthetic import make_problem  # noqa: E402

pkg = importlib.import_module("monocular-visual-odometry_amd")
L = pkg._lib
ctx = pkg.get_context(0)
cam = L.camera_struct([[180, 0, 320], [0, 180, 240], [0, 0, 1]], 640, 480)
sched = L.schedule_struct(20, 10, 1000, 10, 2000, 64, 1, 64, disable_stop=True)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
uv, pw, _ = make_problem(M, seed=[0, 0])
corr = ctx.soa(uv, pw)
for _ in range(3):
    ctx.run(corr, np.eye(4), cam, sched)
ctx.synchronize()
