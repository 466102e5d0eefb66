"""Repeats the single-launch loops many times on fixed inputs: every run must return bit-identical
results (fixed order of all sums) and no launch may give up at an exchange (idle device)."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

pkg = importlib.import_module("monocular-visual-odometry_amd")
L = pkg._lib
ctx = pkg.get_context(0)
cam = L.camera_struct([[180, 0, 320], [0, 180, 240], [0, 0, 1]], 640, 480)
sched = L.schedule_struct(20, 10, 1000, 10, 2000, 64, 1, 64, disable_stop=True)
cases = [(1 << 20, 20000), (70_001, 5000), (1 << 21, 3000), (3 << 20, 2000), (1 << 23, 500)]
if len(sys.argv) > 1:
    cases = [(int(a.split(":")[0]), int(a.split(":")[1])) for a in sys.argv[1:]]
for M, reps in cases:
    corr = bench.device_problem(ctx, M, 0, M)
    plan = ctx.plan(corr, cam, sched, want_history=True)
    res, hist0 = plan(np.eye(4))
    T0 = np.array(list(res.T))
    t0 = time.perf_counter()
    bad = 0
    for k in range(reps):
        res, hist = plan(np.eye(4))
        if not (np.array_equal(np.array(list(res.T)), T0) and (k % 50 or np.array_equal(hist, hist0))):
            bad += 1
        if k % 2000 == 1999:
            print(f"  M={M}: {k + 1} runs, {bad} differing, {ctx.resident_aborts()} abandoned launches", flush=True)
    dt = time.perf_counter() - t0
    print(f"M={M:9d}  {reps} runs in {dt:6.1f} s  differing results: {bad}  abandoned launches: {ctx.resident_aborts()}",
          flush=True)
    del corr, plan
