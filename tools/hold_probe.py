"""The streaming single-launch loop with and without part of its input held in LDS (PICP_PERSISTENT_HOLD),
at sizes beyond the on-chip capacity: time per iteration, and that counts, thresholds and poses agree."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

pkg = importlib.import_module("monocular-visual-odometry_amd")
L = pkg._lib
cam = L.camera_struct([[180, 0, 320], [0, 180, 240], [0, 0, 1]], 640, 480)
sched = L.schedule_struct(20, 10, 1000, 10, 2000, 64, 1, 64, disable_stop=True)
os.environ["PICP_PERSISTENT_HOLD"] = "0"
plain_ctx = L.Context(0)
del os.environ["PICP_PERSISTENT_HOLD"]
hold_ctx = L.Context(0)
sizes = [int(s) for s in sys.argv[1:]] or [(1 << 21) + 2, 3 << 20, 1 << 22, 1 << 23, 1 << 24, 1 << 26]
for M in sizes:
    corr = bench.device_problem(hold_ctx, M, 0, M)
    out = {}
    for name, ctx in (("streamed", plain_ctx), ("held", hold_ctx), ("streamed2", plain_ctx), ("held2", hold_ctx)):
        res, hist = ctx.run(corr, np.eye(4), cam, sched)
        ctx.synchronize()
        reps = 10 if M <= (1 << 24) else 4
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.run(corr, np.eye(4), cam, sched, want_history=False)
        ctx.synchronize()
        out[name] = ((time.perf_counter() - t0) / reps, res, hist)
    (ts, rs, hs), (to, ro, ho) = out["streamed"], out["held"]
    ts, to = min(ts, out["streamed2"][0]), min(to, out["held2"][0])
    same = all(np.array_equal(hs[:, c], ho[:, c]) for c in (L.H_NUM_INLIERS, L.H_N_INSIDE, L.H_N_STAGE1, L.H_N_STAGE2,
                                                               L.H_KERNEL_THR, L.H_DAMPING))
    dT = float(np.max(np.abs(hs[:, :16] - ho[:, :16])))
    print(f"M={M:9d}  all streamed {ts * 1e6 / 20:7.1f} us/it  part held in LDS {to * 1e6 / 20:7.1f} us/it  "
          f"({M * 20 / to / 1e9:5.1f} G corr-it/s, {100 * (1 - to / ts):4.1f} % less)  counts equal {same}  max |dT| {dT:.1e}  "
          f"aborts {hold_ctx.resident_aborts()}", flush=True)
    del corr
