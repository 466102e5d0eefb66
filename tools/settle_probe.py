"""How the launch duration of the on-chip loop develops from a cold start of the process: mean of every ten
launches (events bound to the dispatch), 2^20 correspondences x 20 iterations."""
import ctypes as C
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
M = 1 << 20
corr = bench.device_problem(ctx, M, 0, M)
state = ctx.empty(L.STATE_DOUBLES)
pose = np.ascontiguousarray(np.eye(4).reshape(16))
args = [ctx.handle, *[ctx.ptr(c) for c in corr], M, pose.ctypes.data_as(C.c_void_p), C.byref(cam), C.byref(sched), ctx.ptr(state)]
mean_ms, min_ms = C.c_double(), C.c_double()
idle = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
for phase in range(2):
    out = []
    t0 = time.perf_counter()
    for k in range(40):
        L.check(ctx.lib.picp_profile_run(*args, 10, C.byref(mean_ms), C.byref(min_ms)), "picp_profile_run")
        out.append(mean_ms.value * 1e3)
    print(f"phase {phase}: 400 launches in {(time.perf_counter() - t0) * 1e3:.0f} ms; mean of every ten, us:")
    print(" ".join(f"{v:.1f}" for v in out), flush=True)
    time.sleep(idle)
