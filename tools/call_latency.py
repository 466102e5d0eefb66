"""Wall time of a synchronous picp_run call (through Context.plan) at small and medium sizes: the in-kernel
iterations plus the host's launch and wait."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from tests.synthetic import make_problem
pkg = importlib.import_module("monocular-visual-odometry_amd")
L = pkg._lib
ctx = pkg.get_context(0)
cam = L.camera_struct([[180, 0, 320], [0, 180, 240], [0, 0, 1]], 640, 480)
for M, it in ((300, 30), (300, 250), (5000, 30), (70000, 30)):
    sched = L.schedule_struct(it, 10, 1000, 10, 2000, 64, 1, 64, disable_stop=True)
    uv, pw, _ = make_problem(M, seed=3)
    corr = ctx.soa(uv, pw)
    plan = ctx.plan(corr, cam, sched, want_history=False)
    for _ in range(50): plan(np.eye(4))
    t0 = time.perf_counter()
    n = 2000
    for _ in range(n): plan(np.eye(4))
    dt = (time.perf_counter() - t0) / n
    print(f"M={M} iterations={it}: {dt*1e6:.1f} us per picp_run call ({dt*1e6/it:.2f} us per iteration)")
