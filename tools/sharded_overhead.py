"""One-GPU check of the sharded driver's host overhead: the stage-by-stage Python loop (as used with
one rank per GPU, here without a process group) against picp_run's own streaming loop."""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

pkg = importlib.import_module("monocular-visual-odometry_amd")
L = pkg._lib
ctx = pkg.get_context(0)
cam = L.camera_struct([[180, 0, 320], [0, 180, 240], [0, 0, 1]], 640, 480)
sched = L.schedule_struct(20, 10, 1000, 10, 2000, 64, 1, 64, disable_stop=True)
assert pkg.sharded.init_peer_exchange(ctx)          # one rank: the sharded single launch without peers
for M in [int(a) for a in sys.argv[1:]] or [1 << 21, 1 << 23, 1 << 25]:
    corr = bench.device_problem(ctx, M, 0, M)
    stages = pkg.sharded.HipStages(ctx, corr, np.eye(4), cam, sched, want_history=False)
    for name, limit, fn in (
            ("picp_run (single launch)", None, lambda: ctx.run(corr, np.eye(4), cam, sched, want_history=False)),
            ("picp_run_sharded, 1 rank", None, lambda: ctx.run_sharded(corr, np.eye(4), cam, sched, want_history=False)),
            ("picp_run (stage launches)", 0, lambda: ctx.run(corr, np.eye(4), cam, sched, want_history=False)),
            ("run_sharded rounds, 1 rank", 0, lambda: pkg.sharded.run_sharded(stages, 20, disable_stop=True))):
        ctx.set_resident_limit(limit)
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"M={M:9d} {name:28s} {dt * 1e3:8.3f} ms/loop  {dt / 20 * 1e6:7.1f} us/iteration  {M * 20 / dt / 1e9:6.1f} G corr-it/s",
              flush=True)
    ctx.set_resident_limit(None)
