"""Secondary measurements (not the driver's bench line): the batched configuration (4 096 problems x
2 048 correspondences, 20 iterations each) and the association configuration (10^6 x 10^6 descriptors).

    python bench_extra.py [--reps 5]
Prints one JSON object per configuration."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--per-problem", type=int, default=2048)
    ap.add_argument("--descriptors", type=int, default=1_000_000)
    args = ap.parse_args()
    import torch
    from tests.synthetic import make_problem
    pkg = importlib.import_module("monocular-visual-odometry_amd")
    L = pkg._lib
    ctx = pkg.get_context(0)
    cam = L.camera_struct([[180, 0, 320], [0, 180, 240], [0, 0, 1]], 640, 480)
    sched = L.schedule_struct(20, 10, 1000, 10, 2000, 64, 1, 64, True)

    B, N = args.batch, args.per_problem
    uv = np.empty((B * N, 2))
    pw = np.empty((B * N, 3))
    for p in range(B):
        uv[p * N:(p + 1) * N], pw[p * N:(p + 1) * N], _ = make_problem(N, seed=p)
    corr = ctx.soa(uv, pw)
    offsets = ctx.to_device(np.arange(B + 1, dtype=np.int64) * N, np.int64)
    poses0 = ctx.to_device(np.tile(np.eye(4).reshape(1, 16), (B, 1)))
    ctx.run_batched(corr, offsets, poses0, cam, sched, max_count=N)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        results, _ = ctx.run_batched(corr, offsets, poses0, cam, sched, max_count=N)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    print(json.dumps({"config": f"batched ICP, {B} problems x {N} correspondences, 20 LM iterations, one workgroup each",
                      "ms": dt * 1e3, "correspondence_iterations_per_s": B * N * 20 / dt,
                      "problems_per_s": B / dt,
                      "algorithmic_GBs_at_40B_per_correspondence_per_solve": 40.0 * B * N / dt / 1e9}))

    n = args.descriptors
    rng = np.random.default_rng(7)
    app2 = rng.uniform(-1.0, 1.0, (n, 10))
    perm = rng.permutation(n)
    d1, d2 = ctx.to_device(app2[perm]), ctx.to_device(app2)
    for mode, label in (("scan", "all pairs"), ("hash", "hash join")):
        ctx.set_association_mode(mode)
        ctx.associate(d1, d2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            i1, i2 = ctx.associate(d1, d2)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        assert np.array_equal(i2.cpu().numpy(), perm)
        print(json.dumps({"config": f"appearance association, {n} x {n} descriptors of 10 doubles, {label}",
                          "ms": dt * 1e3, "pair_evaluations_per_s": float(n) * n / dt}))
    ctx.set_association_mode("auto")

if __name__ == "__main__":
    main()
