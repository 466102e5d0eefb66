"""Benchmark of the projective-ICP pose loop on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one complete pose loop: 20 LM iterations (stop test disabled) over a synthetic
correspondence set that is already resident in HBM.  N = 1 runs the single-GPU configuration
(2^20 correspondences); N > 1 shards 2^26 correspondences over the N GPUs with two small
all-reduces per iteration (RCCL, enqueued by the library on the stream of its kernels; the id for
its communicator travels through the torch.distributed process group).  Rank 0 prints one JSON line.

metric = correspondence-iterations per second, whole job.
roofline = the dominant kernel against the 8 TB/s HBM3E peak, algorithmic bytes over its mean
launch duration (HIP events bound to the dispatch):
  * N = 1, 2^20 correspondences: the single-launch loop kernel (correspondences held in registers
    for all 20 iterations).  Algorithmic bytes follow the reference's data flow, two passes of 40 B
    per correspondence per iteration = 80 B x M x 20 per launch, although the kernel reads the
    arrays from HBM only once ("hbm_bytes_read_once").
  * sharded runs (inputs too large for the register files): the stage-2 scan (linearise + H/b
    reduction), 40 B per correspondence per launch.
  "streaming_scan" reports that stage-2 figure for the N = 1 workload as well, and
  "streaming_scan_hbm_resident" for an input larger than the 256 MB Infinity Cache.
cpu_baseline = the C restatement of the reference loop (oracle/) on one host core, on a bounded sample.
"""
from __future__ import annotations

import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E nominal peak
BYTES_PER_CORR_PER_SCAN = 40   # u, v, x, y, z as float64
ITERATIONS = 20


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--corr", type=int, default=0, help="total correspondences (default 2^20 at 1 GPU, 2^26 sharded)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large-scan", action="store_true", help="skip the 2^24-correspondence stage-2 timing")
    ap.add_argument("--cpu-iterations", type=int, default=300)
    return ap.parse_args()


def device_problem(ctx, M, lo, hi, seed=0, chunk=1 << 22):
    """Correspondences [lo, hi) of the seeded M-point problem as five device arrays, generated in
    chunks so the host never holds more than one chunk.  Chunk c uses seed (seed, c), so any
    sharding of the same M yields the same global data."""
    from tests.synthetic import make_problem
    from importlib import import_module
    n = hi - lo
    stride = import_module("monocular-visual-odometry_amd")._lib._pad32(n)
    planes = ctx.zeros((5, stride))
    c0 = lo // chunk
    c1 = (hi + chunk - 1) // chunk
    for c in range(c0, c1):
        beg, end = c * chunk, min((c + 1) * chunk, M)
        uv, pw, _ = make_problem(end - beg, seed=[seed, c])
        a, b = max(lo, beg), min(hi, end)
        if a >= b:
            continue
        block = np.concatenate([uv[a - beg:b - beg].T, pw[a - beg:b - beg].T], axis=0)
        planes[:, a - lo:b - lo] = ctx.to_device(np.ascontiguousarray(block))
    return [planes[k][:n] for k in range(5)]


def time_stage2(ctx, pkg, corr, cam, sched, reps=40):
    """Mean and minimum duration of one stage-2 launch, from HIP events bound to each dispatch on
    the launch stream (picp_profile_stage2: hipExtLaunchKernelGGL start/stop events, i.e. the
    kernel's own begin/end timestamps, the interval a rocprofv3 kernel trace reports)."""
    L = pkg._lib
    lib, h = ctx.lib, ctx.handle
    M = int(corr[0].numel())
    state = ctx.empty(L.STATE_DOUBLES)
    pose = np.ascontiguousarray(np.eye(4).reshape(16))
    ptrs = [ctx.ptr(c) for c in corr]
    L.check(lib.picp_state_init(h, ctx.ptr(state), pose.ctypes.data_as(C.c_void_p), C.byref(sched)), "state_init")
    L.check(lib.picp_stage1(h, *ptrs, M, ctx.ptr(state), C.byref(cam)), "stage1")
    mean_ms, min_ms = C.c_double(), C.c_double()
    L.check(lib.picp_profile_stage2(h, *ptrs, M, ctx.ptr(state), C.byref(cam), 5, C.byref(mean_ms), C.byref(min_ms)),
            "warm")
    L.check(lib.picp_profile_stage2(h, *ptrs, M, ctx.ptr(state), C.byref(cam), reps, C.byref(mean_ms),
                                    C.byref(min_ms)), "picp_profile_stage2")
    return float(mean_ms.value), float(min_ms.value)


def time_resident(ctx, pkg, corr, cam, sched, reps=20):
    """Mean and minimum duration of the single-launch loop kernel (picp_profile_run: events bound
    to the dispatch, the interval a rocprofv3 kernel trace reports)."""
    L = pkg._lib
    M = int(corr[0].numel())
    state = ctx.empty(L.STATE_DOUBLES)
    pose = np.ascontiguousarray(np.eye(4).reshape(16))
    args = [ctx.handle, *[ctx.ptr(c) for c in corr], M, pose.ctypes.data_as(C.c_void_p), C.byref(cam), C.byref(sched),
            ctx.ptr(state)]
    mean_ms, min_ms = C.c_double(), C.c_double()
    L.check(ctx.lib.picp_profile_run(*args, 3, C.byref(mean_ms), C.byref(min_ms)), "warm")
    L.check(ctx.lib.picp_profile_run(*args, reps, C.byref(mean_ms), C.byref(min_ms)), "picp_profile_run")
    return float(mean_ms.value), float(min_ms.value)


def pmc_traffic(M, world, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 counter passes (FETCH_SIZE and
    WRITE_SIZE, separate runs of this same command; profiles/summarize.py applies the gfx950
    corrections).  Counters cannot be read from inside the process, so the figure is only reported
    for the workload it was collected on."""
    import glob
    if world != 1 or M != (1 << 20):
        return None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_2p20*.json")), reverse=True):
        with open(path) as fh:
            k = json.load(fh)["kernels"]
        for name, v in k.items():
            if kernel in name:
                return v["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
    return None, None


def cpu_baseline(M, iterations):
    """The C restatement of the reference loop (oracle/picp_oracle_c.c: one correspondence at a time
    in index order, as the reference works) on one host core, on the first `iterations` iterations of
    the same M-correspondence problem."""
    from oracle import picp_oracle as po
    from oracle import picp_oracle_c as oc
    from tests.synthetic import make_problem
    M = min(M, 1 << 20)
    uv, pw, _ = make_problem(M, seed=[0, 0])
    params = po.IcpParams(num_iterations=iterations, disable_stop=True)
    oc.load()
    t0 = time.perf_counter()
    oc.projective_icp(uv, pw, np.eye(4), params, want_history=False)
    dt = time.perf_counter() - t0
    return {"value": M * iterations / dt, "unit": "correspondence-iterations/s", "cores": 1, "kind": "port",
            "sample": f"C restatement of the reference loop (gcc -O2), {M} correspondences x {iterations} LM "
                      f"iterations, {dt:.1f} s on 1 of {os.cpu_count()} host CPUs"}


def main():
    args = parse()
    if os.environ.get("PICP_BENCH_DEBUG"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["PICP_BENCH_DEBUG"]), exit=True)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # Rehearsal knobs for a box with a single GPU (the real multi-GPU run uses neither): all ranks on
    # device 0 and gloo for the two small all-reduces, since RCCL refuses two ranks on one device.
    backend = os.environ.get("PICP_BENCH_BACKEND", "nccl")
    if os.environ.get("PICP_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    collective = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL over xGMI
            collective = "rccl"
        else:
            dist.init_process_group(backend)
            collective = backend
    from tests.synthetic import true_pose
    pkg = importlib.import_module("monocular-visual-odometry_amd")
    L = pkg._lib
    ctx = pkg.get_context(local_rank)

    if world > 1 and backend == "nccl" and os.environ.get("PICP_BENCH_NATIVE", "1") != "0":
        # the library's own RCCL communicator: all-reduces on the kernels' stream, whole rounds per call
        if pkg.sharded.init_native_exchange(ctx):
            collective = "rccl, enqueued by the library on the kernels' stream"
        else:
            collective = "rccl through torch.distributed (the library's communicator could not be set up)"

    M = args.corr or ((1 << 20) if world == 1 else (1 << 26))
    lo, hi = pkg.sharded.shard_bounds(M, rank, world)
    corr = device_problem(ctx, M, lo, hi)
    cam = L.camera_struct([[180, 0, 320], [0, 180, 240], [0, 0, 1]], 640, 480)
    sched = L.schedule_struct(ITERATIONS, 10, 1000, 10, 2000, 64, 1, 64, disable_stop=True)
    pose0 = np.eye(4)

    plan = ctx.plan(corr, cam, sched) if world == 1 else None     # picp_run with its arguments marshalled once

    def step():
        if world == 1:
            res, _ = plan(pose0)
        else:
            stages = pkg.sharded.HipStages(ctx, corr, pose0, cam, sched, want_history=False)
            res, _ = pkg.sharded.run_sharded(stages, ITERATIONS, disable_stop=True)
        return res

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1 and int(ctx.lib.picp_comm_world(ctx.handle)) == world:
        # one guarded loop through the library's own exchange: if it fails on any rank, every rank
        # drops its communicator and the run goes through torch.distributed's all-reduce instead
        try:
            step()
            torch.cuda.synchronize()
            ok = 1
        except Exception as e:                                       # noqa: BLE001 - reported, then the fallback runs
            print(f"rank {rank}: library exchange failed ({e}); falling back to torch.distributed", file=sys.stderr)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            ctx.lib.picp_comm_destroy(ctx.handle)
            collective = "rccl through torch.distributed (the library's exchange failed in its trial loop)"
    def run_steps(n):
        """n steps.  On one GPU the loops are independent of each other, so two are kept in flight
        (picp_run_enqueue / picp_run_collect): the host enqueues the next loop while the current one
        runs instead of waiting in between.  Every loop's result is collected and checked."""
        last = None
        if world > 1 or n == 0 or os.environ.get("PICP_BENCH_PIPELINE", "1") == "0":
            for _ in range(n):
                last = step()
            return last
        depth = 2
        for k in range(n):
            plan.enqueue(pose0, k % depth)
            if k >= depth - 1:
                last = plan.collect((k - depth + 1) % depth)
                assert last.num_iterations == ITERATIONS, last.num_iterations
        for k in range(max(n - depth + 1, 0), n):
            last = plan.collect(k % depth)
            assert last.num_iterations == ITERATIONS, last.num_iterations
        return last

    res = run_steps(args.warmup) or step()
    fence()
    t0 = time.perf_counter()
    res = run_steps(args.steps) or res
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert res.num_iterations == ITERATIONS, res.num_iterations

    n_local = hi - lo
    s2_mean_ms, s2_min_ms = time_stage2(ctx, pkg, corr, cam, sched)
    resident = world == 1 and 0 < M <= ctx.resident_capacity()
    if world == 1:
        run_mean_ms, run_min_ms = time_resident(ctx, pkg, corr, cam, sched)
    big = None
    if rank == 0 and world == 1 and not args.no_large_scan:
        # the same scan on an input that cannot live in the 256 MB Infinity Cache
        Mb = 1 << 24
        corr_b = device_problem(ctx, Mb, 0, Mb, seed=1)
        b_mean, b_min = time_stage2(ctx, pkg, corr_b, cam, sched, reps=20)
        big = {"correspondences": Mb, "bytes_per_launch": BYTES_PER_CORR_PER_SCAN * Mb, "mean_launch_ms": b_mean,
               "min_launch_ms": b_min, "achieved": BYTES_PER_CORR_PER_SCAN * Mb / (b_mean * 1e-3) / 1e9,
               "frac": BYTES_PER_CORR_PER_SCAN * Mb / (b_mean * 1e-3) / 1e9 / HBM_PEAK_GBS}
        del corr_b
    if rank == 0:
        scan = {"kernel": "stage2_kernel (linearise + H/b reduction)",
                "bytes_per_launch": BYTES_PER_CORR_PER_SCAN * n_local,
                "mean_launch_ms": s2_mean_ms, "min_launch_ms": s2_min_ms,
                "achieved": BYTES_PER_CORR_PER_SCAN * n_local / (s2_mean_ms * 1e-3) / 1e9}
        scan["frac"] = scan["achieved"] / HBM_PEAK_GBS
        if resident:
            kernel = "icp_resident_kernel"
            bytes_per_launch = 2 * BYTES_PER_CORR_PER_SCAN * n_local * ITERATIONS
            achieved = bytes_per_launch / (run_mean_ms * 1e-3) / 1e9
            roof = {"kernel": "icp_resident_kernel (whole loop in one launch, correspondences in registers)",
                    "bytes_per_launch": bytes_per_launch, "mean_launch_ms": run_mean_ms, "min_launch_ms": run_min_ms,
                    "hbm_bytes_read_once": BYTES_PER_CORR_PER_SCAN * n_local,
                    "note": "algorithmic bytes = 80 B per correspondence-iteration (two 40 B passes, the "
                            "reference's data flow); the kernel itself reads each array once per launch"}
        elif world == 1:
            kernel = "icp_persistent_kernel"
            bytes_per_launch = 2 * BYTES_PER_CORR_PER_SCAN * n_local * ITERATIONS
            achieved = bytes_per_launch / (run_mean_ms * 1e-3) / 1e9
            roof = {"kernel": "icp_persistent_kernel (whole loop in one launch, arrays streamed twice per iteration)",
                    "bytes_per_launch": bytes_per_launch, "mean_launch_ms": run_mean_ms, "min_launch_ms": run_min_ms}
        else:
            kernel = "stage2_kernel"
            achieved = scan["achieved"]
            roof = {k: scan[k] for k in ("kernel", "bytes_per_launch", "mean_launch_ms", "min_launch_ms")}
        traffic, traffic_source = pmc_traffic(M, world, kernel)
        out = {
            "metric": "ICP correspondence-iterations/sec",
            "value": M * ITERATIONS * args.steps / dt,
            "unit": "correspondence-iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("synthetic single-pose projective ICP, 2^20 correspondences, 20 LM iterations"
                                    if world == 1 and M == (1 << 20) else
                                    f"synthetic single-pose projective ICP, {M} correspondences sharded over "
                                    f"{world} GPU(s), 20 LM iterations, 2 all-reduces per iteration"),
                       "correspondences": M, "lm_iterations": ITERATIONS,
                       "steps_in_flight": 2 if (world == 1 and os.environ.get("PICP_BENCH_PIPELINE", "1") != "0") else 1,
                       "device_path": ("single launch, register-resident" if resident else
                                       "single launch, arrays streamed twice per iteration" if world == 1 else
                                       "streaming stage launches"),
                       "collective": collective,
                       "final_pose_error": float(np.max(np.abs(np.array(list(res.T)).reshape(4, 4) - true_pose())))},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         **roof,
                         "loop_bytes_per_corr_iteration": 2 * BYTES_PER_CORR_PER_SCAN,
                         "loop_achieved_GBs": 2 * BYTES_PER_CORR_PER_SCAN * n_local * ITERATIONS * args.steps / dt / 1e9,
                         "streaming_scan": scan, "streaming_scan_hbm_resident": big},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(M, args.cpu_iterations)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        pkg.sharded.shutdown_native_exchange(ctx)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
