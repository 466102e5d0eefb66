"""Soak of the exchange between ranks inside the single-launch loop (picp_run_sharded), with all ranks
on ONE GPU (gloo process group, exchange memory mapped through HIP IPC as between GPUs, every kernel
capped to its share of the compute units): many loops in a row on the same exchange memory, so that the
alternating slots and the exchange count that runs on from launch to launch are exercised thousands of
times.  Every loop must return the bits of the first one, on every rank, no launch may give up, and the
first loop must give the counts (exactly) and the pose (1e-9) of the C restatement of the reference loop
on the unsharded input.

    python tools/soak_peer.py [ranks [loops [correspondences ...]]]
"""
import importlib
import os
import socket
import sys
import time

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rank_main(rank, world, port, loops, sizes, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.synthetic import make_problem
    p = importlib.import_module("monocular-visual-odometry_amd")
    L = p._lib
    ctx = L.Context(0)
    ctx.set_grid_limit(192 // world)
    if os.environ.get("PICP_SOAK_STREAMED") == "1":            # every slice through the streaming form of the loop
        ctx.set_resident_limit(0)
    assert p.sharded.init_peer_exchange(ctx), "the exchange could not be set up"
    cam = L.camera_struct([[180, 0, 320], [0, 180, 240], [0, 0, 1]], 640, 480)
    sched = L.schedule_struct(20, 10, 1000, 10, 2000, 64, 1, 64, disable_stop=True)
    report = []
    for M in sizes:
        uv, pw, _ = make_problem(M, seed=[3, M % 1000])
        lo, hi = p.sharded.shard_bounds(M, rank, world)
        corr = ctx.soa(uv[lo:hi], pw[lo:hi])
        res, hist0 = ctx.run_sharded(corr, np.eye(4), cam, sched)
        T0 = np.array(list(res.T))
        differing = failed = 0
        dist.barrier()
        t0 = time.perf_counter()
        for k in range(loops):
            try:
                res, hist = ctx.run_sharded(corr, np.eye(4), cam, sched, want_history=(k % 64 == 0))
            except L.PicpError as e:
                failed += 1
                print(f"rank {rank}: loop {k} at M={M}: {e}", flush=True)
                break
            if not np.array_equal(np.array(list(res.T)), T0) or (hist is not None and not np.array_equal(hist, hist0)):
                differing += 1
        dt = time.perf_counter() - t0
        dist.barrier()
        report.append((M, loops, dt, differing, failed, T0, hist0 if rank == 0 else None))
        if failed:
            break
    out.put((rank, report))
    dist.barrier()
    p.sharded.shutdown_peer_exchange(ctx)
    dist.destroy_process_group()


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    loops = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    sizes = [int(a) for a in sys.argv[3:]] or [4096, 1 << 18, 1 << 22]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mpctx = mp.get_context("spawn")
    out = mpctx.Queue()
    procs = [mpctx.Process(target=rank_main, args=(r, world, port, loops, sizes, out)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(out.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    bad = 0
    from oracle import picp_oracle as po
    from oracle import picp_oracle_c as poc
    from tests.synthetic import make_problem
    L = importlib.import_module("monocular-visual-odometry_amd")._lib
    ints = [L.H_N_INSIDE, L.H_NUM_INLIERS, L.H_N_STAGE1, L.H_N_STAGE2, L.H_KERNEL_THR, L.H_DAMPING]
    for i, (M, n, dt, differing, failed, T0, hist0) in enumerate(got[0][1]):
        # the first loop against the C restatement of the reference loop on the whole, unsharded input
        uv, pw, _ = make_problem(M, seed=[3, M % 1000])
        T_ref, _, rows, _ = poc.projective_icp(uv, pw, np.eye(4), po.IcpParams(num_iterations=20, disable_stop=True))
        exact = np.array_equal(hist0[:, ints], rows[:, ints]) and np.abs(T0.reshape(4, 4) - T_ref).max() < 1e-9
        bad += 0 if exact else 1
        same = all(np.array_equal(T0, g[1][i][5]) for g in got[1:] if len(g[1]) > i)
        differ = sum(g[1][i][3] for g in got if len(g[1]) > i)
        fails = sum(g[1][i][4] for g in got if len(g[1]) > i)
        bad += differ + fails + (0 if same else 1)
        print(f"{world} ranks, M={M:9d}: {n} loops x 40 exchanges in {dt:6.2f} s ({dt / n * 1e3:.3f} ms per loop), "
              f"differing results {differ}, abandoned launches {fails}, ranks agree: {same}, counts and pose of the C restatement: {exact}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
