"""The sharded driver with the HIP stages: one shard without a process group, and two ranks that
share the box's single GPU (gloo carries the two small all-reduces; on a multi-GPU node the same
code runs one rank per GPU over RCCL)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import picp_oracle as po
from tests.helpers import ROOT, rel_err
from tests.synthetic import make_problem

pytestmark = pytest.mark.gpu
pkg = importlib.import_module("monocular-visual-odometry_amd")
L = pkg._lib


def _cam_sched(num_iterations, disable_stop):
    cam = L.camera_struct([[180, 0, 320], [0, 180, 240], [0, 0, 1]], 640, 480)
    sched = L.schedule_struct(num_iterations, 10, 1000, 10, 2000, 64, 1, 64, disable_stop)
    return cam, sched


def test_single_shard_stages_equal_one_call_loop():
    ctx = pkg.get_context(0)
    uv, pw, _ = make_problem(50001, seed=9)
    cam, sched = _cam_sched(25, False)
    corr = ctx.soa(uv, pw)
    res, hist = ctx.run(corr, np.eye(4), cam, sched)
    stages = pkg.sharded.HipStages(ctx, corr, np.eye(4), cam, sched)
    res2, hist2 = pkg.sharded.run_sharded(stages, 25, check_every=4)
    assert res2.num_iterations == res.num_iterations and res2.best_index == res.best_index
    assert np.array_equal(hist2[:, L.H_NUM_INLIERS], hist[:, L.H_NUM_INLIERS])
    assert rel_err(hist2[:, L.H_T:L.H_T + 16], hist[:, L.H_T:L.H_T + 16]) < 1e-12
    T_ref, valid_ref, h_ref = po.projective_icp(uv, pw, np.eye(4), po.IcpParams(num_iterations=25))
    assert hist2[:, L.H_NUM_INLIERS].astype(int).tolist() == h_ref["num_inliers"]
    assert rel_err(np.array(list(res2.T)).reshape(4, 4), T_ref) < 1e-9


def _rank(rank, world, port, M, num_iterations, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = importlib.import_module("monocular-visual-odometry_amd")
    ctx = p.get_context(0)
    uv, pw, _ = make_problem(M, seed=9)
    lo, hi = p.sharded.shard_bounds(M, rank, world)
    cam, sched = _cam_sched(num_iterations, False)
    stages = p.sharded.HipStages(ctx, ctx.soa(uv[lo:hi], pw[lo:hi]), np.eye(4), cam, sched)
    res, hist = p.sharded.run_sharded(stages, num_iterations, check_every=4)
    out.put((rank, np.array(list(res.T)).reshape(4, 4), int(res.valid), hist[:, p._lib.H_NUM_INLIERS].astype(int).tolist(),
             hist[:, p._lib.H_N_INSIDE].astype(int).tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_sharing_the_gpu():
    M, num_iterations = 30011, 25
    uv, pw, _ = make_problem(M, seed=9)
    T_ref, valid_ref, h_ref = po.projective_icp(uv, pw, np.eye(4), po.IcpParams(num_iterations=num_iterations), record_linear=True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mpctx = mp.get_context("spawn")
    out = mpctx.Queue()
    procs = [mpctx.Process(target=_rank, args=(r, 2, port, M, num_iterations, out)) for r in range(2)]
    for p in procs:
        p.start()
    results = [out.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, T, valid, n_inl, n_in in results:
        assert bool(valid) == valid_ref
        assert n_inl == h_ref["num_inliers"] and n_in == h_ref["n_inside"]
        assert rel_err(T, T_ref) < 1e-9
    assert np.array_equal(results[0][1], results[1][1])
