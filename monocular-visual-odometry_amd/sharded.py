"""The pose loop over correspondences sharded across the GPUs of one node.

Every rank (one process per GPU) holds a contiguous slice of the five correspondence arrays and a
replica of the loop state.  One iteration is

    stage 1 scan (local)  ->  all-reduce of {sum chi, count, in-image count}      (4 doubles)
    stage 2 scan (local)  ->  all-reduce of the inlier set's H, b, sum chi, count (32 doubles)
    solve + schedules (replicated: identical inputs give identical state on every rank)

The two exchanges are `torch.distributed.all_reduce` calls (RCCL over xGMI with the "nccl"
backend).  The reference needs the global mean of chi before it can choose the inliers
(src/ProjectiveICP.py:301-310), hence two exchanges per iteration rather than one.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check


def shard_bounds(M: int, rank: int, world: int):
    """Contiguous, near-equal slice [lo, hi) of M correspondences for ``rank``.  Boundaries are
    even numbers so that every slice of 16-byte aligned arrays stays 16-byte aligned."""
    per = -(-M // world)
    per += per & 1
    lo = min(rank * per, M)
    hi = min(lo + per, M)
    return lo, hi


class HipStages:
    """The loop's stages on this rank's GPU (csrc/picp_icp.hip) with caller-owned exchange records."""

    def __init__(self, ctx, corr, w_T_c0, cam, sched, want_history=True):
        self.ctx, self.corr, self.cam, self.sched = ctx, corr, cam, sched
        self.M = int(corr[0].numel())
        self.pose = np.ascontiguousarray(np.asarray(w_T_c0, dtype=np.float64).reshape(16))
        self.state = ctx.empty(_lib.STATE_DOUBLES)
        self.s1 = ctx.zeros(_lib.S1_DOUBLES)
        self.s2 = ctx.zeros(_lib.S2_DOUBLES)
        self.rows = max(int(sched.num_iterations), 1)
        self.hist = ctx.zeros((self.rows, _lib.HIST_STRIDE)) if want_history else None
        self._corr_ptrs = [ctx.ptr(c) for c in corr]

    def state_init(self):
        check(self.ctx.lib.picp_state_init(self.ctx.handle, self.ctx.ptr(self.state),
                                           self.pose.ctypes.data_as(C.c_void_p), C.byref(self.sched)), "picp_state_init")

    def stage1(self):
        lib, h = self.ctx.lib, self.ctx.handle
        check(lib.picp_stage1(h, *self._corr_ptrs, self.M, self.ctx.ptr(self.state), C.byref(self.cam)), "picp_stage1")
        check(lib.picp_stage1_reduce(h, self.ctx.ptr(self.s1)), "picp_stage1_reduce")
        return self.s1

    def stage2(self, s1):
        lib, h = self.ctx.lib, self.ctx.handle
        check(lib.picp_stage2(h, *self._corr_ptrs, self.M, self.ctx.ptr(self.state), C.byref(self.cam),
                              self.ctx.ptr(s1)), "picp_stage2")
        check(lib.picp_stage2_reduce(h, self.ctx.ptr(self.s2)), "picp_stage2_reduce")
        return self.s2

    def solve(self, s1, s2):
        check(self.ctx.lib.picp_solve(self.ctx.handle, self.ctx.ptr(self.state), C.byref(self.sched), self.ctx.ptr(s1),
                                      self.ctx.ptr(s2), self.ctx.ptr(self.hist), self.rows if self.hist is not None else 0),
              "picp_solve")

    def stopped(self):
        return bool(self.state[_lib.ST_STOP].item() != 0.0)

    def finish(self):
        res = _lib.ResultStruct()
        check(self.ctx.lib.picp_finish(self.ctx.handle, self.ctx.ptr(self.state), self.ctx.ptr(self.hist), self.rows,
                                       self.pose.ctypes.data_as(C.c_void_p), C.byref(res)), "picp_finish")
        hist = self.hist[: res.num_iterations].cpu().numpy() if self.hist is not None else None
        return res, hist


def _all_reduce_sum(t, group, through_host):
    import torch.distributed as dist
    if through_host and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def run_sharded(stages, num_iterations, group=None, check_every=8, disable_stop=False, through_host=None):
    """Drives ``stages`` (this rank's shard) through the loop with the two all-reduces per iteration.

    ``stages`` provides state_init / stage1 / stage2 / solve / stopped / finish (see HipStages).
    With ``group`` None and no initialised process group the exchanges are skipped (one shard).

    A round is stage 1 -> all-reduce -> stage 2 -> all-reduce -> solve.  It normally completes one
    iteration; when the mean-filtered inlier set turns out too small the solve step only switches
    stage 2 to the stage-1 set and the next round completes the iteration.  The loop therefore runs
    until the replicated state's stop flag is set (stop test fired, or num_iterations completed),
    reading the flag back -- a host synchronisation -- every ``check_every`` rounds, or only after
    num_iterations rounds when ``disable_stop`` is set.  The flag is identical on all ranks, so every
    rank leaves the loop after the same round.

    ``through_host``: stage the two records through host memory for the all-reduce.  Default: only
    when the process group's backend is not "nccl" (RCCL reduces device buffers in place; the gloo
    backend is a CPU transport and is used for rehearsals on machines without one GPU per rank)."""
    import torch.distributed as dist
    exchange = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    if through_host is None:
        through_host = exchange and dist.get_backend(group) != "nccl"
    stages.state_init()
    total = int(num_iterations)
    rounds = 0
    next_check = total if disable_stop else min(check_every, total)
    while total > 0 and rounds < 2 * total:
        s1 = stages.stage1()
        if exchange:
            _all_reduce_sum(s1, group, through_host)
        s2 = stages.stage2(s1)
        if exchange:
            _all_reduce_sum(s2, group, through_host)
        stages.solve(s1, s2)
        rounds += 1
        if rounds >= next_check:
            if stages.stopped():
                break
            next_check = rounds + (1 if disable_stop else check_every)
    return stages.finish()
