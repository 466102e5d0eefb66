"""The pose loop over correspondences sharded across the GPUs of one node.

Every rank (one process per GPU) holds a contiguous slice of the five correspondence arrays and a
replica of the loop state.  One iteration is

    stage 1 scan (local)  ->  all-reduce of {sum chi, count, in-image count}      (4 doubles)
    stage 2 scan (local)  ->  all-reduce of both candidate sets' H, b, chi, count (64 doubles)
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


def run_sharded(stages, num_iterations, group=None, check_every=8, disable_stop=False):
    """Drives ``stages`` (this rank's shard) through the loop with the two all-reduces per iteration.

    ``stages`` provides state_init / stage1 / stage2 / solve / stopped / finish (see HipStages).
    With ``group`` None and no initialised process group the exchanges are skipped (one shard).
    The stop flag is identical on all ranks, so every rank leaves the loop at the same iteration;
    it is read back every ``check_every`` iterations (a host synchronisation), never when
    ``disable_stop`` is set."""
    import torch.distributed as dist
    exchange = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    stages.state_init()
    for it in range(int(num_iterations)):
        s1 = stages.stage1()
        if exchange:
            dist.all_reduce(s1, op=dist.ReduceOp.SUM, group=group)
        s2 = stages.stage2(s1)
        if exchange:
            dist.all_reduce(s2, op=dist.ReduceOp.SUM, group=group)
        stages.solve(s1, s2)
        if not disable_stop and (it + 1) % check_every == 0 and it + 1 < num_iterations and stages.stopped():
            break
    return stages.finish()
