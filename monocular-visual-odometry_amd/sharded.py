"""The pose loop over correspondences sharded across the GPUs of one node.

Every rank (one process per GPU) holds a contiguous slice of the five correspondence arrays and a
replica of the loop state.  One iteration is

    stage 1 scan (local)  ->  all-reduce of {sum chi, count, in-image count}      (4 doubles)
    stage 2 scan (local)  ->  all-reduce of the inlier set's H, b, sum chi, count (32 doubles)
    solve + schedules (replicated: identical inputs give identical state on every rank)

The reference needs the global mean of chi before it can choose the inliers
(src/ProjectiveICP.py:301-310), hence two exchanges per iteration rather than one.

Two ways to run the exchanges:
  * native (`init_native_exchange`): the library owns an RCCL communicator and enqueues
    ncclAllReduce on the same stream as its kernels, whole rounds per call (`picp_sharded_rounds`):
    an iteration is one in-order chain on one stream and costs the host one call;
  * generic: `torch.distributed.all_reduce` between the stage calls (any backend; with gloo the
    records are staged through host memory).  Used for rehearsals without one GPU per rank and by
    the CPU tests, and as the fallback when the native communicator cannot be set up.
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
        self._host_state = None
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
        self._host_state = None
        check(self.ctx.lib.picp_solve(self.ctx.handle, self.ctx.ptr(self.state), C.byref(self.sched), self.ctx.ptr(s1),
                                      self.ctx.ptr(s2), self.ctx.ptr(self.hist), self.rows if self.hist is not None else 0),
              "picp_solve")

    def rounds(self, n):
        """Enqueues n rounds, the all-reduces included when the ctx has a communicator."""
        self._host_state = None
        check(self.ctx.lib.picp_sharded_rounds(self.ctx.handle, *self._corr_ptrs, self.M, self.ctx.ptr(self.state),
                                               C.byref(self.cam), C.byref(self.sched), self.ctx.ptr(self.s1),
                                               self.ctx.ptr(self.s2), self.ctx.ptr(self.hist),
                                               self.rows if self.hist is not None else 0, int(n)), "picp_sharded_rounds")

    @property
    def comm_world(self):
        """Ranks of the ctx's own RCCL communicator (0: it has none)."""
        return int(self.ctx.lib.picp_comm_world(self.ctx.handle))

    @property
    def native_exchange(self):
        """True when `rounds` issues the exchange itself (the ctx has a communicator)."""
        return self.comm_world >= 1

    def stopped(self):
        """Reads the replicated state block back (one synchronising copy, kept for `finish`)."""
        self._host_state = self.state.cpu().numpy()
        return bool(self._host_state[_lib.ST_STOP] != 0.0)

    def state_checksum(self):
        """62-bit checksum of the replicated state block (see `state_checksum`), for `check_replicas`."""
        return state_checksum(self.state)

    def finish(self):
        res = _lib.ResultStruct()
        h = getattr(self, "_host_state", None)
        if h is not None and h[_lib.ST_STOP] != 0.0:
            # the state read back by `stopped` is final: the best-iteration rule of picp_finish on it
            # (src/ProjectiveICP.py:249-257), without a second synchronising copy
            n_it = int(h[_lib.ST_IT])
            valid = n_it > 0 and not (h[_lib.ST_BEST_ERR] > 5.0)
            T = h[_lib.ST_BEST_T:_lib.ST_BEST_T + 16] if valid else self.pose
            for i in range(16):
                res.T[i] = float(T[i])
            res.valid, res.num_iterations = int(valid), n_it
            res.best_index, res.best_error = int(h[_lib.ST_BEST_IT]), float(h[_lib.ST_BEST_ERR])
            hist = self.hist[:n_it].cpu().numpy() if self.hist is not None else None
            return res, hist
        check(self.ctx.lib.picp_finish(self.ctx.handle, self.ctx.ptr(self.state), self.ctx.ptr(self.hist), self.rows,
                                       self.pose.ctypes.data_as(C.c_void_p), C.byref(res)), "picp_finish")
        hist = self.hist[: res.num_iterations].cpu().numpy() if self.hist is not None else None
        return res, hist


def _injected(what):
    """Failure injection for rehearsals (PICP_INJECT_FAILURE=<what>[,<what>...]): the LAST rank pretends that
    step failed, so that the collective fallbacks can be exercised on a healthy machine."""
    import os
    import torch.distributed as dist
    if what not in os.environ.get("PICP_INJECT_FAILURE", "").split(","):
        return False
    multi = dist.is_available() and dist.is_initialized()
    return (not multi) or dist.get_rank() == dist.get_world_size() - 1


COMM_INIT_TIMEOUT_S = 120.0


def init_native_exchange(ctx, group=None, timeout_s=None):
    """Gives ``ctx`` an RCCL communicator over the ranks of ``group`` (default: the world), for
    `HipStages.rounds`.  The id RCCL needs is made on rank 0 and handed round through the existing
    process group, whatever its backend.  Collective: every rank must call it.  Returns True when all
    ranks have their communicator; otherwise every rank is left without one (and the generic
    exchange is used).

    ncclCommInitRank blocks until every rank has joined, so a rank that cannot join must be known
    BEFORE anyone calls it: the ranks first agree, through the process group, that each of them can load
    RCCL and holds the id.  The call itself then runs in a helper thread and is given ``timeout_s``
    (default COMM_INIT_TIMEOUT_S, or PICP_COMM_INIT_TIMEOUT): a rank whose peers never arrive reports
    failure instead of waiting for ever (its helper thread stays blocked; the process goes on without
    a communicator)."""
    import os
    import threading
    import torch
    import torch.distributed as dist
    lib = ctx.lib
    if int(lib.picp_comm_world(ctx.handle)) >= 1:
        return True
    multi = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if multi else 0
    world = dist.get_world_size(group) if multi else 1
    box = [None]
    if rank == 0:
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        if lib.picp_comm_unique_id(buf, _lib.COMM_ID_BYTES) == _lib.PICP_OK:
            box[0] = bytes(buf.raw)
    if world > 1:
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    ready = box[0] is not None and int(lib.picp_comm_available()) == 1 and not _injected("comm_ready")
    if not _agree(ready, group):                       # nobody enters ncclCommInitRank unless everybody will
        return False
    if timeout_s is None:
        timeout_s = float(os.environ.get("PICP_COMM_INIT_TIMEOUT", COMM_INIT_TIMEOUT_S))
    done = {}

    def join():
        done["rc"] = lib.picp_comm_init(ctx.handle, box[0], rank, world)

    if _injected("comm_init"):
        ok = False                                     # this rank never joins: the others run into their time limit
    else:
        t = threading.Thread(target=join, daemon=True)
        t.start()
        t.join(timeout_s)
        ok = (not t.is_alive()) and done.get("rc") == _lib.PICP_OK
        if t.is_alive():
            import sys
            print(f"rank {rank}: picp_comm_init did not return within {timeout_s:.0f} s (a rank never joined); "
                  "going on without the library's communicator", file=sys.stderr)
            ctx._comm_init_stuck = True                # the communicator may still appear later: never use or destroy it
    ok = _agree(ok, group)
    if not ok and not getattr(ctx, "_comm_init_stuck", False):
        lib.picp_comm_destroy(ctx.handle)
    return ok


def shutdown_native_exchange(ctx):
    """Destroys the ctx's RCCL communicator (collective; call it on every rank before the process
    group is torn down, not at interpreter exit)."""
    check(ctx.lib.picp_comm_destroy(ctx.handle), "picp_comm_destroy")


def init_peer_exchange(ctx, group=None):
    """Sets up the in-kernel exchange between the GPUs of ``group`` (default: the world) on ``ctx``: every
    rank allocates its exchange memory, the HIP IPC handles travel through the existing process group
    (whatever its backend) together with each GPU's PCI bus id and the host's name, every rank checks that
    its GPU can map every peer's memory (same host, peer access between the two GPUs) and only then maps it.
    Collective: every rank must call it.  Returns True when every rank is connected; otherwise every rank is
    left without an exchange (and `run_sharded` / `picp_sharded_rounds` are used), and the rank that found
    the obstacle says on stderr which pair of GPUs it was.  At most PEER_MAX_RANKS ranks, one node."""
    import socket
    import sys
    import torch.distributed as dist
    lib = ctx.lib
    if int(lib.picp_peer_world(ctx.handle)) >= 1:
        return True
    multi = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if multi else 0
    world = dist.get_world_size(group) if multi else 1
    ok = world <= _lib.PEER_MAX_RANKS
    buf = C.create_string_buffer(_lib.PEER_HANDLE_BYTES)
    ok = ok and not _injected("peer_create") and lib.picp_peer_create(ctx.handle, rank, world, buf) == _lib.PICP_OK
    if ok and _injected("peer_uncached") and peer_memory_kind(ctx).startswith("uncached"):
        ok = False                                 # rehearsal: as if uncached memory could not be shared between processes
        print(f"rank {rank}: injected failure (PICP_INJECT_FAILURE=peer_uncached): uncached exchange memory refused", file=sys.stderr)
    bus = C.create_string_buffer(_lib.BUS_ID_BYTES)
    bus_id = bus.value if lib.picp_device_bus_id(ctx.handle, bus, _lib.BUS_ID_BYTES) == _lib.PICP_OK else b""
    mine = (bytes(buf.raw) if ok else None, bus_id, socket.gethostname())
    everyone = [mine]
    if world > 1:
        everyone = [None] * world
        dist.all_gather_object(everyone, mine, group=group)           # also: every rank is past its create
    ok = all(h is not None for h, _, _ in everyone)
    if ok and world > 1:
        for peer, (_, peer_bus, peer_host) in enumerate(everyone):
            if peer == rank:
                continue
            can = C.c_int32(-1)
            if peer_host != mine[2]:
                can.value = 0
                why = f"it is on another host ({peer_host!r}, this rank on {mine[2]!r}): the exchange works inside one node"
            else:
                lib.picp_peer_can_access(ctx.handle, peer_bus, C.byref(can))
                why = "hipDeviceCanAccessPeer says no"
            if _injected("peer_access") and peer == 0:
                can.value, why = 0, "injected failure (PICP_INJECT_FAILURE=peer_access)"
            if can.value == 0:                                        # -1: not visible from here; the mapping will tell
                print(f"rank {rank} (GPU {bus_id.decode() or '?'}) cannot map the exchange memory of rank {peer} "
                      f"(GPU {peer_bus.decode() or '?'}): {why}", file=sys.stderr)
                ok = False
    ok = _agree(ok, group)                                              # nobody maps anything unless every pair is fine
    if ok and world > 1:
        ok = lib.picp_peer_connect(ctx.handle, b"".join(h for h, _, _ in everyone)) == _lib.PICP_OK
        if not ok:
            print(f"rank {rank}: {lib.picp_last_error().decode()}", file=sys.stderr)
        ok = _agree(ok, group)                                          # also: every rank is past its connect
    if not ok:
        lib.picp_peer_destroy(ctx.handle)
    return ok


def peer_world(ctx):
    """Ranks of the ctx's usable in-kernel exchange (0: none, or shut down after a failure)."""
    return int(ctx.lib.picp_peer_world(ctx.handle))


def peer_memory_kind(ctx):
    """How the ctx's exchange memory was allocated ("uncached device memory", or "fine-grained device memory"
    where the former is refused; "none" without an exchange)."""
    return _lib.PEER_MEMORY_KINDS.get(int(ctx.lib.picp_peer_memory_kind(ctx.handle)), "unknown")


def shutdown_peer_exchange(ctx, group=None):
    """Unmaps and frees the exchange memory (collective: every rank has finished its loops)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.barrier(group=group)
    check(ctx.lib.picp_peer_destroy(ctx.handle), "picp_peer_destroy")


def run_peer(ctx, corr, w_T_c0, cam, sched, want_history=True, group=None, fallback=True):
    """`Context.run_sharded` with the outcome AGREED between the ranks before anyone acts on it.

    picp_run_sharded reports per rank: one rank can give up at the last exchange after it has published its
    own record, while a later rank finds every record present and returns success.  A caller that fell back
    on the failing rank alone would then enter a collective the others never join.  Here every rank learns
    whether all succeeded (one small all-gather per loop); if not, all of them tear the exchange down and --
    with ``fallback`` -- run the same loop through the stage launches with the process group's all-reduces
    (`run_sharded`), so every rank returns the same result either way.  Returns (result, history, used_peer)."""
    res = hist = None
    ok = True
    try:
        if _injected("run_peer"):                  # rehearsal: this rank never launches, its peers run into their bound
            raise _lib.PicpRetry("injected failure (PICP_INJECT_FAILURE=run_peer)")
        res, hist = ctx.run_sharded(corr, w_T_c0, cam, sched, want_history=want_history)
    except _lib.PicpError:
        ok = False
    if _agree(ok, group):
        return res, hist, True
    shutdown_peer_exchange(ctx, group)
    if not fallback:
        raise _lib.PicpRetry("the in-kernel exchange failed on at least one rank; it has been shut down on all")
    stages = HipStages(ctx, corr, w_T_c0, cam, sched, want_history=want_history)
    res, hist = run_sharded(stages, int(sched.num_iterations), group=group, disable_stop=bool(sched.disable_stop),
                            native=False)
    return res, hist, False


def _agree(ok, group=None):
    """Logical AND of ``ok`` over the ranks (all ranks get the same answer)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return bool(ok)
    flags = [None] * dist.get_world_size(group)
    dist.all_gather_object(flags, bool(ok), group=group)
    return all(flags)


def state_checksum(state):
    """62-bit checksum of the bit patterns of a loop state block (device tensor of 96 doubles): a difference in
    any single bit of any entry changes it -- the last bit of a mantissa as well as a sign (the weighted sum is
    formed modulo 2^64 with odd weights, so no single bit is lost, and its top half is folded onto the bottom half
    before the two top bits, where a sign or top exponent bit lands, are cut off)."""
    import torch
    bits = state.view(torch.int64)
    weights = torch.arange(1, 2 * bits.numel(), 2, dtype=torch.int64, device=bits.device)
    total = int((bits * weights).sum().item()) & ((1 << 64) - 1)
    return (total ^ (total >> 32)) & ((1 << 62) - 1)


def replicas_identical(state, group=None):
    """Collective.  Compares the ranks' copies of the replicated loop state ``state`` (device tensor of 96
    doubles, e.g. the block a `picp_run_sharded` call left behind) bit for bit through their checksums: one
    all-reduce (MAX) of [c, -c], i.e. the maximum and the minimum over the ranks.  Returns
    (identical, lowest checksum, highest checksum), the same on every rank.  A store between GPUs that did
    not become visible, or one rank's sums differing in the last bit, shows here as False."""
    import torch
    import torch.distributed as dist
    c = state_checksum(state)
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return True, c, c
    on_gpu = dist.get_backend(group) == "nccl"
    t = torch.tensor([c, -c], dtype=torch.int64, device=state.device if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    hi, lo = int(t[0].item()), -int(t[1].item())
    return hi == lo, lo, hi


def check_replicas(stages, group=None):
    """Raises PicpError on EVERY rank when the replicated loop states of the ranks differ.

    The sharded loop relies on all ranks holding bit-identical state (same sums from the all-reduce, same
    deterministic solve): they then leave the loop after the same round.  If that ever fails -- a
    collective whose result differs between ranks, a rank fed different inputs -- ranks would enqueue
    different numbers of rounds and the job would hang in a later collective.  Called wherever the stop
    flag is read back: one all-reduce (MAX) of [c, -c] for a checksum c of the state, i.e. its maximum
    and minimum over the ranks; every rank sees both and so takes the same decision."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return
    if not hasattr(stages, "state_checksum"):
        return
    c = int(stages.state_checksum())
    on_gpu = dist.get_backend(group) == "nccl"
    device = stages.state.device if (on_gpu and hasattr(stages, "state")) else "cpu"
    t = torch.tensor([c, -c], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    hi, lo = int(t[0].item()), -int(t[1].item())
    if hi != lo:
        raise _lib.PicpError("sharded loop: the replicated loop state differs between ranks (state checksums "
                             f"{lo:#x} .. {hi:#x}, this rank {c:#x}); stopping instead of risking a hang in a collective")


def _all_reduce_sum(t, group, through_host):
    import torch.distributed as dist
    if through_host and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def run_sharded(stages, num_iterations, group=None, check_every=8, disable_stop=False, through_host=None,
                native=None):
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

    ``native``: let the library enqueue whole rounds, its own RCCL all-reduces included
    (`HipStages.rounds`).  Default: when ``stages`` can do that -- the ctx has a communicator
    (`init_native_exchange`) or there is only one shard; otherwise the generic loop below runs.

    ``through_host``: stage the two records through host memory for the all-reduce.  Default: only
    when the process group's backend is not "nccl" (RCCL reduces device buffers in place; the gloo
    backend is a CPU transport and is used for rehearsals on machines without one GPU per rank)."""
    import torch.distributed as dist
    exchange = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    ranks = dist.get_world_size(group) if exchange else 1
    covered = getattr(stages, "comm_world", 0) == ranks          # the ctx's communicator spans exactly these ranks
    if native is None:
        native = hasattr(stages, "rounds") and (covered or not exchange)
    if native:
        if exchange and not covered:
            raise _lib.PicpError("run_sharded(native=True) needs init_native_exchange over the same ranks first")
        return _run_native(stages, int(num_iterations), check_every, disable_stop, group)
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
            stopped = stages.stopped()
            if exchange:
                check_replicas(stages, group)
            if stopped:
                break
            next_check = rounds + (1 if disable_stop else check_every)
    return stages.finish()


def _run_native(stages, total, check_every, disable_stop, group=None):
    """The loop with whole rounds enqueued by the library (its own all-reduces in between, if the ctx
    has a communicator).  Same stop protocol as the generic loop: the flag is read back after every
    chunk of rounds, all ranks read the same value and so enqueue the same number of rounds."""
    stages.state_init()
    rounds = 0
    while total > 0 and rounds < 2 * total:
        n = (total if rounds == 0 else 1) if disable_stop else min(check_every, 2 * total - rounds)
        stages.rounds(n)
        rounds += n
        stopped = stages.stopped()
        check_replicas(stages, group)
        if stopped:
            break
    return stages.finish()


def row_bounds(N: int, rank: int, world: int):
    """Contiguous, near-equal slice [lo, hi) of N query rows for ``rank``."""
    per = -(-N // world)
    lo = min(rank * per, N)
    return lo, min(lo + per, N)


def associate_sharded(associate, app1_local, app2, lo, group=None):
    """Appearance association with the query rows (set 1) sharded and set 2 replicated.

    ``associate(app1_local, app2) -> (idx1, idx2)`` is the single-device association (``Context.associate``;
    int64 tensors, query-major order, idx1 local to the shard).  Rank r holds the rows
    [lo, hi) = row_bounds(N1, r, world) of set 1.  The pairs of a shard are already in the reference's
    order (src/data_association.py:29-30) and the shards are contiguous in rank order, so the global
    list is the concatenation of the shards' lists: one all-gather of the pair counts and one of the
    (padded) index lists, no other exchange.  Returns (idx1, idx2) with global set-1 row numbers,
    identical on every rank; with no process group it is the local result shifted by ``lo``."""
    import torch
    import torch.distributed as dist
    i1, i2 = associate(app1_local, app2)
    i1 = i1 + int(lo)
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return i1, i2
    world = dist.get_world_size(group)
    through_host = dist.get_backend(group) != "nccl" and i1.is_cuda
    dev = i1.device
    if through_host:
        i1, i2 = i1.cpu(), i2.cpu()
    n = torch.tensor([i1.numel()], dtype=torch.int64, device=i1.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    width = max(max(counts), 1)
    mine = torch.zeros((2, width), dtype=torch.int64, device=i1.device)
    mine[0, : i1.numel()] = i1
    mine[1, : i2.numel()] = i2
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    g1 = torch.cat([p[0, :c] for p, c in zip(parts, counts)])
    g2 = torch.cat([p[1, :c] for p, c in zip(parts, counts)])
    return g1.to(dev), g2.to(dev)
