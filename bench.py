"""Benchmark of the projective-ICP pose loop (and of its two sibling configurations) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config loop|batched|assoc]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

--config loop (default; the driver's line).  A step is one complete pose loop: 20 LM iterations (stop
test disabled) over a synthetic correspondence set that is already resident in HBM.  N = 1 runs the
single-GPU configuration (2^20 correspondences); N > 1 shards 2^26 correspondences over the N GPUs.
The sharded loop is one launch per rank and step, with the two per-iteration sums exchanged between
the GPUs inside the kernel (peer memory over xGMI); if that exchange cannot be set up or fails its
trial against the other path, the run falls back to per-iteration stage launches with RCCL
all-reduces (enqueued by the library on the kernels' stream, or through torch.distributed).
config.collective says which path ran.  Rank 0 prints one JSON line.

metric = correspondence-iterations per second, whole job.
roofline = the dominant kernel, algorithmic bytes over its mean launch duration (HIP events bound to
the dispatch) against the 8 TB/s HBM3E peak.  Algorithmic bytes follow the reference's data flow, two
passes of 40 B per correspondence per iteration.  For the on-chip single-launch loop (N = 1, 2^20)
that figure is an equivalence, not memory traffic -- the kernel reads the arrays from HBM once -- so
the block also carries what really bounds that kernel: executed fp64 vector instructions against the
SIMDs' issue rate ("valu_utilisation", from the committed SQ counter pass) and the real HBM rate.
cpu_baseline = the C restatement of the reference loop (oracle/) on one host core, on a bounded sample.

--config batched: 4 096 independent problems x 2 048 correspondences, one workgroup each, 20 iterations.
--config assoc:   appearance association, 10^6 x 10^6 descriptors of 10 doubles.
Both print one JSON line of the same schema; with --gpus N the batched problems are split over the
ranks (replicas only, no exchange) and the association shards its query rows (set 2 replicated, the
index lists all-gathered).
"""
from __future__ import annotations

import argparse
import ctypes as C
import glob
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
SIMDS = 256 * 4                # compute units x SIMDs
CLOCK_HZ = 2.4e9               # nominal peak engine clock
FP64_CYCLES_PER_WAVE_INSTRUCTION = 4   # a 64-lane float64 vector instruction occupies its SIMD for 4 cycles (16 lanes / clock)
REFERENCE_LOOP_RATE = 1.8e3    # correspondence-iterations/s of the reference's own Python loop (src/ProjectiveICP.py:286-317),
                               # measured once in the build container on the bundled frames (1 thread); cannot run on the GPU box


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=["loop", "batched", "assoc"], default="loop")
    ap.add_argument("--corr", type=int, default=0, help="total correspondences (default 2^20 at 1 GPU, 2^26 sharded)")
    ap.add_argument("--method", choices=["hash", "scan"], default="hash",
                    help="--config assoc: which of picp_associate's two methods the line's value reports (hash: the "
                         "join it picks at this size; scan: the all-pairs test, the reference's double loop)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large-scan", action="store_true", help="skip the 2^24-correspondence stage-2 timing")
    ap.add_argument("--no-single-gpu", action="store_true",
                    help="sharded runs: skip rank 0's run of the whole workload on its one GPU")
    ap.add_argument("--cpu-iterations", type=int, default=300)
    return ap.parse_args()


def device_problem(ctx, M, lo, hi, seed=0, chunk=1 << 22):
    """Correspondences [lo, hi) of the seeded M-point problem as five device arrays, generated in
    chunks so the host never holds more than one chunk.  Chunk c uses seed (seed, c), so any
    sharding of the same M yields the same global data."""
    from tests.synthetic import make_problem
    n = hi - lo
    stride = importlib.import_module("monocular-visual-odometry_amd")._lib._pad32(n)
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


def committed_counters(pattern, kernel):
    """Per-launch counter values of `kernel` from the newest committed rocprofv3 counter summary that
    matches `pattern` (profiles/summarize.py writes them).  Hardware counters cannot be read from inside
    the measured process, so these come from separate --pmc runs of this same command, kept under
    profiles/; returns (values dict, file) or (None, None)."""
    paths = [q for pat in pattern.split("|") for q in glob.glob(os.path.join(ROOT, "profiles", pat))]
    for path in sorted(paths, key=os.path.basename, reverse=True):
        with open(path) as fh:
            k = json.load(fh)["kernels"]
        for name, v in k.items():
            if kernel in name:
                return v, os.path.relpath(path, ROOT)
    return None, None


def valu_block(counters, source, launch_ms):
    """Share of the SIMDs' float64 issue slots a launch used: executed vector instructions (SQ_INSTS_VALU,
    per wave) x 4 cycles over 1 024 SIMDs x launch duration at the nominal 2.4 GHz.  Integer and 32-bit
    vector instructions are priced like float64 ones, so the figure is an upper estimate."""
    if not counters or "SQ_INSTS_VALU" not in counters:
        return None
    insts = float(counters["SQ_INSTS_VALU"])
    busy_s = insts * FP64_CYCLES_PER_WAVE_INSTRUCTION / (SIMDS * CLOCK_HZ)
    return {"valu_wave_instructions_per_launch": insts, "valu_issue_time_ms": busy_s * 1e3,
            "valu_utilisation": busy_s / (launch_ms * 1e-3), "counter_source": source,
            "note": "committed counter pass of this command, not collected in this run"}


def cpu_baseline(M, iterations):
    """The C restatement of the reference loop (oracle/picp_oracle_c.c: one correspondence at a time
    in index order, as the reference works) on one host core, on the first `iterations` iterations of
    the same M-correspondence problem; the vectorised numpy restatement on a shorter sample beside it."""
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
    n_np = 12
    t0 = time.perf_counter()
    po.projective_icp(uv, pw, np.eye(4), po.IcpParams(num_iterations=n_np, disable_stop=True))
    dt_np = time.perf_counter() - t0
    return {"value": M * iterations / dt, "unit": "correspondence-iterations/s", "cores": 1, "kind": "port",
            "sample": f"C restatement of the reference loop (gcc -O2), {M} correspondences x {iterations} LM "
                      f"iterations, {dt:.1f} s on 1 of {os.cpu_count()} host CPUs; vectorised numpy restatement on the same "
                      f"correspondences x {n_np} iterations: {M * n_np / dt_np:.3g} correspondence-iterations/s "
                      f"({dt_np:.1f} s, numpy elementwise = 1 core); the reference's own per-point Python loop "
                      f"(src/ProjectiveICP.py:286-317) ran at about {REFERENCE_LOOP_RATE:.3g} correspondence-iterations/s "
                      "in the build container (1 thread) and cannot run on the GPU box",
            "numpy_restatement_value": M * n_np / dt_np, "reference_python_loop_value": REFERENCE_LOOP_RATE}


def single_gpu_same_workload(ctx, M, cam, sched, pose0, sharded_result, single_device, world, steps=5):
    """Rank 0 alone, the other ranks waiting at a barrier: the whole M-correspondence problem (the same
    seeded data the shards hold) through picp_run on this one GPU, `steps` synchronous loops after two
    untimed ones.  Returns the block the sharded bench line carries as `single_gpu_same_workload`."""
    import torch
    if single_device:
        ctx.set_grid_limit(0)                       # rehearsal on one GPU: the other ranks are idle now
    whole = device_problem(ctx, M, 0, M)
    plan = ctx.plan(whole, cam, sched)
    for _ in range(2):
        one = plan(pose0)[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one = plan(pose0)[0]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    diff = float(np.max(np.abs(result_pose(one) - result_pose(sharded_result))))
    block = {"value": M * ITERATIONS / dt, "ms_per_step": dt * 1e3, "steps": steps, "correspondences": M,
             "max_pose_difference_to_sharded_result": diff,
             "same_best_index": int(one.best_index) == int(sharded_result.best_index),
             "note": "measured in this run: rank 0's GPU alone on the same correspondences (picp_run, one launch per "
                     "loop), the other ranks idle at a barrier"}
    del plan, whole
    torch.cuda.empty_cache()
    if single_device:
        ctx.set_grid_limit(192 // world)
    if diff > 1e-9:
        raise SystemExit(f"the sharded loop's pose differs from the one-GPU loop's on the same correspondences by {diff:.3g}")
    return block


def headline_roofline(algorithmic_GBs, roof):
    """bound / achieved / peak / unit / frac of a loop line.  A kernel that streams the arrays is priced against
    HBM with the algorithmic bytes.  The on-chip loop reads HBM once per launch, so pricing its 80 B per
    correspondence-iteration against HBM says nothing about the kernel (the figure can even pass 1): its bound is
    the SIMDs' float64 issue rate, and the headline is executed vector lane-operations over that peak (the
    instruction count from the committed SQ counter pass); the algorithmic-bytes equivalence stays beside it as
    algorithmic_GBs / algorithmic_frac."""
    valu = roof.get("valu")
    if "hbm_bytes_read_once" in roof and valu:
        peak = SIMDS * 16 * CLOCK_HZ / 1e12
        achieved = valu["valu_wave_instructions_per_launch"] * 64 / (roof["mean_launch_ms"] * 1e-3) / 1e12
        return {"bound": "fp64_valu", "achieved": achieved, "peak": peak, "unit": "1e12 float64 lane-operations/s",
                "frac": achieved / peak, "frac_real_bound": valu["valu_utilisation"],
                "peak_derivation": "1 024 SIMDs x 16 float64 lanes per clock x 2.4 GHz",
                "algorithmic_GBs": algorithmic_GBs, "algorithmic_frac": algorithmic_GBs / HBM_PEAK_GBS,
                "algorithmic_peak_GBs": HBM_PEAK_GBS}
    return {"bound": "hbm", "achieved": algorithmic_GBs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": algorithmic_GBs / HBM_PEAK_GBS}


def streamed_block(ctx, n_local, launch_ms):
    """What the streaming single launch really asks of memory: the correspondences it holds on chip are read once
    per launch, the others twice per iteration (and of those reads the 256 MB memory-side cache serves a part when
    the arrays are not much larger than it).  `achieved` above prices all correspondences at 80 B per iteration."""
    held = ctx.streaming_held(n_local)
    streamed = 2 * BYTES_PER_CORR_PER_SCAN * (n_local - held) * ITERATIONS + BYTES_PER_CORR_PER_SCAN * held
    return {"held_on_chip_correspondences": held, "requested_bytes_per_launch": streamed,
            "requested_GBs": streamed / (launch_ms * 1e-3) / 1e9,
            "requested_note": "bytes the kernel's loads ask for (held correspondences once per launch, the others twice "
                              "per iteration); achieved / frac price every correspondence at 80 B per iteration, as "
                              "the metric's algorithmic figure does"}


def result_pose(res):
    return np.array(list(res.T)).reshape(4, 4)


# ---- sharded runs: choice of the exchange ------------------------------------------------------------
def choose_exchange(pkg, ctx, dist, world, backend, step_peer, step_stages, stage_state):
    """Sets up the exchange between the ranks and returns (mode, description).  Every decision is taken
    collectively (all ranks end on the same path).  The reference is the stage-launch loop with
    torch.distributed's own all-reduce between the stages, which needs nothing from this library beyond
    its kernels.  The in-kernel exchange (all of its waits are bounded, so a failure ends in an error, not
    in a hang) is tried first and used only after twenty trial loops through it have each given the reference's
    result and left bit-identical loop states on all ranks; failing that, the library's own RCCL communicator is
    tried the same way.
    stage_state(): the state block the stage-launch loops leave their final state in."""
    S = pkg.sharded
    want = os.environ.get("PICP_BENCH_EXCHANGE", "auto")   # auto | peer | rccl | torch
    rank = dist.get_rank()
    ref = step_stages("torch")

    def same(got):
        return (got.num_iterations == ref.num_iterations and got.best_index == ref.best_index and
                np.max(np.abs(result_pose(got) - result_pose(ref))) <= 1e-9)

    def loop_ms(fn, n=3):
        """Wall time of one loop through `fn`, the slowest rank's mean of n (the same number on every rank).
        `fn` must not fail here: it has just passed its trial on every rank."""
        import torch
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return max_over_ranks((time.perf_counter() - t0) / n, world, backend) * 1e3

    TRIAL_LOOPS = 20

    def trial(fn, state_of, what):
        """TRIAL_LOOPS loops through `fn`, each compared with the reference result and, bit for bit, between the
        ranks (checksum of the 96-double loop state every rank is left with).  Collective; returns (ok, why) the
        same on every rank.  A rank whose loop raises keeps taking part in the comparisons, so nobody waits alone."""
        for k in range(TRIAL_LOOPS):
            ok, why = True, ""
            try:
                got = fn()
                if not same(got):
                    ok, why = False, f"loop {k + 1} of its trial disagreed with the stage launches"
                if os.environ.get("PICP_INJECT_FAILURE") == "trial_bits" and rank == world - 1 and k == 7:
                    state_of()[5] += 1e-13                      # rehearsal: one rank's state differs in its last bits
            except pkg._lib.PicpError as e:
                ok, why = False, f"loop {k + 1} of its trial failed ({e})"
            identical = True
            if S._agree(ok):
                identical, lo_c, hi_c = S.replicas_identical(state_of())
                if not identical:
                    why = (f"after loop {k + 1} of its trial the ranks' loop states differed (checksums {lo_c:#x} .. "
                           f"{hi_c:#x})")
            elif not why:
                why = f"loop {k + 1} of its trial failed on another rank"
            if not (ok and identical) or why:
                if why:
                    print(f"rank {rank}: {what} not used: {why}", file=sys.stderr)
                return False, why or "its trial failed"
        return True, ""

    notes = []
    # The exchange memory is uncached device memory where the runtime grants it; should the set-up or the trial fail
    # with it, the whole attempt is repeated once with fine-grained memory (the same decision on every rank: it
    # depends on agreed outcomes and on the environment the launcher gave all ranks).
    kinds = [] if want not in ("auto", "peer") else ([None, "finegrained"] if "PICP_PEER_MEMORY" not in os.environ else [None])
    for kind in kinds:
        again = " (second attempt, fine-grained exchange memory)" if kind else ""
        if kind:
            os.environ["PICP_PEER_MEMORY"] = kind
        if not S.init_peer_exchange(ctx):
            notes.append(f"the in-kernel exchange could not be set up{again} (see stderr for the pair of GPUs)")
            continue
        ok, why = trial(step_peer, lambda: ctx._run_state, "in-kernel exchange")
        if not ok:
            notes.append(f"the in-kernel exchange was set up but failed its trial{again} ({why})")
            S.shutdown_peer_exchange(ctx)
            continue
        # both paths work: keep the in-kernel exchange unless the stage launches are faster on this
        # machine (they are not where it was rehearsed; between real GPUs the peer stores are untried)
        text = (f"in-kernel exchange through peer memory (HIP IPC, xGMI stores into {S.peer_memory_kind(ctx)}), one "
                f"launch per loop; {TRIAL_LOOPS} trial loops equal to the stage launches' result and bit-identical "
                "between the ranks")
        if notes:
            text += "; before that: " + "; ".join(notes)
        if want == "peer":
            return "peer", text, None
        try:
            ms_peer, failed = loop_ms(step_peer), False
        except pkg._lib.PicpError as e:
            print(f"rank {rank}: in-kernel exchange failed while being timed: {e}", file=sys.stderr)
            ms_peer, failed = float("inf"), True
            max_over_ranks(0.0, world, backend)                  # the collective of loop_ms a rank without the failure is in
        failed = not S._agree(not failed)
        ms_stages = loop_ms(lambda: step_stages("torch"))
        if not failed and ms_peer <= ms_stages:
            return "peer", (f"{text} ({ms_peer:.3f} ms per loop in the trial; stage launches with {backend} "
                            f"all-reduces {ms_stages:.3f} ms)"), ms_peer
        why = ("it failed while being timed" if failed else
               f"the stage launches were faster in the trial ({ms_stages:.3f} ms per loop against {ms_peer:.3f} ms)")
        notes.append(f"the in-kernel exchange passed its trial but {why}")
        print(f"rank {rank}: in-kernel exchange not used: {why}", file=sys.stderr)
        S.shutdown_peer_exchange(ctx)
        break
    # (on gloo only for rehearsals of the set-up's failure paths: RCCL refuses two ranks on one device)
    native_allowed = backend == "nccl" or want == "rccl" or os.environ.get("PICP_BENCH_NATIVE_ON_GLOO") == "1"
    if want in ("auto", "rccl") and native_allowed and S.init_native_exchange(ctx):
        # one all-reduce of known values through the library's communicator, then a trial
        probe = ctx.to_device(np.full(4, float(rank + 1)))
        ok = ctx.lib.picp_comm_allreduce(ctx.handle, ctx.ptr(probe), 4) == 0
        ctx.synchronize()
        ok = ok and bool(np.all(probe.cpu().numpy() == world * (world + 1) / 2))
        if S._agree(ok):
            ok, why = trial(lambda: step_stages("rccl"), lambda: stage_state(), "library all-reduce")
        if ok:
            return "rccl", "; ".join(["rccl all-reduces enqueued by the library on the kernels' stream, two launches per "
                                      f"iteration; {TRIAL_LOOPS} trial loops bit-identical between the ranks"] + notes), None
        S.shutdown_native_exchange(ctx)
        notes.append("the library's own RCCL communicator failed its trial")
    elif want in ("auto", "rccl") and native_allowed:
        notes.append("the library's own RCCL communicator could not be set up")
    return "torch", "; ".join([f"{backend} all-reduces through torch.distributed between stage launches"] + notes), None


SETTLE_SECONDS = 0.03          # see settle_device


def settle_device(step, seconds=SETTLE_SECONDS, most=2000):
    """Runs `step` back to back for `seconds` of wall time before the warm-up steps.  After an idle period
    (the set-up of the inputs is one) the device needs about 12 ms of continuous work to reach its steady
    clocks: measured with events bound to each dispatch, the first ten 2^20 x 20 loops of a process, or
    after half a second of idling, take 250 us each, the next forty come down to 231 us one by one, and
    every later one stays there (tools/settle_probe.py).  A run of `--warmup 5 --steps 20` lies inside that
    ramp from its first to its last loop and would measure the ramp, not the kernel.  The W warm-up steps
    and the K timed steps follow this phase unchanged; the bench line says how long it was."""
    t0 = time.perf_counter()
    stamps = [t0]
    while len(stamps) - 1 < most and stamps[-1] - t0 < seconds:
        step()
        stamps.append(time.perf_counter())
    n = len(stamps) - 1
    k = min(10, n)
    return {"loops": n, "ms": (stamps[-1] - t0) * 1e3,
            "first_loops_ms_per_loop": (stamps[k] - stamps[0]) / k * 1e3,      # synchronous calls, one after the other:
            "last_loops_ms_per_loop": (stamps[-1] - stamps[-1 - k]) / k * 1e3,   # the ramp shows as the difference
            "note": "untimed loops run back to back before the warm-up steps until the device has left its clock ramp "
                    "after idling (about 12 ms of continuous work); the warm-up and timed steps are as requested"}


def init_ranks(args):
    """One process per GPU: reads the launcher's environment, selects this rank's device and joins the
    process group.  Returns (world, rank, local_rank, backend, single_device)."""
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
    # device 0, each kernel capped to its share of the compute units, and gloo for the process group,
    # since RCCL refuses two ranks on one device.
    backend = os.environ.get("PICP_BENCH_BACKEND", "nccl")
    single_device = os.environ.get("PICP_BENCH_SINGLE_DEVICE") == "1"
    if single_device:
        local_rank = 0
    elif torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: --gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) are visible "
                         "(PICP_BENCH_SINGLE_DEVICE=1 PICP_BENCH_BACKEND=gloo rehearses the ranks on one GPU)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL over xGMI
        else:
            dist.init_process_group(backend)
    return world, rank, local_rank, backend, single_device


def max_over_ranks(dt, world, backend):
    """The slowest rank's time (every rank gets it)."""
    if world == 1:
        return dt
    import torch
    import torch.distributed as dist
    t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def bench_loop(args):
    if os.environ.get("PICP_BENCH_DEBUG"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["PICP_BENCH_DEBUG"]), exit=True)
    import torch
    import torch.distributed as dist
    world, rank, local_rank, backend, single_device = init_ranks(args)
    from tests.synthetic import true_pose
    pkg = importlib.import_module("monocular-visual-odometry_amd")
    L = pkg._lib
    ctx = pkg.get_context(local_rank)
    if single_device and world > 1:
        ctx.set_grid_limit(192 // world)

    M = args.corr or ((1 << 20) if world == 1 else (1 << 26))
    lo, hi = pkg.sharded.shard_bounds(M, rank, world)
    corr = device_problem(ctx, M, lo, hi)
    cam = L.camera_struct([[180, 0, 320], [0, 180, 240], [0, 0, 1]], 640, 480)
    sched = L.schedule_struct(ITERATIONS, 10, 1000, 10, 2000, 64, 1, 64, disable_stop=True)
    pose0 = np.eye(4)

    plan = ctx.plan(corr, cam, sched) if world == 1 else None     # picp_run with its arguments marshalled once

    stage_set = []

    def step_stages(mode):
        if not stage_set:                                       # state block and exchange records: once, not per step
            stage_set.append(pkg.sharded.HipStages(ctx, corr, pose0, cam, sched, want_history=False))
        res, _ = pkg.sharded.run_sharded(stage_set[0], ITERATIONS, disable_stop=True, native=(mode == "rccl"))
        return res

    peer_calls = [0]
    fail_at = int(os.environ.get("PICP_BENCH_FAIL_STEP", "0"))   # rehearsal of the fallback: the last rank drops out at this call

    def step_peer():
        peer_calls[0] += 1
        if fail_at and rank == world - 1 and peer_calls[0] == fail_at:
            raise L.PicpError("injected failure (PICP_BENCH_FAIL_STEP)")
        return ctx.run_sharded(corr, pose0, cam, sched, want_history=False)[0]

    mode, collective, trial_ms = None, None, None
    if world > 1:
        ctx.lib.picp_set_launch_timing(ctx.handle, 1)
        mode, collective, trial_ms = choose_exchange(pkg, ctx, dist, world, backend, step_peer, step_stages,
                                                     lambda: stage_set[0].state)

    def step():
        if world == 1:
            return plan(pose0)[0]
        return step_peer() if mode == "peer" else step_stages(mode)

    fences = [0]

    def fence():
        fences[0] += 1
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    launch_ms = []

    def run_steps(n, record=False):
        """n steps.  On one GPU the loops are independent of each other, so two are kept in flight
        (picp_run_enqueue / picp_run_collect): the host enqueues the next loop while the current one
        runs instead of waiting in between.  Every loop's result is collected and checked."""
        last = None
        if world > 1 or n == 0 or os.environ.get("PICP_BENCH_PIPELINE", "1") == "0":
            for _ in range(n):
                last = step()
                if record and mode == "peer":
                    ms = C.c_double()
                    ctx.lib.picp_last_launch_ms(ctx.handle, C.byref(ms))
                    launch_ms.append(float(ms.value))
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

    def timed_region():
        if world == 1:
            settle_info.update(settle_device(step))
        else:
            t0 = time.perf_counter()
            for _ in range(sharded_settle_loops):
                step()
            settle_info.update({"loops": sharded_settle_loops, "ms": (time.perf_counter() - t0) * 1e3,
                                "note": "untimed loops before the warm-up steps (the device's clock ramp after idling), "
                                        "the same number on every rank"})
        res = run_steps(args.warmup) or step()
        fence()
        t0 = time.perf_counter()
        res = run_steps(args.steps, record=True) or res
        fence()
        return res, time.perf_counter() - t0

    # one GPU: a loop is a quarter of a millisecond, so the requested warm-up alone would leave the timed steps
    # inside the device's clock ramp.  Sharded: every step is an exchange between the ranks, so all ranks run the
    # same number of them (from the trial's loop time, which is the slowest rank's and the same number everywhere).
    settle_info = {}
    sharded_settle_loops = 0
    if world > 1:
        sharded_settle_loops = int(min(max(np.ceil(SETTLE_SECONDS * 1e3 / trial_ms), 1), 500)) if trial_ms else 10

    if mode == "peer":
        # The in-kernel exchange has passed one trial loop, not thousands.  Its waits are bounded, so a
        # failure shows as an error from a step (on every rank within the same step: a rank that gives up
        # stops publishing); the steps themselves contain no collective of the process group, so every
        # rank reaches the agreement below at the same point, and all then repeat the measurement with
        # the stage launches.
        before = fences[0]
        try:
            res, dt = timed_region()
            ok = True
        except L.PicpError as e:
            print(f"rank {rank}: in-kernel exchange failed during the run: {e}", file=sys.stderr)
            ok = False
            while fences[0] < before + 2:              # the barriers a rank without the failure passes through
                fence()
        if not pkg.sharded._agree(ok):
            pkg.sharded.shutdown_peer_exchange(ctx)
            mode = "torch"
            collective = (f"{backend} all-reduces through torch.distributed between stage launches; the in-kernel "
                          "exchange passed its trial loop but failed during the run")
            launch_ms.clear()
            res, dt = timed_region()
    else:
        res, dt = timed_region()
    dt = max_over_ranks(dt, world, backend)
    assert res.num_iterations == ITERATIONS, res.num_iterations

    # Sharded: (1) every rank must have ended the timed steps with the same 96-double loop state, bit for bit
    # (the sums are added in rank order on every rank and the solve is deterministic); a peer store that did
    # not arrive whole, or arrived on one rank only, shows here and fails the run.  (2) rank 0 then runs the
    # SAME correspondences alone, so that the line carries its own one-GPU figure for this workload.
    replicas = single = None
    if world > 1:
        final_state = ctx._run_state if mode == "peer" else stage_set[0].state
        same_state, c_lo, c_hi = pkg.sharded.replicas_identical(final_state)
        replicas = {"replicas_identical": same_state, "state_checksum": f"{c_lo:#x}",
                    "note": "checksum of the final loop state (96 doubles) of the last timed step, maximum and minimum "
                            "over the ranks compared"}
        if not same_state:
            raise SystemExit(f"rank {rank}: the ranks ended with different loop states (checksums {c_lo:#x} .. {c_hi:#x}): "
                             "the exchange between the GPUs delivered different sums to different ranks")
        if not args.no_single_gpu:
            dist.barrier()
            if rank == 0:
                single = single_gpu_same_workload(ctx, M, cam, sched, pose0, res, single_device, world)
            dist.barrier()

    n_local = hi - lo
    s2_mean_ms, s2_min_ms = time_stage2(ctx, pkg, corr, cam, sched)
    resident = world == 1 and 0 < M <= ctx.resident_capacity()
    shard_on_chip = world > 1 and mode == "peer" and 0 < n_local <= ctx.resident_capacity()   # rank 0's slice
    if world == 1:
        run_mean_ms, run_min_ms = time_resident(ctx, pkg, corr, cam, sched)
    host_buffers = None
    if world == 1:
        # A caller that holds the five arrays in (pinned) host memory pays one host-to-device copy of
        # 40 B per correspondence before each loop; `value` does not include it, this block does.
        host = torch.empty((5, n_local), dtype=torch.float64).pin_memory()
        for k in range(5):
            host[k].copy_(corr[k])
        dev = torch.empty((5, n_local), dtype=torch.float64, device=corr[0].device)
        dev.copy_(host, non_blocking=True)
        torch.cuda.synchronize()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            dev.copy_(host, non_blocking=True)
        torch.cuda.synchronize()
        h2d_ms = (time.perf_counter() - t0) / reps * 1e3
        step_ms = dt / args.steps * 1e3
        host_buffers = {"h2d_ms_per_step": h2d_ms, "h2d_GBs": 40.0 * n_local / (h2d_ms * 1e-3) / 1e9,
                        "value_including_h2d": M * ITERATIONS / ((h2d_ms + step_ms) * 1e-3),
                        "note": "five float64 arrays copied from pinned host memory before every loop, copy and loop "
                                "one after the other; not what `value` reports (inputs resident in HBM)"}
        del host, dev
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
        loop_bytes = 2 * BYTES_PER_CORR_PER_SCAN * n_local * ITERATIONS
        traffic = traffic_source = None
        if resident:
            achieved = loop_bytes / (run_mean_ms * 1e-3) / 1e9
            once = BYTES_PER_CORR_PER_SCAN * n_local
            roof = {"kernel": "icp_resident_kernel (whole loop in one launch, correspondences in registers)",
                    "bytes_per_launch": loop_bytes, "mean_launch_ms": run_mean_ms, "min_launch_ms": run_min_ms,
                    "hbm_bytes_read_once": once,
                    "hbm_utilisation_real": once / (run_mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "real_bound": "fp64 vector issue of the two stages plus the latency of the in-kernel "
                                  "reductions, exchanges and the solve step; not HBM",
                    "note": "algorithmic_GBs / algorithmic_frac price the reference's data flow (80 B per "
                            "correspondence-iteration, two 40 B passes) against HBM, as the metric asks; the kernel itself "
                            "reads each array once per launch (hbm_utilisation_real) and is bound as real_bound says, "
                            "which is what bound / achieved / frac report"}
            if M == (1 << 20):
                tr, traffic_source = committed_counters("r*_pmc_traffic_loop*.json|r*_pmc_traffic_2p20*.json", "icp_resident_kernel")
                traffic = tr["hbm_bytes_per_launch"] if tr else None
                sq, sq_src = committed_counters("r*_pmc_sq_loop*.json|r*_pmc_sq_2p20*.json", "icp_resident_kernel")
                roof["valu"] = valu_block(sq, sq_src, run_mean_ms)
                roof["traffic_note"] = "from the committed FETCH_SIZE/WRITE_SIZE passes of this command, not collected in this run"
        elif world == 1:
            achieved = loop_bytes / (run_mean_ms * 1e-3) / 1e9
            roof = {"kernel": "icp_persistent_kernel (whole loop in one launch; ten correspondences per thread held in LDS and "
                              "registers, the other arrays streamed twice per iteration)",
                    "bytes_per_launch": loop_bytes, "mean_launch_ms": run_mean_ms, "min_launch_ms": run_min_ms,
                    **streamed_block(ctx, n_local, run_mean_ms)}
            size_name = {1 << 23: "2p23", 1 << 24: "2p24", 1 << 26: "2p26"}.get(M)
            if size_name:
                tr, traffic_source = committed_counters(f"r*_pmc_traffic_stream_{size_name}.json", "icp_persistent_kernel")
                traffic = tr["hbm_bytes_per_launch"] if tr else None
                roof["traffic_note"] = ("bytes the L2s fetched and wrote per launch, from the committed FETCH_SIZE / WRITE_SIZE "
                                        "passes of this command (the memory-side cache sits behind the L2s, so its hits are in "
                                        "this figure); not collected in this run")
        elif mode == "peer":
            mean_ms = sum(launch_ms) / max(len(launch_ms), 1)
            achieved = loop_bytes / (mean_ms * 1e-3) / 1e9
            roof = {"kernel": ("icp_resident_kernel, sharded (whole loop in one launch per rank, the slice held on chip, "
                               "sums exchanged between the GPUs inside the kernel); rank 0's launches" if shard_on_chip else
                               "icp_persistent_kernel, sharded (whole loop in one launch per rank; ten correspondences per thread "
                               "held in LDS and registers, the rest of the slice streamed twice per iteration; sums exchanged "
                               "between the GPUs inside the kernel); rank 0's launches"),
                    "bytes_per_launch": loop_bytes, "mean_launch_ms": mean_ms, "min_launch_ms": min(launch_ms)}
            if not shard_on_chip:
                roof.update(streamed_block(ctx, n_local, mean_ms))
        else:
            achieved = scan["achieved"]
            roof = {k: scan[k] for k in ("kernel", "bytes_per_launch", "mean_launch_ms", "min_launch_ms")}
        out = {
            "metric": "ICP correspondence-iterations/sec",
            "value": M * ITERATIONS * args.steps / dt,
            "unit": "correspondence-iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if world > 1 else None,
            "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("synthetic single-pose projective ICP, 2^20 correspondences, 20 LM iterations"
                                    if world == 1 and M == (1 << 20) else
                                    f"synthetic single-pose projective ICP, {M} correspondences on one GPU, 20 LM iterations"
                                    if world == 1 else
                                    f"synthetic single-pose projective ICP, {M} correspondences sharded over "
                                    f"{world} GPU(s), 20 LM iterations, 2 exchanges per iteration"),
                       "correspondences": M, "lm_iterations": ITERATIONS,
                       "steps_in_flight": 2 if (world == 1 and os.environ.get("PICP_BENCH_PIPELINE", "1") != "0") else 1,
                       "device_path": ("single launch, register-resident" if (resident or shard_on_chip) else
                                       "single launch, part of the arrays held on chip, the rest streamed twice per iteration"
                                       if (world == 1 or mode == "peer")
                                       else "streaming stage launches"),
                       "collective": collective,
                       "final_pose_error": float(np.max(np.abs(result_pose(res) - true_pose())))},
            "roofline": {**headline_roofline(achieved, roof), "traffic": traffic, "traffic_source": traffic_source,
                         **roof,
                         "loop_bytes_per_corr_iteration": 2 * BYTES_PER_CORR_PER_SCAN,
                         "loop_achieved_GBs": loop_bytes * args.steps / dt / 1e9,
                         "streaming_scan": scan, "streaming_scan_hbm_resident": big},
        }
        if replicas:
            out.update(replicas)
        if single:
            # what strong scaling is measured against: the N = 1 line of this bench is the 2^20 configuration, which
            # runs on chip, not this workload
            single["speedup"] = out["value"] / single["value"]
            single["efficiency"] = single["speedup"] / world
            out["single_gpu_same_workload"] = single
        if settle_info:
            out["settle"] = settle_info
        if host_buffers:
            out["host_buffers"] = host_buffers
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(M, args.cpu_iterations)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        if mode == "peer":
            pkg.sharded.shutdown_peer_exchange(ctx)
        pkg.sharded.shutdown_native_exchange(ctx)
        dist.destroy_process_group()


# ---- config 3: batched problems ------------------------------------------------------------------------
def bench_batched(args):
    """N > 1: the problems are independent, so the ranks take contiguous shares of them and exchange
    nothing (replicas only); value = all problems over the slowest rank's time."""
    import torch
    import torch.distributed as dist
    from oracle import picp_oracle as po
    from oracle import picp_oracle_c as oc
    from tests.synthetic import make_problem
    world, rank, local_rank, backend, single_device = init_ranks(args)
    pkg = importlib.import_module("monocular-visual-odometry_amd")
    L = pkg._lib
    ctx = pkg.get_context(local_rank)
    cam = L.camera_struct([[180, 0, 320], [0, 180, 240], [0, 0, 1]], 640, 480)
    sched = L.schedule_struct(ITERATIONS, 10, 1000, 10, 2000, 64, 1, 64, True)
    B_all, N = 4096, 2048
    p_lo, p_hi = pkg.sharded.row_bounds(B_all, rank, world)
    B = p_hi - p_lo
    uv = np.empty((B * N, 2))
    pw = np.empty((B * N, 3))
    for p in range(B):
        uv[p * N:(p + 1) * N], pw[p * N:(p + 1) * N], _ = make_problem(N, seed=p_lo + p)
    corr = ctx.soa(uv, pw)
    offsets = ctx.to_device(np.arange(B + 1, dtype=np.int64) * N, np.int64)
    poses0 = ctx.to_device(np.tile(np.eye(4).reshape(1, 16), (B, 1)))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def synchronous_step():
        ctx.run_batched(corr, offsets, poses0, cam, sched, max_count=N)
        torch.cuda.synchronize()

    settle = settle_device(synchronous_step)
    for _ in range(max(args.warmup, 1)):
        results, _ = ctx.run_batched(corr, offsets, poses0, cam, sched, max_count=N)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        results, _ = ctx.run_batched(corr, offsets, poses0, cam, sched, max_count=N)
    fence()
    dt = max_over_ranks(time.perf_counter() - t0, world, backend) / args.steps
    # launch duration from events bound to the stream around single launches
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ms = []
    for _ in range(10):
        ev[0].record()
        ctx.run_batched(corr, offsets, poses0, cam, sched, max_count=N)
        ev[1].record()
        ev[1].synchronize()
        ms.append(ev[0].elapsed_time(ev[1]))
    launch_ms = float(np.mean(ms))
    r = ctx.decode_results(results.cpu().numpy())
    assert (r["num_iterations"] == ITERATIONS).all() and r["valid"].all()
    units = B_all * N * ITERATIONS
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    sq, sq_src = committed_counters("r*_pmc_sq_batched*.json", "icp_batched_resident_kernel")
    if sq and "SQ_INSTS_VALU" in sq and B != B_all:          # the counter pass ran all 4 096 problems on one GPU
        sq = {"SQ_INSTS_VALU": float(sq["SQ_INSTS_VALU"]) * B / B_all}
    valu = valu_block(sq, sq_src, launch_ms)
    tr, tr_src = committed_counters("r*_pmc_traffic_batched*.json", "icp_batched_resident_kernel")
    peak = SIMDS * 16 * CLOCK_HZ / 1e12                          # float64 lane-operations per second, in 1e12
    achieved = (valu["valu_wave_instructions_per_launch"] * 64 / (launch_ms * 1e-3) / 1e12) if valu else None
    out = {"metric": "ICP correspondence-iterations/sec", "value": units / dt, "unit": "correspondence-iterations/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True,
           "scaling": "strong" if world > 1 else None, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"batched independent ICP, {B_all} problems x {N} correspondences, 20 LM iterations, "
                                  "one pose per workgroup", "problems": B_all, "correspondences_per_problem": N,
                      "lm_iterations": ITERATIONS, "problems_per_s": B_all / dt,
                      "collective": None if world == 1 else f"none: replicas only, {B} problems on rank 0"},
           "roofline": {"bound": "fp64_valu", "kernel": "icp_batched_resident_kernel (problem held in registers and LDS)",
                        "achieved": achieved, "peak": peak, "unit": "1e12 float64 lane-operations/s",
                        "frac": (achieved / peak) if achieved else None,
                        "peak_derivation": "1 024 SIMDs x 16 float64 lanes per clock x 2.4 GHz",
                        "mean_launch_ms": launch_ms, "valu": valu,
                        "traffic": (tr["hbm_bytes_per_launch"] * B / B_all) if tr else None, "traffic_source": tr_src,
                        "hbm_algorithmic_GBs": 40.0 * B * N / (launch_ms * 1e-3) / 1e9,
                        "problems_per_launch": B,
                        "note": "every correspondence is read from HBM once per solve (40 B); the loop runs on chip, so "
                                "the bound is the SIMDs' float64 issue rate: executed vector instructions x 64 lanes "
                                "over the launch duration, instruction count from the committed SQ counter pass"}}
    out["settle"] = settle
    if not args.no_cpu_baseline and world == 1:
        oc.load()
        params = po.IcpParams(num_iterations=ITERATIONS, disable_stop=True)
        n_cpu = 1024
        t0 = time.perf_counter()
        for p in range(n_cpu):
            oc.projective_icp(uv[p * N:(p + 1) * N], pw[p * N:(p + 1) * N], np.eye(4), params, want_history=False)
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n_cpu * N * ITERATIONS / dtc, "unit": "correspondence-iterations/s", "cores": 1,
                               "kind": "port",
                               "sample": f"C restatement of the reference loop, the first {n_cpu} of the {B} problems one "
                                         f"after the other, {dtc:.1f} s on 1 of {os.cpu_count()} host CPUs"}
    print(json.dumps(out), flush=True)


# ---- config 5: appearance association ------------------------------------------------------------------
def bench_assoc(args):
    """N > 1: the query rows (set 1) are sharded over the ranks, set 2 is replicated; a step is the
    association of every rank's rows plus the all-gathers that give every rank the whole pair list."""
    import torch
    import torch.distributed as dist
    from oracle import picp_oracle as po
    world, rank, local_rank, backend, single_device = init_ranks(args)
    pkg = importlib.import_module("monocular-visual-odometry_amd")
    ctx = pkg.get_context(local_rank)
    n, D = 1_000_000, 10
    rng = np.random.default_rng(7)
    app2 = rng.uniform(-1.0, 1.0, (n, D))
    perm = rng.permutation(n)
    app1 = app2[perm]
    lo, hi = pkg.sharded.row_bounds(n, rank, world)
    d1, d2 = ctx.to_device(app1[lo:hi]), ctx.to_device(app2)

    def associate():
        if world == 1:
            return ctx.associate(d1, d2)
        return pkg.sharded.associate_sharded(ctx.associate, d1, d2, lo)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    timing = {}
    settle = None
    for mode in ("scan", "hash"):
        ctx.set_association_mode(mode)
        reps = args.steps if mode == "hash" else max(2, min(args.steps, 4))
        if mode == "hash" and world == 1:
            settle = settle_device(associate)
        for _ in range(max(args.warmup, 1) if mode == "hash" else 1):
            i1, i2 = associate()
        fence()
        t0 = time.perf_counter()
        for _ in range(reps):
            i1, i2 = associate()
        fence()
        timing[mode] = max_over_ranks(time.perf_counter() - t0, world, backend) / reps
        assert np.array_equal(i2.cpu().numpy(), perm) and np.array_equal(i1.cpu().numpy(), np.arange(n))
    ctx.set_association_mode("auto")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    pairs = float(n) * n
    dt = timing["hash"]                                           # what picp_associate runs at this size
    sqs, sqs_src = committed_counters("r*_pmc_sq_assoc*.json", "assoc_scan_kernel")
    if sqs and "SQ_INSTS_VALU" in sqs and world > 1:         # the counter pass ran all the query rows on one GPU
        sqs = {"SQ_INSTS_VALU": float(sqs["SQ_INSTS_VALU"]) * (hi - lo) / n}
    scan_valu = valu_block(sqs, sqs_src, timing["scan"] * 1e3)
    # the join reads both descriptor sets once and writes / probes an 8-byte-per-slot table, priced at 2 n slots
    # rounded up to a power of two (the library spreads it over 4 n: shorter probe sequences for more bytes, which
    # are its own affair), and writes the two index lists
    table_bytes = 8.0 * (1 << int(np.ceil(np.log2(2 * n))))
    join_bytes = 2 * n * D * 8.0 + 2 * table_bytes + 2 * n * 8.0
    # counter traffic of one join call: the kernels of the join, each at its per-launch figure from the committed
    # FETCH_SIZE / WRITE_SIZE passes of this command (one launch of each per call; the table clear runs behind the call)
    join_traffic = join_traffic_src = join_traffic_by_kernel = None
    if world == 1:
        parts = {}
        for kern in ("hash_place_kernel", "hash_settle_kernel", "hash_count_kernel", "scan_block_offsets_kernel",
                     "hash_emit_kernel", "hash_clear_kernel"):
            v, src = committed_counters("r*_pmc_traffic_assoc*.json", kern)
            if v:
                parts[kern] = v["hbm_bytes_per_launch"]
                join_traffic_src = src
        if parts:
            join_traffic, join_traffic_by_kernel = sum(parts.values()), parts
    scan_traffic = scan_traffic_src = None
    if world == 1:
        v, scan_traffic_src = committed_counters("r*_pmc_traffic_assoc*.json", "assoc_scan_kernel")
        scan_traffic = v["hbm_bytes_per_launch"] if v else None
    out = {"metric": "appearance-association pair evaluations/sec (N1 x N2 per call: the pairs the reference's double "
                     "loop visits, src/data_association.py:29-31)",
           "value": pairs / dt, "unit": "pair-evaluations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "strong" if world > 1 else None,
           "vs_baseline": None, "dtype": "f64",
           "data": "synthetic",
           "config": {"workload": f"appearance data association, {n} x {n} descriptors of {D} float64, every query has "
                                  "exactly one exact match, projection = 0", "method": "hash join (picp_associate's "
                                  "choice at this size; identical pair list to the all-pairs scan)",
                      "collective": None if world == 1 else f"query rows sharded over {world} ranks, set 2 replicated; one "
                                    f"all-gather of the pair counts and one of the index lists per call ({backend}); the hash "
                                    "join is build-bound -- every rank builds the whole table, only the probe side shrinks "
                                    "with the row split (one rank's work alone: 1.6x faster at an eighth of the rows) -- "
                                    "while the all-pairs scan (--method scan) divides with the rows (6.5x at an eighth)",
                      "all_pairs_scan_ms": timing["scan"] * 1e3, "all_pairs_scan_pair_evaluations_per_s": pairs / timing["scan"]},
           "roofline": {"bound": "hbm", "kernel": "hash_place_kernel + hash_settle_kernel + hash_count_kernel + offsets + hash_emit_kernel (hash join)",
                        "achieved": join_bytes / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": join_bytes / dt / 1e9 / HBM_PEAK_GBS, "traffic": join_traffic,
                        "traffic_source": join_traffic_src, "traffic_by_kernel": join_traffic_by_kernel,
                        "bytes_per_call": join_bytes,
                        "note": "algorithmic bytes of the join: both descriptor sets once, the table written and probed "
                                "once, two index lists; time = whole picp_associate call (five launches and one "
                                "wait), not one kernel",
                        "all_pairs_scan": {"bound": "int_valu", "valu": scan_valu,
                                           "instructions_per_pair": (scan_valu["valu_wave_instructions_per_launch"] * 64 / (float(hi - lo) * n))
                                           if scan_valu else 1.5,
                                           "note": "exact equality forbids the distance-GEMM form; one 32-bit screening "
                                                   "test per pair on the vector ALU (about 1.5 instructions per pair)"}}}
    if args.method == "scan":
        # the line for the all-pairs method: every one of the N1 x N2 pairs really is evaluated
        ds = timing["scan"]
        peak = SIMDS * 16 * CLOCK_HZ / 1e12
        achieved = (scan_valu["valu_wave_instructions_per_launch"] * 64 / ds / 1e12) if scan_valu else None
        out["value"], out["ms_per_step"] = pairs / ds, ds * 1e3
        out["steps"] = max(2, min(args.steps, 4))
        out["config"]["method"] = "all-pairs scan (picp_set_association_mode 1): the reference's double loop, one 32-bit screening test per pair"
        out["config"]["hash_join_ms"] = dt * 1e3
        out["roofline"] = {"bound": "int_valu", "kernel": "assoc_scan_kernel<8> (keys as scalar operands, xor / min3 screening)",
                           "achieved": achieved, "peak": peak, "unit": "1e12 vector lane-operations/s",
                           "frac": (achieved / peak) if achieved else None,
                           "peak_derivation": "1 024 SIMDs x 16 lanes per clock x 2.4 GHz (a 64-lane vector instruction "
                                              "occupies its SIMD for 4 cycles)",
                           "valu": scan_valu, "traffic": scan_traffic, "traffic_source": scan_traffic_src,
                           "instructions_per_pair": out["roofline"]["all_pairs_scan"]["instructions_per_pair"],
                           "note": "exact equality forbids the distance-GEMM form; about 1.5 vector instructions per pair"}
    if settle:
        out["settle"] = settle
    if not args.no_cpu_baseline and world == 1:
        q = 512
        t0 = time.perf_counter()
        r1, r2 = po.associate(app1[:q], app2, chunk=64)
        dtc = time.perf_counter() - t0
        assert np.array_equal(r2, perm[:q])
        out["cpu_baseline"] = {"value": q * float(n) / dtc, "unit": "pair-evaluations/s", "cores": 1, "kind": "port",
                               "sample": f"vectorised numpy restatement of the reference's all-pairs equality test, the "
                                         f"first {q} query rows against all {n} rows, {dtc:.1f} s on 1 of {os.cpu_count()} "
                                         "host CPUs (the reference's own double loop of Python list comparisons is "
                                         "orders of magnitude slower)"}
    print(json.dumps(out), flush=True)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: this process starts the N ranks itself (one
    process per GPU under torch.distributed.run, rendezvous on 127.0.0.1) and hands their exit status on.
    It never touches the GPU, nor imports torch: the children are started before anything could initialise
    a device here, and nothing is exec'ed.  Rank 0's JSON line reaches stdout through the inherited pipe."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL and the peer exchange need here
    env.setdefault("OMP_NUM_THREADS", "4")
    print(f"bench.py: starting {args.gpus} ranks: {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    if args.config == "batched":
        return bench_batched(args)
    if args.config == "assoc":
        return bench_assoc(args)
    return bench_loop(args)


if __name__ == "__main__":
    main()
