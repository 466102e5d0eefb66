"""Error behaviour of the C ABI (include/picp_hip.h): a call with unusable arguments returns PICP_ERR_INVALID (or
PICP_ERR_CAPACITY) and leaves a message for picp_last_error(), touches nothing it was given, and the context goes on
working.  The reference raises Python exceptions for the same mistakes (a missing pose row makes `@` fail, an empty
measurement raises in np.mean); a C boundary reports codes instead."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch

from tests.synthetic import make_problem, true_pose

pytestmark = pytest.mark.gpu

pkg = importlib.import_module("monocular-visual-odometry_amd")
L = pkg._lib
INVALID, CAPACITY = -1, -3


@pytest.fixture(scope="module")
def env():
    ctx = pkg.get_context(0)
    cam = L.camera_struct([[180, 0, 320], [0, 180, 240], [0, 0, 1]], 640, 480)
    sched = L.schedule_struct(10, 10, 1000, 10, 2000, 64, 1, 64)
    uv, pw, _ = make_problem(5000, seed=[3, 0])
    corr = ctx.soa(uv, pw)
    return ctx, cam, sched, corr


def message(ctx):
    return ctx.lib.picp_last_error().decode()


def run_args(ctx, cam, sched, corr, pose, state, hist, rows, res, M=None):
    return [ctx.handle, *[c.data_ptr() for c in corr], corr[0].numel() if M is None else M, pose.ctypes.data, C.byref(cam),
            C.byref(sched), state.data_ptr(), hist.data_ptr() if hist is not None else 0, rows, C.byref(res)]


def test_loop_calls_reject_unusable_arguments(env):
    ctx, cam, sched, corr = env
    lib = ctx.lib
    pose = np.ascontiguousarray(true_pose().reshape(16))
    state = ctx.empty(L.STATE_DOUBLES)
    hist = ctx.zeros((10, L.HIST_STRIDE))
    res = L.ResultStruct()
    good = run_args(ctx, cam, sched, corr, pose, state, hist, 10, res)
    assert lib.picp_run(*good) == L.PICP_OK and res.num_iterations > 0
    reference = (np.array(list(res.T)), res.num_iterations, res.best_index)

    def expect_invalid(args, needle):
        before = state.clone()
        assert lib.picp_run(*args) == INVALID
        assert needle in message(ctx), message(ctx)
        assert torch.equal(before, state)                       # nothing was launched

    bad = list(good); bad[0] = None
    assert lib.picp_run(*bad) == INVALID                        # null context
    bad = list(good); bad[6] = -1
    expect_invalid(bad, "negative correspondence count")
    bad = list(good); bad[3] = 0
    expect_invalid(bad, "null correspondence array")
    bad = list(good); bad[7] = 0
    expect_invalid(bad, "null pose")
    broken = pose.copy(); broken[15] = 2.0                      # last row of the pose is not 0 0 0 1
    bad = list(good); bad[7] = broken.ctypes.data
    expect_invalid(bad, "(0,0,0,1)")
    bad = list(good); bad[12] = 5                               # fewer history rows than iterations
    expect_invalid(bad, "hist")
    negative = L.schedule_struct(-1, 10, 1000, 10, 2000, 64, 1, 64)
    bad = list(good); bad[9] = C.byref(negative)
    expect_invalid(bad, "num_iterations")

    # the asynchronous form: slots and sizes
    enq = good[:-1] + [7]
    assert lib.picp_run_enqueue(*enq) == INVALID and "slot" in message(ctx)
    assert lib.picp_run_collect(ctx.handle, -1, pose.ctypes.data, C.byref(res)) == INVALID
    zero = L.schedule_struct(0, 10, 1000, 10, 2000, 64, 1, 64)
    enq = list(good[:-1]) + [0]; enq[9] = C.byref(zero)
    assert lib.picp_run_enqueue(*enq) == INVALID and "positive" in message(ctx)

    # the sharded single launch without an exchange between GPUs
    assert lib.picp_run_sharded(*good) == INVALID and "exchange" in message(ctx)

    # settings
    assert lib.picp_set_resident_limit(ctx.handle, -5) == INVALID
    assert lib.picp_set_stage2_order(ctx.handle, 3) == INVALID
    assert lib.picp_set_association_mode(ctx.handle, 9) == INVALID
    assert lib.picp_set_grid_limit(ctx.handle, -1) == INVALID

    # the context still works, and gives what it gave before
    assert lib.picp_run(*good) == L.PICP_OK
    assert np.array_equal(np.array(list(res.T)), reference[0]) and (res.num_iterations, res.best_index) == reference[1:]


def test_empty_input_runs_and_reports_an_invalid_solve(env):
    """No correspondences: the reference's loop runs its iterations on empty lists (error inf every time) and
    declares the solve invalid (src/ProjectiveICP.py:249-257); so does the library, without a launch failing."""
    ctx, cam, sched, corr = env
    pose = np.ascontiguousarray(true_pose().reshape(16))
    state = ctx.empty(L.STATE_DOUBLES)
    res = L.ResultStruct()
    args = run_args(ctx, cam, sched, [c[:0] for c in corr], pose, state, None, 0, res, M=0)
    args[1:6] = [0] * 5                                         # null arrays are fine when M = 0
    assert ctx.lib.picp_run(*args) == L.PICP_OK
    assert res.valid == 0 and np.array_equal(np.array(list(res.T)), pose)


def test_association_rejects_unusable_arguments_and_reports_capacity(env):
    ctx, _, _, _ = env
    lib = ctx.lib
    rng = np.random.default_rng(5)
    app2 = rng.standard_normal((300, 10))
    app1 = app2[rng.permutation(300)[:200]].copy()
    d1, d2 = ctx.to_device(app1), ctx.to_device(app2)
    idx1 = torch.full((256,), -7, dtype=torch.int64, device=d1.device)
    idx2 = torch.full((256,), -7, dtype=torch.int64, device=d1.device)
    count = C.c_int64(-1)
    good = [ctx.handle, d1.data_ptr(), 200, d2.data_ptr(), 300, 10, idx1.data_ptr(), idx2.data_ptr(), 256, C.byref(count)]
    assert lib.picp_associate(*good) == L.PICP_OK and count.value == 200
    pairs = (idx1[:200].clone(), idx2[:200].clone())

    for pos, value, needle in [(2, -1, "bad sizes"), (4, -3, "bad sizes"), (5, 0, "bad sizes"), (8, -1, "bad sizes"),
                               (1, 0, "null descriptor"), (3, 0, "null descriptor"), (6, 0, "null index"),
                               (9, None, "null argument")]:
        bad = list(good); bad[pos] = value
        assert lib.picp_associate(*bad) == INVALID, (pos, value)
        assert needle in message(ctx), message(ctx)

    # too small an output: the needed size is reported, nothing is written past the capacity
    for mode in (1, 2):                                         # all pairs, hash join
        assert lib.picp_set_association_mode(ctx.handle, mode) == L.PICP_OK
        idx1.fill_(-7); idx2.fill_(-7)
        small = list(good); small[8] = 50
        assert lib.picp_associate(*small) == CAPACITY
        assert "200 pairs" in message(ctx)
        assert bool((idx1[50:] == -7).all()) and bool((idx2[50:] == -7).all())
        assert lib.picp_associate(*good) == L.PICP_OK and count.value == 200
        assert torch.equal(idx1[:200], pairs[0]) and torch.equal(idx2[:200], pairs[1])
    assert lib.picp_set_association_mode(ctx.handle, 0) == L.PICP_OK

    # empty sets are not an error (the reference's double loop simply finds nothing, src/data_association.py:29-31)
    empty = list(good); empty[2] = 0; empty[1] = 0
    assert lib.picp_associate(*empty) == L.PICP_OK and count.value == 0
    empty = list(good); empty[4] = 0; empty[3] = 0
    assert lib.picp_associate(*empty) == L.PICP_OK and count.value == 0


def test_batched_and_linearize_reject_unusable_arguments(env):
    ctx, cam, sched, corr = env
    lib = ctx.lib
    pose = np.ascontiguousarray(true_pose().reshape(16))
    H, b = np.zeros(36), np.zeros(6)
    n, err, nin = C.c_int64(), C.c_double(), C.c_int64()
    good = [ctx.handle, *[c.data_ptr() for c in corr], corr[0].numel(), pose.ctypes.data, C.byref(cam), 1000.0, 10,
            H.ctypes.data, b.ctypes.data, C.byref(n), C.byref(err), C.byref(nin)]
    assert lib.picp_linearize(*good) == L.PICP_OK and n.value > 0
    bad = list(good); bad[11] = 0
    assert lib.picp_linearize(*bad) == INVALID and "null argument" in message(ctx)
    bad = list(good); bad[6] = -2
    assert lib.picp_linearize(*bad) == INVALID
    broken = pose.copy(); broken[12] = 1.0
    bad = list(good); bad[7] = broken.ctypes.data
    assert lib.picp_linearize(*bad) == INVALID and "(0,0,0,1)" in message(ctx)

    offsets = torch.tensor([0, 2000, 5000], dtype=torch.int64, device=corr[0].device)
    poses = ctx.to_device(np.stack([true_pose().reshape(16)] * 2))
    results = torch.zeros(2 * C.sizeof(L.ResultStruct), dtype=torch.uint8, device=corr[0].device)
    args = [ctx.handle, *[c.data_ptr() for c in corr], offsets.data_ptr(), 2, poses.data_ptr(), C.byref(cam),
            C.byref(sched), results.data_ptr(), 0, 0, 4096]
    assert lib.picp_run_batched(*args) == L.PICP_OK
    bad = list(args); bad[7] = -1
    assert lib.picp_run_batched(*bad) == INVALID and "negative" in message(ctx)
    bad = list(args); bad[14] = -1
    assert lib.picp_run_batched(*bad) == INVALID and "negative" in message(ctx)
    bad = list(args); bad[6] = 0
    assert lib.picp_run_batched(*bad) == INVALID and "null argument" in message(ctx)
    bad = list(args); bad[11] = 0
    assert lib.picp_run_batched(*bad) == INVALID and "null argument" in message(ctx)


def test_projection_gather_and_exchange_calls_reject_misuse(env):
    ctx, cam, _, corr = env
    lib = ctx.lib
    N = 100
    inside = torch.zeros(N, dtype=torch.int32, device=corr[0].device)
    uu, vv = ctx.empty(N), ctx.empty(N)
    c_T_w = np.ascontiguousarray(np.linalg.inv(true_pose()).reshape(16))
    good = [ctx.handle, corr[2].data_ptr(), corr[3].data_ptr(), corr[4].data_ptr(), N, c_T_w.ctypes.data, C.byref(cam),
            inside.data_ptr(), uu.data_ptr(), vv.data_ptr()]
    assert lib.picp_project(*good) == L.PICP_OK
    bad = list(good); bad[4] = -1
    assert lib.picp_project(*bad) == INVALID and "negative" in message(ctx)
    bad = list(good); bad[7] = 0
    assert lib.picp_project(*bad) == INVALID and "null array" in message(ctx)
    broken = c_T_w.copy(); broken[13] = 0.5
    bad = list(good); bad[5] = broken.ctypes.data
    assert lib.picp_project(*bad) == INVALID

    idx = torch.arange(N, dtype=torch.int64, device=corr[0].device)
    pos1 = ctx.to_device(np.zeros((N, 2)))
    pos2 = ctx.to_device(np.zeros((N, 3)))
    out = [ctx.empty(N) for _ in range(5)]
    good = [ctx.handle, idx.data_ptr(), idx.data_ptr(), N, pos1.data_ptr(), pos2.data_ptr(), *[o.data_ptr() for o in out]]
    assert lib.picp_gather_pairs(*good) == L.PICP_OK
    bad = list(good); bad[3] = -4
    assert lib.picp_gather_pairs(*bad) == INVALID and "negative" in message(ctx)
    bad = list(good); bad[5] = 0
    assert lib.picp_gather_pairs(*bad) == INVALID and "null array" in message(ctx)

    # the exchanges between GPUs, used out of order
    handles = C.create_string_buffer(L.PEER_HANDLE_BYTES * 2)
    assert lib.picp_peer_connect(ctx.handle, handles) == INVALID and "picp_peer_create first" in message(ctx)
    one = C.create_string_buffer(L.PEER_HANDLE_BYTES)
    assert lib.picp_peer_create(ctx.handle, 3, 2, one) == INVALID and "rank 3 of 2" in message(ctx)
    assert lib.picp_peer_create(ctx.handle, 0, L.PEER_MAX_RANKS + 1, one) == INVALID
    assert lib.picp_peer_world(ctx.handle) == 0
    buf = ctx.zeros(4)
    assert lib.picp_comm_allreduce(ctx.handle, buf.data_ptr(), 4) == INVALID and "no communicator" in message(ctx)
    assert lib.picp_comm_init(ctx.handle, None, 0, 1) == INVALID
    small = C.create_string_buffer(8)
    assert lib.picp_comm_unique_id(small, 8) == INVALID and "bytes" in message(ctx)
    bus = C.create_string_buffer(4)
    assert lib.picp_device_bus_id(ctx.handle, bus, 4) == INVALID
    assert lib.picp_peer_destroy(ctx.handle) == L.PICP_OK            # nothing to destroy is not an error
