"""ctypes binding of libpicp_hip.so (C ABI declared in include/picp_hip.h).

There is no CPU fallback: importing this module without the built library, or creating a
context without a gfx950 GPU, raises.  PyTorch is used only to own device memory and to name
the HIP stream work is enqueued on.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PICP_LIB selects another build of the same library (A/B timing of kernel variants on one box).
LIB_PATH = os.environ.get("PICP_LIB") or os.path.join(_HERE, "csrc", "libpicp_hip.so")

HIST_STRIDE = 80
STATE_DOUBLES = 96
S1_DOUBLES = 4
S2_DOUBLES = 32

# history row layout (include/picp_hip.h)
H_T, H_ERROR, H_NUM_INLIERS, H_KERNEL_THR, H_DAMPING = 0, 16, 17, 18, 19
H_N_INSIDE, H_DONE, H_N_STAGE1, H_N_STAGE2, H_USED_STAGE2, H_MEAN_STAGE1 = 20, 21, 22, 23, 24, 25
H_DX, H_B, H_H = 26, 32, 38
# state block layout (csrc/picp_device.h)
ST_W_T_C, ST_C_T_W, ST_THR, ST_LAMBDA, ST_IT, ST_STOP = 0, 16, 32, 33, 39, 40
ST_BEST_ERR, ST_BEST_IT, ST_BEST_T = 51, 52, 64

LINEARIZE_OUT_DOUBLES = 48
NOALLOC_MAX_CORRESPONDENCES = 1 << 32

PICP_OK = 0
PICP_ERR_CAPACITY = -3
PICP_ERR_COMM = -5
PICP_ERR_RETRY = -6
FULL_FORM_CALLS = 8    # PICP_FULL_FORM_CALLS
ASYNC_SLOTS = 4
COMM_ID_BYTES = 128
PEER_MAX_RANKS = 8
PEER_HANDLE_BYTES = 64
BUS_ID_BYTES = 64
PEER_MEMORY_KINDS = {0: "none", 1: "uncached device memory", 2: "fine-grained device memory"}


class PicpError(RuntimeError):
    pass


class PicpRetry(PicpError):
    """A single-launch loop gave up at a bounded wait (PICP_ERR_RETRY): nothing was computed; run the
    loop again on another path."""


class CameraStruct(C.Structure):
    _fields_ = [("K", C.c_double * 9), ("width", C.c_double), ("height", C.c_double)]


class ScheduleStruct(C.Structure):
    _fields_ = [("num_iterations", C.c_int32), ("min_inliers", C.c_int32),
                ("kernel_threshold", C.c_double), ("min_kernel_threshold", C.c_double),
                ("max_kernel_threshold", C.c_double), ("damping", C.c_double),
                ("min_damping", C.c_double), ("max_damping", C.c_double),
                ("disable_stop", C.c_int32), ("reserved", C.c_int32)]


class ResultStruct(C.Structure):
    _fields_ = [("T", C.c_double * 16), ("valid", C.c_int32), ("num_iterations", C.c_int32),
                ("best_index", C.c_int32), ("reserved", C.c_int32), ("best_error", C.c_double)]


RESULT_DOUBLES = C.sizeof(ResultStruct) // 8
assert C.sizeof(ResultStruct) == 8 * 19

_P = C.c_void_p
_SIGNATURES = {
    "picp_version": (C.c_int, []),
    "picp_last_error": (C.c_char_p, []),
    "picp_create": (C.c_int, [C.POINTER(_P), C.c_int, _P]),
    "picp_destroy": (C.c_int, [_P]),
    "picp_set_stream": (C.c_int, [_P, _P]),
    "picp_synchronize": (C.c_int, [_P]),
    "picp_run": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, _P, C.POINTER(CameraStruct),
                           C.POINTER(ScheduleStruct), _P, _P, C.c_int32, C.POINTER(ResultStruct)]),
    "picp_run_enqueue": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, _P, C.POINTER(CameraStruct),
                                   C.POINTER(ScheduleStruct), _P, _P, C.c_int32, C.c_int32]),
    "picp_run_collect": (C.c_int, [_P, C.c_int32, _P, C.POINTER(ResultStruct)]),
    "picp_set_resident_limit": (C.c_int, [_P, C.c_int64]),
    "picp_resident_capacity": (C.c_int64, [_P]),
    "picp_streaming_held": (C.c_int64, [_P, C.c_int64]),
    "picp_resident_aborts": (C.c_int, [_P]),
    "picp_immoderate_exits": (C.c_int, [_P]),
    "picp_profile_run": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, _P, C.POINTER(CameraStruct),
                                   C.POINTER(ScheduleStruct), _P, C.c_int32, C.POINTER(C.c_double),
                                   C.POINTER(C.c_double)]),
    "picp_run_batched": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int32, _P, C.POINTER(CameraStruct),
                                   C.POINTER(ScheduleStruct), _P, _P, C.c_int32, C.c_int64]),
    "picp_state_init": (C.c_int, [_P, _P, _P, C.POINTER(ScheduleStruct)]),
    "picp_stage1": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, _P, C.POINTER(CameraStruct)]),
    "picp_stage1_reduce": (C.c_int, [_P, _P]),
    "picp_stage2": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, _P, C.POINTER(CameraStruct), _P]),
    "picp_stage2_reduce": (C.c_int, [_P, _P]),
    "picp_set_stage2_order": (C.c_int, [_P, C.c_int32]),
    "picp_profile_stage2": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, _P, C.POINTER(CameraStruct), C.c_int32,
                                      C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "picp_solve": (C.c_int, [_P, _P, C.POINTER(ScheduleStruct), _P, _P, _P, C.c_int32]),
    "picp_comm_unique_id": (C.c_int, [_P, C.c_int64]),
    "picp_comm_init": (C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    "picp_comm_destroy": (C.c_int, [_P]),
    "picp_comm_world": (C.c_int, [_P]),
    "picp_comm_allreduce": (C.c_int, [_P, _P, C.c_int64]),
    "picp_sharded_rounds": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, _P, C.POINTER(CameraStruct),
                                      C.POINTER(ScheduleStruct), _P, _P, _P, C.c_int32, C.c_int32]),
    "picp_peer_create": (C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    "picp_peer_connect": (C.c_int, [_P, _P]),
    "picp_peer_destroy": (C.c_int, [_P]),
    "picp_peer_world": (C.c_int, [_P]),
    "picp_device_bus_id": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "picp_peer_can_access": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int32)]),
    "picp_peer_memory_kind": (C.c_int, [_P]),
    "picp_comm_available": (C.c_int, []),
    "picp_run_sharded": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, _P, C.POINTER(CameraStruct),
                                   C.POINTER(ScheduleStruct), _P, _P, C.c_int32, C.POINTER(ResultStruct)]),
    "picp_set_grid_limit": (C.c_int, [_P, C.c_int32]),
    "picp_set_launch_timing": (C.c_int, [_P, C.c_int32]),
    "picp_last_launch_ms": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "picp_finish": (C.c_int, [_P, _P, _P, C.c_int32, _P, C.POINTER(ResultStruct)]),
    "picp_linearize": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, _P, C.POINTER(CameraStruct), C.c_double,
                                 C.c_int32, _P, _P, C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                 C.POINTER(C.c_int64)]),
    "picp_linearize_noalloc": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, _P, C.POINTER(CameraStruct), C.c_double,
                                         C.c_int32, _P]),
    "picp_associate": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, C.c_int32, _P, _P, C.c_int64,
                                 C.POINTER(C.c_int64)]),
    "picp_set_association_mode": (C.c_int, [_P, C.c_int32]),
    "picp_association_fallbacks": (C.c_int, [_P]),
    "picp_project": (C.c_int, [_P, _P, _P, _P, C.c_int64, _P, C.POINTER(CameraStruct), _P, _P, _P]),
    "picp_gather_pairs": (C.c_int, [_P, _P, _P, C.c_int64, _P, _P, _P, _P, _P, _P, _P]),
    "picp_invert_pose": (C.c_int, [_P, _P, _P]),
    "picp_v2T": (C.c_int, [_P, _P, _P]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load_library():
    """Loads libpicp_hip.so and declares every entry point's signature."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PicpError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C monocular-visual-odometry_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != PICP_OK:
        msg = load_library().picp_last_error()
        raise PicpError(f"{what} failed with status {rc}: {msg.decode() if msg else ''}")


def camera_struct(K, width, height) -> CameraStruct:
    c = CameraStruct()
    Kf = np.asarray(K, dtype=np.float64).reshape(9)
    for i in range(9):
        c.K[i] = float(Kf[i])
    c.width = float(width)
    c.height = float(height)
    return c


def schedule_struct(num_iterations, min_inliers, kernel_threshold, min_kernel_threshold, max_kernel_threshold,
                    damping, min_damping, max_damping, disable_stop=False) -> ScheduleStruct:
    s = ScheduleStruct()
    s.num_iterations = int(num_iterations)
    s.min_inliers = int(min_inliers)
    s.kernel_threshold = float(kernel_threshold)
    s.min_kernel_threshold = float(min_kernel_threshold)
    s.max_kernel_threshold = float(max_kernel_threshold)
    s.damping = float(damping)
    s.min_damping = float(min_damping)
    s.max_damping = float(max_damping)
    s.disable_stop = 1 if disable_stop else 0
    s.reserved = 0
    return s


def _pad32(n):
    """Plane stride (in doubles) that keeps every SoA plane 256-byte aligned, so the scans use 16-byte loads."""
    return max((int(n) + 31) // 32 * 32, 32)


def _host16(T):
    a = np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(16))
    return a, a.ctypes.data_as(_P)


class Context:
    """One picp_ctx: a device, the torch stream current at creation, and the reduction workspace."""

    def __init__(self, device=0):
        import torch
        if not torch.cuda.is_available():
            raise PicpError("no HIP device visible to PyTorch; the projective-ICP kernels need an MI355X (gfx950)")
        self.torch = torch
        self.lib = load_library()
        self.device = torch.device("cuda", int(device) if not isinstance(device, torch.device) else device.index or 0)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            handle = _P()
            check(self.lib.picp_create(C.byref(handle), self.device.index, _P(stream)), "picp_create")
        self.handle = handle
        self._run_state = None

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.picp_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ---- plumbing ----------------------------------------------------------------------------
    def to_device(self, a, dtype=np.float64):
        """Host array -> contiguous device tensor (float64 unless told otherwise)."""
        t = self.torch
        if isinstance(a, t.Tensor):
            want = {np.float64: t.float64, np.int64: t.int64, np.uint8: t.uint8}[dtype]
            return a.to(device=self.device, dtype=want).contiguous()
        return t.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=dtype))).to(self.device)

    def empty(self, shape, dtype="float64"):
        t = self.torch
        return t.empty(shape, dtype=getattr(t, dtype), device=self.device)

    def zeros(self, shape, dtype="float64"):
        t = self.torch
        return t.zeros(shape, dtype=getattr(t, dtype), device=self.device)

    @staticmethod
    def ptr(tensor):
        return _P(tensor.data_ptr()) if tensor is not None else _P(0)

    def use_current_stream(self):
        stream = self.torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.picp_set_stream(self.handle, _P(stream)), "picp_set_stream")

    def synchronize(self):
        check(self.lib.picp_synchronize(self.handle), "picp_synchronize")

    def soa(self, uv, pw):
        """(M,2) pixels + (M,3) points -> five contiguous device arrays u, v, x, y, z."""
        uv = np.asarray(uv, dtype=np.float64).reshape(-1, 2)
        pw = np.asarray(pw, dtype=np.float64).reshape(-1, 3)
        if uv.shape[0] != pw.shape[0]:
            raise ValueError(f"{uv.shape[0]} image points but {pw.shape[0]} world points")
        M = uv.shape[0]
        planes = np.zeros((5, _pad32(M)))                       # plane stride padded: every plane 256-B aligned
        planes[:2, :M] = uv.T
        planes[2:, :M] = pw.T
        d = self.to_device(planes)
        return [d[k][:M] for k in range(5)]

    # ---- loop ----------------------------------------------------------------------------------
    def set_resident_limit(self, max_correspondences=None):
        """Largest input for which picp_run uses its single-launch, register-resident loop
        (None: whatever fits the device; 0: always the streaming kernels)."""
        n = (1 << 62) if max_correspondences is None else int(max_correspondences)
        check(self.lib.picp_set_resident_limit(self.handle, n), "picp_set_resident_limit")

    def set_stage2_order(self, order="auto"):
        """Direction of the second scan of the streaming path: "forward" (as the first scan), "reverse"
        (against it: reuse of the memory-side cache across the two scans) or "auto" (by input size)."""
        check(self.lib.picp_set_stage2_order(self.handle, {"auto": -1, "forward": 0, "reverse": 1}[order]),
              "picp_set_stage2_order")

    def resident_capacity(self):
        return int(self.lib.picp_resident_capacity(self.handle))

    def streaming_held(self, M):
        """Correspondences of an M-correspondence input that the streaming single launch keeps on chip."""
        return int(self.lib.picp_streaming_held(self.handle, int(M)))

    def resident_aborts(self):
        """Single-launch loops that gave up at a grid-wide wait and were repeated with the streaming kernels."""
        return int(self.lib.picp_resident_aborts(self.handle))

    def immoderate_exits(self):
        """On-chip loops that met a value that is not finite and of moderate magnitude and were run again in the
        form of the kernel that assumes nothing about magnitudes."""
        return int(self.lib.picp_immoderate_exits(self.handle))

    def run(self, corr, w_T_c0, cam: CameraStruct, sched: ScheduleStruct, want_history=True):
        """picp_run on device arrays corr = [u, v, x, y, z].  Returns (ResultStruct, history ndarray)."""
        # The call is synchronous and the state block is scratch, so one block per context serves
        # every call; pointers go in as plain integers (this path is timed in microseconds).
        state = self._run_state
        if state is None:
            state = self._run_state = self.empty(STATE_DOUBLES)
        rows = max(int(sched.num_iterations), 1)
        hist = self.zeros((rows, HIST_STRIDE)) if want_history else None
        res = ResultStruct()
        pose = np.ascontiguousarray(w_T_c0, dtype=np.float64).reshape(16)
        rc = self.lib.picp_run(self.handle, corr[0].data_ptr(), corr[1].data_ptr(), corr[2].data_ptr(),
                               corr[3].data_ptr(), corr[4].data_ptr(), corr[0].numel(), pose.ctypes.data, C.byref(cam),
                               C.byref(sched), state.data_ptr(), hist.data_ptr() if want_history else 0,
                               rows if want_history else 0, C.byref(res))
        if rc != PICP_OK:
            check(rc, "picp_run")
        h = hist[: res.num_iterations].cpu().numpy() if want_history else None
        return res, h

    def set_grid_limit(self, max_workgroups=0):
        """Cap on the workgroups of the single-launch loops (0: one per compute unit).  For rehearsals of a
        sharded run with several ranks on one GPU, whose kernels must all be resident together."""
        check(self.lib.picp_set_grid_limit(self.handle, int(max_workgroups)), "picp_set_grid_limit")

    def peer_world(self):
        """Ranks of the ctx's usable in-kernel exchange between GPUs (0: none)."""
        return int(self.lib.picp_peer_world(self.handle))

    def run_sharded(self, corr, w_T_c0, cam: CameraStruct, sched: ScheduleStruct, want_history=True):
        """picp_run_sharded on this rank's slice: the single-launch loop with the exchange between the GPUs
        inside the kernel (sharded.init_peer_exchange first; every rank calls this with the same pose and
        schedule).  Raises PicpRetry when the launch gave up at an exchange."""
        state = self._run_state
        if state is None:
            state = self._run_state = self.empty(STATE_DOUBLES)
        rows = max(int(sched.num_iterations), 1)
        hist = self.zeros((rows, HIST_STRIDE)) if want_history else None
        res = ResultStruct()
        pose = np.ascontiguousarray(w_T_c0, dtype=np.float64).reshape(16)
        rc = self.lib.picp_run_sharded(self.handle, corr[0].data_ptr(), corr[1].data_ptr(), corr[2].data_ptr(),
                                       corr[3].data_ptr(), corr[4].data_ptr(), corr[0].numel(), pose.ctypes.data,
                                       C.byref(cam), C.byref(sched), state.data_ptr(),
                                       hist.data_ptr() if want_history else 0, rows if want_history else 0, C.byref(res))
        if rc == PICP_ERR_RETRY:
            raise PicpRetry(self.lib.picp_last_error().decode())
        if rc != PICP_OK:
            check(rc, "picp_run_sharded")
        h = hist[: res.num_iterations].cpu().numpy() if want_history else None
        return res, h

    def plan(self, corr, cam: CameraStruct, sched: ScheduleStruct, want_history=False):
        """picp_run on fixed device arrays with the call's arguments marshalled once: for callers that
        run many loops over the same correspondences (different start poses), where the few microseconds
        of per-call argument conversion are visible next to a loop of a few hundred microseconds."""
        return LoopPlan(self, corr, cam, sched, want_history)

    def run_batched(self, corr, offsets, poses0, cam: CameraStruct, sched: ScheduleStruct, want_history=False,
                    max_count=None):
        """picp_run_batched.  offsets: int64 device tensor [B+1]; poses0: device [B,16].
        max_count: upper bound on the problem sizes (None: read back from ``offsets``, one small
        synchronising copy); problems of up to 4 096 correspondences then stay on chip for the whole loop.
        Returns (results tensor [B, RESULT_DOUBLES] viewed as float64, history tensor or None); does not sync."""
        B = int(offsets.numel()) - 1
        results = self.empty((B, RESULT_DOUBLES))
        rows = max(int(sched.num_iterations), 1)
        hist = self.zeros((B, rows, HIST_STRIDE)) if want_history else None
        if max_count is None:
            max_count = int((offsets[1:] - offsets[:-1]).max().item()) if B > 0 else 0
        check(self.lib.picp_run_batched(self.handle, *[self.ptr(c) for c in corr], self.ptr(offsets), B,
                                        self.ptr(poses0), C.byref(cam), C.byref(sched), self.ptr(results),
                                        self.ptr(hist), rows if want_history else 0, int(max_count)), "picp_run_batched")
        return results, hist

    @staticmethod
    def decode_results(results_host: np.ndarray):
        """[B, RESULT_DOUBLES] float64 host array -> dict of arrays (T, valid, num_iterations, best_index, best_error)."""
        raw = np.ascontiguousarray(results_host)
        ints = raw[:, 16:18].copy().view(np.int32).reshape(-1, 4)
        return {"T": raw[:, :16].reshape(-1, 4, 4).copy(), "valid": ints[:, 0].astype(bool),
                "num_iterations": ints[:, 1].copy(), "best_index": ints[:, 2].copy(), "best_error": raw[:, 18].copy()}

    def linearize(self, corr, w_T_c0, cam: CameraStruct, kernel_threshold, min_inliers):
        M = int(corr[0].numel())
        H = np.zeros(36)
        b = np.zeros(6)
        n, err, nin = C.c_int64(), C.c_double(), C.c_int64()
        pose, pose_p = _host16(w_T_c0)
        check(self.lib.picp_linearize(self.handle, *[self.ptr(c) for c in corr], M, pose_p, C.byref(cam),
                                      float(kernel_threshold), int(min_inliers), H.ctypes.data_as(_P),
                                      b.ctypes.data_as(_P), C.byref(n), C.byref(err), C.byref(nin)), "picp_linearize")
        return H.reshape(6, 6), b, int(n.value), float(err.value), int(nin.value)

    def linearize_noalloc(self, corr, w_T_c0, cam: CameraStruct, kernel_threshold, min_inliers, out):
        """picp_linearize_noalloc: the same computation with no allocation inside the call; the results go
        to ``out`` (device tensor of LINEARIZE_OUT_DOUBLES float64: H, b, num_inliers, error, n_inside).
        Only enqueues; all calls on one GPU share one static workspace and must not overlap."""
        pose, pose_p = _host16(w_T_c0)
        check(self.lib.picp_linearize_noalloc(self.handle, *[self.ptr(c) for c in corr], int(corr[0].numel()), pose_p,
                                              C.byref(cam), float(kernel_threshold), int(min_inliers), self.ptr(out)),
              "picp_linearize_noalloc")

    # ---- association ---------------------------------------------------------------------------
    def set_association_mode(self, mode="auto"):
        """"auto": hash join for large inputs, all-pairs scan otherwise; "scan"; "hash"."""
        check(self.lib.picp_set_association_mode(self.handle, {"auto": 0, "scan": 1, "hash": 2}[mode]),
              "picp_set_association_mode")

    def association_fallbacks(self):
        return int(self.lib.picp_association_fallbacks(self.handle))

    def associate(self, app1, app2, capacity=None):
        """Device descriptor tensors [N1,D], [N2,D] -> (idx1, idx2) int64 device tensors, query-major order."""
        N1, N2 = int(app1.shape[0]), int(app2.shape[0])
        if N1 == 0 or N2 == 0:
            e = self.empty(0, "int64")
            return e, e.clone()
        D = int(app1.shape[1])
        if int(app2.shape[1]) != D:
            raise ValueError(f"descriptor widths differ: {D} vs {int(app2.shape[1])}")
        cap = int(capacity) if capacity is not None else max(N1, 1)
        while True:
            idx1 = self.empty(cap, "int64")
            idx2 = self.empty(cap, "int64")
            count = C.c_int64()
            rc = self.lib.picp_associate(self.handle, self.ptr(app1), N1, self.ptr(app2), N2, D, self.ptr(idx1),
                                         self.ptr(idx2), cap, C.byref(count))
            if rc == PICP_ERR_CAPACITY:
                cap = int(count.value)
                continue
            check(rc, "picp_associate")
            n = int(count.value)
            return idx1[:n], idx2[:n]

    def gather_pairs(self, idx1, idx2, pos1, pos2):
        n = int(idx1.numel())
        out = self.empty((5, _pad32(n)))
        check(self.lib.picp_gather_pairs(self.handle, self.ptr(idx1), self.ptr(idx2), n, self.ptr(pos1), self.ptr(pos2),
                                         *[self.ptr(out[k]) for k in range(5)]), "picp_gather_pairs")
        return [out[k][:n] for k in range(5)]

    def project(self, xyz, c_T_w, cam: CameraStruct):
        """xyz: device [3, N] planes.  Returns (inside uint8 [N], u [N], v [N]) device tensors."""
        N = int(xyz.shape[1])
        inside = self.empty(max(N, 1), "uint8")
        uv = self.empty((2, max(N, 1)))
        pose, pose_p = _host16(c_T_w)
        check(self.lib.picp_project(self.handle, self.ptr(xyz[0]), self.ptr(xyz[1]), self.ptr(xyz[2]), N, pose_p,
                                    C.byref(cam), self.ptr(inside), self.ptr(uv[0]), self.ptr(uv[1])), "picp_project")
        return inside[:N], uv[0][:N], uv[1][:N]

    def invert_pose(self, T):
        a, ap = _host16(T)
        out = np.zeros(16)
        check(self.lib.picp_invert_pose(self.handle, ap, out.ctypes.data_as(_P)), "picp_invert_pose")
        return out.reshape(4, 4)

    def v2T(self, v):
        a = np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(6))
        out = np.zeros(16)
        check(self.lib.picp_v2T(self.handle, a.ctypes.data_as(_P), out.ctypes.data_as(_P)), "picp_v2T")
        return out.reshape(4, 4)


class LoopPlan:
    """See Context.plan.  Call it with a start pose; returns (ResultStruct, history ndarray or None).
    The ResultStruct is reused by the next call."""

    def __init__(self, ctx, corr, cam, sched, want_history):
        self.ctx = ctx
        self._keep = (list(corr), cam, sched)                       # the arrays and structs the pointers refer to
        self.state = ctx.empty(STATE_DOUBLES)
        self.rows = max(int(sched.num_iterations), 1)
        self.hist = ctx.zeros((self.rows, HIST_STRIDE)) if want_history else None
        self.result = ResultStruct()
        self.pose = np.zeros(16)
        self._args = (ctx.handle, *[_P(c.data_ptr()) for c in corr], C.c_int64(int(corr[0].numel())),
                      _P(self.pose.ctypes.data), C.byref(cam), C.byref(sched), _P(self.state.data_ptr()),
                      _P(self.hist.data_ptr()) if want_history else _P(0), C.c_int32(self.rows if want_history else 0),
                      C.byref(self.result))
        self._fn = ctx.lib.picp_run

    def __call__(self, w_T_c0):
        self.pose[:] = np.asarray(w_T_c0, dtype=np.float64).reshape(16)
        rc = self._fn(*self._args)
        if rc != PICP_OK:
            check(rc, "picp_run")
        hist = self.hist[: self.result.num_iterations].cpu().numpy() if self.hist is not None else None
        return self.result, hist

    # Several loops in flight: enqueue(pose, slot) only enqueues, collect(slot) waits for that loop.
    # Each slot has its own state block, start pose and result; no per-iteration history in this mode.
    def enqueue(self, w_T_c0, slot):
        if not hasattr(self, "_slots"):
            self._slots = {}
        if slot not in self._slots:
            corr, cam, sched = self._keep
            state, pose, res = self.ctx.empty(STATE_DOUBLES), np.zeros(16), ResultStruct()
            args = (self.ctx.handle, *[_P(c.data_ptr()) for c in corr], C.c_int64(int(corr[0].numel())),
                    _P(pose.ctypes.data), C.byref(cam), C.byref(sched), _P(state.data_ptr()), _P(0), C.c_int32(0),
                    C.c_int32(slot))
            self._slots[slot] = (state, pose, res, args)
        state, pose, res, args = self._slots[slot]
        pose[:] = np.asarray(w_T_c0, dtype=np.float64).reshape(16)
        rc = self.ctx.lib.picp_run_enqueue(*args)
        if rc != PICP_OK:
            check(rc, "picp_run_enqueue")

    def collect(self, slot):
        state, pose, res, args = self._slots[slot]
        rc = self.ctx.lib.picp_run_collect(self.ctx.handle, slot, pose.ctypes.data, C.byref(res))
        if rc == PICP_ERR_RETRY:                      # the launch gave up on a shared device: run it synchronously
            corr, cam, sched = self._keep
            return self.ctx.run(corr, pose.reshape(4, 4), cam, sched, want_history=False)[0]
        if rc != PICP_OK:
            check(rc, "picp_run_collect")
        return res


_contexts = {}


def get_context(device=0) -> Context:
    """Process-wide context per device (created on first use)."""
    key = int(device)
    ctx = _contexts.get(key)
    if ctx is None:
        ctx = Context(key)
        _contexts[key] = ctx
    ctx.use_current_stream()
    return ctx
