"""ctypes access to the C restatement of the reference loop (oracle/picp_oracle_c.c).

TEST INFRASTRUCTURE, like the rest of oracle/: used by tests/ and by bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "liboracle.so")
HIST_STRIDE = 80
H_T, H_ERROR, H_NUM_INLIERS, H_KERNEL_THR, H_DAMPING, H_N_INSIDE = 0, 16, 17, 18, 19, 20
H_DX, H_B, H_H = 26, 32, 38


class Params(C.Structure):
    _fields_ = [("K", C.c_double * 9), ("width", C.c_double), ("height", C.c_double),
                ("num_iterations", C.c_int32), ("min_inliers", C.c_int32),
                ("kernel_threshold", C.c_double), ("min_kernel_threshold", C.c_double),
                ("max_kernel_threshold", C.c_double), ("damping", C.c_double), ("min_damping", C.c_double),
                ("max_damping", C.c_double), ("disable_stop", C.c_int32), ("reserved", C.c_int32)]


_lib = None


def build():
    subprocess.run(["make", "-C", HERE], check=True, capture_output=True)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.oc_projective_icp.restype = C.c_int
        _lib.oc_projective_icp.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(Params), C.c_void_p,
                                           C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p,
                                           C.c_int32]
        _lib.oc_v2T.restype = None
        _lib.oc_v2T.argtypes = [C.c_void_p, C.c_void_p]
    return _lib


def params_struct(p) -> Params:
    """From an oracle.picp_oracle.IcpParams."""
    s = Params()
    K = np.asarray(p.K, dtype=np.float64).reshape(9)
    for i in range(9):
        s.K[i] = float(K[i])
    s.width, s.height = float(p.width), float(p.height)
    s.num_iterations, s.min_inliers = int(p.num_iterations), int(p.min_inliers)
    s.kernel_threshold, s.min_kernel_threshold = float(p.kernel_threshold), float(p.min_kernel_threshold)
    s.max_kernel_threshold = float(p.max_kernel_threshold)
    s.damping, s.min_damping, s.max_damping = float(p.damping), float(p.min_damping), float(p.max_damping)
    s.disable_stop = 1 if p.disable_stop else 0
    return s


def v2T(v):
    v = np.ascontiguousarray(v, dtype=np.float64)
    T = np.empty(16)
    load().oc_v2T(v.ctypes.data, T.ctypes.data)
    return T.reshape(4, 4)


def projective_icp(uv, pw, w_T_start, params, want_history=True):
    """Returns (T, valid, history rows [n_it, HIST_STRIDE] or None, best_index)."""
    uv = np.ascontiguousarray(uv, dtype=np.float64).reshape(-1, 2)
    pw = np.ascontiguousarray(pw, dtype=np.float64).reshape(-1, 3)
    T0 = np.ascontiguousarray(w_T_start, dtype=np.float64).reshape(16)
    ps = params_struct(params)
    rows = max(int(params.num_iterations), 1)
    hist = np.zeros((rows, HIST_STRIDE)) if want_history else None
    T = np.empty(16)
    valid, n_it, best = C.c_int32(), C.c_int32(), C.c_int32()
    rc = load().oc_projective_icp(uv.ctypes.data, pw.ctypes.data, len(uv), T0.ctypes.data, C.byref(ps), T.ctypes.data,
                                  C.byref(valid), C.byref(n_it), C.byref(best),
                                  hist.ctypes.data if want_history else None, rows if want_history else 0)
    if rc != 0:
        raise MemoryError("oc_projective_icp")
    return T.reshape(4, 4), bool(valid.value), (hist[: n_it.value] if want_history else None), int(best.value)
