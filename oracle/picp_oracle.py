"""CPU restatement (numpy, float64) of the reference projective-ICP hot path.

THIS FILE IS TEST INFRASTRUCTURE.  It is the checker the HIP path is compared
against (tests/, __graft_entry__.smoke(), and bench.py's cpu_baseline leg).  The
shipped package never imports it; the product path fails loudly when the HIP
library is missing.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function below
against vectors captured from the reference itself (imported and run in the
build container by oracle/gen_golden.py) and against the reference's own
published outputs (outputs/final_results/*.dat).

Each function cites the reference lines it restates (paths relative to the
reference checkout).  The restatement is vectorised: where the reference loops
over correspondences in Python and sums in index order, this file uses numpy
array arithmetic, so float64 sums can differ from the reference in the last
bits (tolerances are stated in the tests); integer results (inlier counts,
association indices) are identical.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

# --------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------


@dataclass
class IcpParams:
    """Scalar inputs of the loop.

    Defaults are the shipped YAML values (config/config.yaml:11-22), the ones
    the published errors.dat was produced with, and the bundled camera
    (data/camera.dat:2-13; K is parsed with int(), src/Camera.py:23).
    """

    K: np.ndarray = field(default_factory=lambda: np.array(
        [[180.0, 0.0, 320.0], [0.0, 180.0, 240.0], [0.0, 0.0, 1.0]]))
    width: float = 640.0
    height: float = 480.0
    num_iterations: int = 250
    min_inliers: int = 10
    kernel_threshold: float = 1000.0
    min_kernel_threshold: float = 10.0
    max_kernel_threshold: float = 2000.0
    damping: float = 64.0
    min_damping: float = 1.0
    max_damping: float = 64.0
    # the reference stops on error < 0.01 or stuck > 100 (src/ProjectiveICP.py:228);
    # benchmarks run a fixed iteration count, which this switch provides.
    disable_stop: bool = False


# --------------------------------------------------------------------------
# geometry (src/geometry_utils.py)
# --------------------------------------------------------------------------


def v2T(v) -> np.ndarray:
    """Pose increment -> 4x4 (src/geometry_utils.py:38-45, :19-23, :4-17).

    R = Rx Ry Rz.  The three angles are cast to float32 first
    (``rotation_euler.astype(np.float32)``, :41), so every sin/cos is a float32
    value; they are then stored into float64 matrices and multiplied in
    float64.  The translation stays float64.
    """
    v = np.asarray(v, dtype=np.float64)
    a = v[3:6].astype(np.float32)
    cx, sx = float(np.cos(a[0])), float(np.sin(a[0]))
    cy, sy = float(np.cos(a[1])), float(np.sin(a[1]))
    cz, sz = float(np.cos(a[2])), float(np.sin(a[2]))
    Rx = np.array([[1.0, 0.0, 0.0], [0.0, cx, -sx], [0.0, sx, cx]])
    Ry = np.array([[cy, 0.0, sy], [0.0, 1.0, 0.0], [-sy, 0.0, cy]])
    Rz = np.array([[cz, -sz, 0.0], [sz, cz, 0.0], [0.0, 0.0, 1.0]])
    T = np.eye(4)
    T[:3, :3] = Rx @ Ry @ Rz
    T[:3, 3] = v[:3]
    return T


def skew(w) -> np.ndarray:
    """src/geometry_utils.py:59-62."""
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


# --------------------------------------------------------------------------
# camera projection (src/Camera.py:82-121)
# --------------------------------------------------------------------------


def project_points(c_T_w: np.ndarray, params: IcpParams, pw: np.ndarray):
    """Vectorised ``Camera.project_point``.

    Returns (inside[M] bool, uv[M,2], pc[M,3]).  Order of operations follows
    src/Camera.py:104-119: homogeneous transform and division by w, K @ pc,
    division by the third component (before any test, so z == 0 yields inf/nan
    which the tests then reject), ``z <= 0`` reject, image-bounds reject.
    """
    pw = np.asarray(pw, dtype=np.float64).reshape(-1, 3)
    K = np.asarray(params.K, dtype=np.float64)
    ph = np.empty((pw.shape[0], 4))
    for r in range(4):
        ph[:, r] = (c_T_w[r, 0] * pw[:, 0] + c_T_w[r, 1] * pw[:, 1]
                    + c_T_w[r, 2] * pw[:, 2] + c_T_w[r, 3])
    with np.errstate(divide="ignore", invalid="ignore"):
        pc = ph[:, :3] / ph[:, 3:4]
        h = np.empty_like(pc)
        for r in range(3):
            h[:, r] = K[r, 0] * pc[:, 0] + K[r, 1] * pc[:, 1] + K[r, 2] * pc[:, 2]
        uv = h[:, :2] / h[:, 2:3]
        behind = pc[:, 2] <= 0
        outside = ((uv[:, 0] < 0) | (uv[:, 0] >= params.width)
                   | (uv[:, 1] < 0) | (uv[:, 1] >= params.height))
    inside = ~(behind | outside)
    return inside, uv, pc


# --------------------------------------------------------------------------
# linearisation (src/ProjectiveICP.py:276-353)
# --------------------------------------------------------------------------


def errors_and_jacobians(uv_meas, pw, w_T_c0, params: IcpParams, c_T_w=None):
    """Vectorised ``__error_and_jacobian`` (src/ProjectiveICP.py:326-353).

    Returns (inside[M], e[M,2], J[M,2,6]).  ``c_T_w`` is the camera state the
    reference projects with (src/Camera.py:104); inside the reference loop it
    always equals ``inv(w_T_c0)`` (src/ProjectiveICP.py:232), which is also the
    matrix used for p_hat (:336), so one transform serves both.
    """
    if c_T_w is None:
        c_T_w = np.linalg.inv(w_T_c0)
    uv_meas = np.asarray(uv_meas, dtype=np.float64).reshape(-1, 2)
    K = np.asarray(params.K, dtype=np.float64)
    inside, uv, _ = project_points(c_T_w, params, pw)
    e = uv_meas - uv                                                    # :333
    _, _, p_hat = project_points(np.linalg.inv(w_T_c0), params, pw)     # :336-337
    M = p_hat.shape[0]
    with np.errstate(divide="ignore", invalid="ignore"):
        p_cam = p_hat @ K.T                                             # :338
        z_inv = 1.0 / p_cam[:, 2]                                       # :345
        z_inv2 = z_inv * z_inv
        J_proj = np.zeros((M, 2, 3))                                    # :348-349
        J_proj[:, 0, 0] = z_inv
        J_proj[:, 0, 2] = -p_cam[:, 0] * z_inv2
        J_proj[:, 1, 1] = z_inv
        J_proj[:, 1, 2] = -p_cam[:, 1] * z_inv2
        J_icp = np.zeros((M, 3, 6))                                     # :341-343
        J_icp[:, 0, 0] = J_icp[:, 1, 1] = J_icp[:, 2, 2] = 1.0
        x, y, z = p_hat[:, 0], p_hat[:, 1], p_hat[:, 2]
        J_icp[:, 0, 4] = z
        J_icp[:, 0, 5] = -y
        J_icp[:, 1, 3] = -z
        J_icp[:, 1, 5] = x
        J_icp[:, 2, 3] = y
        J_icp[:, 2, 4] = -x
        J = (J_proj @ K) @ J_icp                                        # :351
    return inside, e, J


def linearize(uv_meas, pw, kernel_threshold, w_T_c0, params: IcpParams, c_T_w=None):
    """``__linearize`` (src/ProjectiveICP.py:276-323).

    Stage 1 keeps ``chi <= kernel_threshold`` (:294); if nothing survives the
    result is (0, 0, 0, inf) (:300).  Stage 2 keeps ``chi < mean(chi)`` (strict,
    :301-302) and replaces the stage-1 set only when at least ``min_inliers``
    survive (:307-310).  Returns (H[6,6], b[6], num_inliers, error, aux) where
    aux carries the intermediate counts the device path also reports.
    """
    inside, e, J = errors_and_jacobians(uv_meas, pw, w_T_c0, params, c_T_w)
    H = np.zeros((6, 6))
    b = np.zeros(6)
    with np.errstate(invalid="ignore"):
        # :293 `error.T @ error`: numpy's dot of a 2-vector is fma(e1, e1, e0 * e0) on this build; the
        # batched matrix product goes through the same BLAS kernel and reproduces it bit for bit
        chi = (e[:, None, :] @ e[:, :, None]).reshape(-1)
        keep1 = inside & (chi <= kernel_threshold)                      # :290, :294
    aux = {"n_inside": int(inside.sum()), "n_stage1": int(keep1.sum()), "n_stage2": 0,
           "used_stage2": False}
    if not keep1.any():
        return H, b, 0, np.inf, aux                                     # :300
    chi1 = chi[keep1]
    mean1 = np.mean(chi1)                                               # :301
    keep2 = keep1 & (chi < mean1)                                       # :302
    aux["n_stage2"] = int(keep2.sum())
    aux["mean_stage1"] = float(mean1)
    if keep2.sum() >= params.min_inliers:                               # :307
        final = keep2
        aux["used_stage2"] = True
    else:
        final = keep1
    Jf, ef = J[final], e[final]
    H = np.einsum("mri,mrj->ij", Jf, Jf)                                # :316
    b = np.einsum("mri,mr->i", Jf, ef)                                  # :317
    n = int(final.sum())
    err = float(np.mean(chi[final])) if n > 0 else np.inf               # :319-321
    return H, b, n, err, aux


def one_step(uv_meas, pw, w_T_c0, kernel_threshold, damping, params: IcpParams, c_T_w=None):
    """``__one_step`` (src/ProjectiveICP.py:260-273)."""
    uv_meas = np.asarray(uv_meas, dtype=np.float64).reshape(-1, 2)
    if uv_meas.shape[0] == 0:                                           # :263
        return {"T": w_T_c0, "num_inliers": 0, "error": np.inf, "computation_done": False,
                "H": np.zeros((6, 6)), "b": np.zeros(6), "dx": np.zeros(6)}
    H, b, n, err, aux = linearize(uv_meas, pw, kernel_threshold, w_T_c0, params, c_T_w)
    Hd = H + np.eye(6) * damping                                        # :269
    dx = np.linalg.lstsq(Hd, -b, rcond=None)[0]                         # :270
    T = w_T_c0 @ v2T(dx)                                                # :271
    return {"T": T, "num_inliers": n, "error": err, "computation_done": True,
            "H": H, "b": b, "dx": dx, "aux": aux}


# --------------------------------------------------------------------------
# appearance association (src/data_association.py:3-57)
# --------------------------------------------------------------------------


def associate(app1, app2, chunk: int = 2048):
    """All (i, j) with ``appearance_1[i] == appearance_2[j]`` in i-major,
    j-minor order (src/data_association.py:29-31).

    The reference compares two Python lists of floats, i.e. element-wise float
    equality of every component: -0.0 == 0.0, NaN never equal.
    Returns (idx1[int64], idx2[int64]).
    """
    a1 = np.asarray(app1, dtype=np.float64)
    a2 = np.asarray(app2, dtype=np.float64)
    if a1.size == 0 or a2.size == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    a1 = a1.reshape(a1.shape[0], -1)
    a2 = a2.reshape(a2.shape[0], -1)
    out_i, out_j = [], []
    for s in range(0, a1.shape[0], chunk):
        blk = a1[s:s + chunk]
        # first component prunes almost everything before the full compare
        cand_i, cand_j = np.nonzero(blk[:, None, 0] == a2[None, :, 0])
        if cand_i.size:
            ok = (blk[cand_i] == a2[cand_j]).all(axis=1)
            out_i.append(cand_i[ok] + s)
            out_j.append(cand_j[ok])
    if not out_i:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    return np.concatenate(out_i).astype(np.int64), np.concatenate(out_j).astype(np.int64)


def data_association_on_appearance(set_1, set_2, projection=0, c_T_w=None, params=None):
    """Dict-of-lists form of the reference function (src/data_association.py:3-57).

    ``projection`` 1/2 additionally drops pairs whose set_1/set_2 position does
    not project inside the image (:36-50); the projected point is returned too.
    """
    idx1, idx2 = associate(set_1["appearance"], set_2["appearance"])
    p1 = [set_1["position"][i] for i in idx1]
    p2 = [set_2["position"][j] for j in idx2]
    app = [set_1["appearance"][i] for i in idx1]
    if projection in (1, 2):
        pts = np.asarray(p1 if projection == 1 else p2, dtype=np.float64).reshape(-1, 3)
        inside, uv, _ = project_points(c_T_w, params, pts)
        keep = np.nonzero(inside)[0]
        key = "projected_points_1" if projection == 1 else "projected_points_2"
        out = {"points_1": [p1[k] for k in keep], "points_2": [p2[k] for k in keep],
               key: [uv[k] for k in keep], "appearance": [app[k] for k in keep]}
        out["_idx1"] = idx1[keep]
        out["_idx2"] = idx2[keep]
        return out
    return {"points_1": p1, "points_2": p2, "appearance": app, "_idx1": idx1, "_idx2": idx2}


# --------------------------------------------------------------------------
# the LM loop with its schedules (src/ProjectiveICP.py:154-257)
# --------------------------------------------------------------------------


def projective_icp(uv_meas, pw, w_T_start, params: IcpParams, record_linear=False):
    """``__projective_ICP`` on pre-associated correspondences.

    ``uv_meas[M,2]`` / ``pw[M,3]`` are the appearance-matched pairs in the
    reference's i-major/j-minor order (the equality part of the association at
    src/ProjectiveICP.py:182 does not depend on the pose).  The pose-dependent
    part of that call -- dropping pairs whose world point does not project
    inside the image under the current ``c_T_w`` (src/data_association.py:44-46)
    -- is re-evaluated here every iteration, as the reference does.

    Returns (T, is_valid, history) with history holding the reference's five
    lists (:173, :233-237) plus 'best_index'.
    """
    uv_meas = np.asarray(uv_meas, dtype=np.float64).reshape(-1, 2)
    pw = np.asarray(pw, dtype=np.float64).reshape(-1, 3)
    w_T_c0 = np.array(w_T_start, dtype=np.float64)
    c_T_w = np.linalg.inv(w_T_c0)                                       # camera state on entry (:360)
    thr = params.kernel_threshold                                       # :160
    lam = params.damping                                                # :161
    limit = 10                                                          # :164
    error_prev = np.inf
    ring = np.zeros(limit)
    mean_slope = 0.0
    sigma_slope = 0.0
    stuck = 0
    flicker = 0
    hist = {"T": [], "error": [], "num_inliers": [], "kernel_threshold": [], "dumping_factor": []}
    if record_linear:
        hist.update({"H": [], "b": [], "dx": [], "n_inside": []})
    stop = False
    it = 0
    while not stop:                                                     # :177
        if it == params.num_iterations:                                 # :178
            break
        it += 1
        inside, _, _ = project_points(c_T_w, params, pw)                # :182 (projection=2 filter)
        uv_i, pw_i = uv_meas[inside], pw[inside]
        res = one_step(uv_i, pw_i, w_T_c0, thr, lam, params, c_T_w)     # :193
        w_T_c1, n_inl, error, done = res["T"], res["num_inliers"], res["error"], res["computation_done"]
        n_pts = uv_i.shape[0]
        if n_inl == n_pts:                                              # :200-205
            thr = params.min_kernel_threshold
        elif n_inl < params.min_inliers:
            if thr + 10 <= params.max_kernel_threshold:
                thr += 10
        elif n_inl >= params.min_inliers:
            if thr - 10 >= params.min_kernel_threshold:
                thr -= 10
        if it > 1:                                                      # :208-213
            if error == np.inf or error_prev == np.inf:
                slope = 1e3
            else:
                slope = np.abs(error_prev - error)
            ring[it % limit] = slope
            mean_slope = np.mean(ring)
            sigma_slope = np.std(ring)
        if done and mean_slope < 1e-2 and sigma_slope < 1e-2:           # :216-217
            stuck += 1
        else:
            stuck = 0
        if done and mean_slope > 1 and sigma_slope > 1:                 # :220-221
            flicker += 1
        else:
            flicker = 0
        if (lam / 2) >= params.min_damping and (stuck > limit or (stuck == 0 and flicker == 0)):  # :224
            lam /= 2
        if (lam * 2) <= params.max_damping and flicker > limit:         # :225
            lam *= 2
        if done and (error < 0.01 or stuck > 100) and not params.disable_stop:  # :228
            stop = True
        w_T_c0 = w_T_c1                                                 # :231
        c_T_w = np.linalg.inv(w_T_c0)                                   # :232
        hist["T"].append(w_T_c0)
        hist["error"].append(error)
        hist["num_inliers"].append(n_inl)
        hist["kernel_threshold"].append(thr)
        hist["dumping_factor"].append(lam)
        if record_linear:
            hist["H"].append(res["H"])
            hist["b"].append(res["b"])
            hist["dx"].append(res["dx"])
            hist["n_inside"].append(n_pts)
        error_prev = error
    best = int(np.argmin(hist["error"]))                                # :249
    hist["best_index"] = best
    if hist["error"][best] > 5:                                         # :249-251
        return np.array(w_T_start, dtype=np.float64), False, hist
    return hist["T"][best], True, hist                                  # :254-257
