"""The C restatement of the reference loop (oracle/picp_oracle_c.c) against the vectors captured from
the reference itself, and against the numpy restatement."""
import numpy as np
import pytest

from oracle import picp_oracle as po
from oracle import picp_oracle_c as oc
from tests.helpers import b_err, load_golden, rel_err
from tests.synthetic import make_problem

TOL = 1e-9


@pytest.fixture(scope="module", autouse=True)
def _built():
    oc.build()


def _check_rows(rows, g, pre, lo=None, hi=None):
    sl = slice(lo, hi)
    e_ref = g[f"{pre}error"][sl]
    assert rows.shape[0] == len(e_ref)
    assert np.array_equal(rows[:, oc.H_NUM_INLIERS].astype(np.int64), g[f"{pre}num_inliers"][sl])
    assert np.array_equal(rows[:, oc.H_N_INSIDE].astype(np.int64), g[f"{pre}n_inside"][sl])
    assert np.array_equal(rows[:, oc.H_KERNEL_THR], g[f"{pre}kernel_threshold"][sl])
    assert np.array_equal(rows[:, oc.H_DAMPING], g[f"{pre}dumping_factor"][sl])
    e = rows[:, oc.H_ERROR]
    assert np.array_equal(np.isinf(e), np.isinf(e_ref))
    fin = np.isfinite(e_ref)
    assert np.all(np.abs(e[fin] - e_ref[fin]) <= 1e-8 * np.abs(e_ref[fin]) + 1e-18)
    assert rel_err(rows[:, oc.H_T:oc.H_T + 16].reshape(-1, 4, 4), g[f"{pre}T"][sl]) < TOL
    Href, bref = g[f"{pre}H"][sl], g[f"{pre}b"][sl]
    for k in range(rows.shape[0]):
        if np.max(np.abs(Href[k])) > 0:
            assert rel_err(rows[k, oc.H_H:oc.H_H + 36].reshape(6, 6), Href[k]) < TOL, k
        assert b_err(rows[k, oc.H_B:oc.H_B + 6], bref[k], Href[k], g[f"{pre}num_inliers"][sl][k], e_ref[k]) < TOL, k


def _check_result(T, valid, best, e_ref, T_hist, T_out):
    if valid:
        ref_best = int(np.argmin(e_ref))
        if best == ref_best:
            assert rel_err(T, T_out) < TOL
        else:       # converged errors tie to rounding noise
            assert abs(e_ref[best] - e_ref[ref_best]) <= 1e-9 * abs(e_ref[ref_best])
            assert rel_err(T, T_hist[best]) < TOL
    else:
        assert rel_err(T, T_out) < TOL


def test_v2T_float32_trig_matches_reference():
    g = load_golden("ref_v2T.npz")
    for v, T in zip(g["v"], g["T"]):
        Tc = oc.v2T(v)
        if np.max(np.abs(v[3:])) <= 1e-2:
            assert np.max(np.abs(Tc - T)) <= 4.5e-16, v
        else:
            assert np.max(np.abs(Tc - T)) < 2.5e-7, v      # numpy's float32 sin/cos are within 1.5 ulp, not exact


@pytest.mark.parametrize("name", ["ref_synthetic.npz", "ref_general_camera.npz"])
def test_full_loops(name):
    g = load_golden(name)
    for li in range(int(g["n_loops"])):
        pre = f"loop{li}_"
        kw = {"K": g["K"]} if "K" in g else {}
        params = po.IcpParams(num_iterations=int(g[f"{pre}num_iterations"]), **kw)
        T, valid, rows, best = oc.projective_icp(g[f"{pre}uv"], g[f"{pre}pw"], g[f"{pre}pose"], params)
        _check_rows(rows, g, pre)
        assert valid == bool(g[f"{pre}valid"])
        _check_result(T, valid, best, g[f"{pre}error"], g[f"{pre}T"], g[f"{pre}T_out"])


def test_full_loops_degenerate_points():
    g = load_golden("ref_degenerate.npz")
    for li in range(int(g["n_loops"])):
        pre = f"loop{li}_"
        params = po.IcpParams(num_iterations=int(g["num_iterations"]))
        T, valid, rows, best = oc.projective_icp(g["uv"], g["pw"], g[f"{pre}pose"], params)
        _check_rows(rows, g, pre)
        assert valid == bool(g[f"{pre}valid"])
        _check_result(T, valid, best, g[f"{pre}error"], g[f"{pre}T"], g[f"{pre}T_out"])


def test_gtmap_icp_all_120_frames(bundled):
    """The reference's 1 574 iterations on the bundled measurements with the ground-truth map."""
    g = load_golden("ref_gtmap_icp.npz")
    params = po.IcpParams()
    offs = g["iter_offsets"]
    mo = bundled["meas_offsets"]
    for f in range(120):
        uv = bundled["meas_uv"][mo[f + 1]:mo[f + 2]]
        act = bundled["meas_actual_id"][mo[f + 1]:mo[f + 2]]
        T, valid, rows, best = oc.projective_icp(uv, g["map_position"][act], g["start_pose"][f], params)
        _check_rows(rows, g, "", offs[f], offs[f + 1])
        assert valid == bool(g["valid"][f])
        _check_result(T, valid, best, g["error"][offs[f]:offs[f + 1]], g["T"][offs[f]:offs[f + 1]], g["T_out"][f])


def test_agrees_with_the_numpy_restatement_on_a_large_problem():
    uv, pw, _ = make_problem(200_000, seed=4)
    params = po.IcpParams(num_iterations=8, disable_stop=True)
    T, valid, rows, best = oc.projective_icp(uv, pw, np.eye(4), params)
    T2, valid2, h2 = po.projective_icp(uv, pw, np.eye(4), params, record_linear=True)
    assert valid == valid2
    assert rows[:, oc.H_NUM_INLIERS].astype(int).tolist() == h2["num_inliers"]
    assert rows[:, oc.H_N_INSIDE].astype(int).tolist() == h2["n_inside"]
    assert rel_err(rows[:, oc.H_T:oc.H_T + 16].reshape(-1, 4, 4), np.array(h2["T"])) < TOL
    assert rel_err(T, T2) < TOL


def test_agrees_with_the_numpy_restatement_on_degenerate_points():
    """Points on and behind the camera plane and with non-finite coordinates or measurements inside a
    regular problem, from a rotated start pose: the reference's homogeneous division makes a point with
    an infinite or NaN coordinate all-NaN (src/Camera.py:104-106), which passes the in-image test and
    fails chi <= threshold; both restatements must count alike."""
    uv, pw, _ = make_problem(3000, seed=41)
    big, tiny = 1e300, 5e-324
    odd_pw = np.array([[0.0, 0.0, 0.0], [1.0, 1.0, 0.0], [0.3, -0.2, -4.0], [np.nan, 0.0, 5.0], [0.0, np.inf, 5.0],
                       [0.0, 0.0, np.inf], [big, 0.0, big], [tiny, tiny, 1.0], [-big, big, 2.0], [-np.inf, 1.0, 1.0],
                       [0.0, 0.0, 1.0], [np.nan, np.nan, np.nan]])
    odd_uv = np.array([[320.0, 240.0]] * len(odd_pw))
    odd_uv[5] = [np.nan, 240.0]
    odd_uv[7] = [np.inf, -np.inf]
    where = np.linspace(3, 2990, len(odd_pw)).astype(int)
    uv[where], pw[where] = odd_uv, odd_pw
    start = po.v2T([0.02, -0.01, 0.05, 0.02, -0.03, 0.01])
    params = po.IcpParams(num_iterations=5, disable_stop=True)
    with np.errstate(all="ignore"):
        T2, valid2, h2 = po.projective_icp(uv, pw, start, params, record_linear=True)
    T, valid, rows, best = oc.projective_icp(uv, pw, start, params)
    assert valid == valid2
    assert rows[:, oc.H_N_INSIDE].astype(int).tolist() == h2["n_inside"]
    assert rows[:, oc.H_NUM_INLIERS].astype(int).tolist() == h2["num_inliers"]
    assert rel_err(rows[:, oc.H_T:oc.H_T + 16].reshape(-1, 4, 4), np.array(h2["T"])) < TOL
