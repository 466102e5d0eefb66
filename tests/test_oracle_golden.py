"""The CPU restatement (oracle/picp_oracle.py) against vectors captured from the reference itself.

Bar: integer results (inlier counts, in-image counts, association indices, iteration
counts, best index, validity) identical; float results within 1e-9 norm-relative
(observed: ~1e-13).
"""
import numpy as np
import pytest

from tests.helpers import load_golden, rel_err
from oracle import picp_oracle as po

TOL = 1e-9
from tests.helpers import b_err


def test_v2T_matches_reference_float32_trig():
    g = load_golden("ref_v2T.npz")
    for v, T in zip(g["v"], g["T"]):
        assert np.array_equal(po.v2T(v), T)        # bit-exact: same float32 sin/cos, same products


def test_association_matches_reference():
    g = load_golden("ref_association.npz")
    s1 = {"position": g["pos1"].tolist(), "appearance": g["app1"].tolist()}
    s2 = {"position": g["pos2"].tolist(), "appearance": g["app2"].tolist()}
    m0 = po.data_association_on_appearance(s1, s2)
    assert np.array_equal(np.array(m0["points_1"]), g["m0_p1"])
    assert np.array_equal(np.array(m0["points_2"]), g["m0_p2"])
    assert np.array_equal(np.array(m0["appearance"]), g["m0_app"], equal_nan=True)
    params = po.IcpParams()
    m2 = po.data_association_on_appearance(s1, s2, projection=2, c_T_w=g["c_T_w"], params=params)
    assert np.array_equal(np.array(m2["points_1"]), g["m2_p1"])
    assert np.array_equal(np.array(m2["points_2"]), g["m2_p2"])
    assert rel_err(np.array(m2["projected_points_2"]), g["m2_proj"]) < 1e-14
    s1b = {"position": g["pos2"][: len(g["app1"])].tolist(), "appearance": g["app1"].tolist()}
    m1 = po.data_association_on_appearance(s1b, s2, projection=1, c_T_w=g["c_T_w"], params=params)
    assert np.array_equal(np.array(m1["points_1"]), g["m1_p1"])
    assert np.array_equal(np.array(m1["points_2"]), g["m1_p2"])
    assert rel_err(np.array(m1["projected_points_1"]), g["m1_proj"]) < 1e-14


def test_linearize_and_one_step_synthetic():
    g = load_golden("ref_synthetic.npz")
    params = po.IcpParams()
    for ci in range(int(g["n_cases"])):
        uv, pw, thr, pose = g[f"c{ci}_uv"], g[f"c{ci}_pw"], float(g[f"c{ci}_thr"]), g[f"c{ci}_pose"]
        H, b, n, err, _ = po.linearize(uv, pw, thr, pose, params) if len(uv) else (np.zeros((6, 6)), np.zeros(6), 0, np.inf, None)
        assert n == int(g[f"c{ci}_n"]), ci
        assert rel_err(H, g[f"c{ci}_H"]) < TOL and rel_err(b, g[f"c{ci}_b"]) < TOL, ci
        if np.isinf(g[f"c{ci}_err"]):
            assert np.isinf(err)
        else:
            assert abs(err - g[f"c{ci}_err"]) <= TOL * abs(g[f"c{ci}_err"])
        step = po.one_step(uv, pw, pose, thr, 64.0, params)
        assert step["computation_done"] == bool(g[f"c{ci}_done"])
        assert rel_err(step["T"], g[f"c{ci}_T1"]) < TOL, ci


def test_linearize_edge_cases():
    g = load_golden("ref_synthetic.npz")
    params = po.IcpParams()
    for ti in range(int(g["n_edge"])):
        H, b, n, err, aux = po.linearize(g["edge_uv"], g["edge_pw"], float(g[f"edge{ti}_thr"]), np.eye(4), params)
        assert n == int(g[f"edge{ti}_n"])
        assert rel_err(H, g[f"edge{ti}_H"]) < TOL and rel_err(b, g[f"edge{ti}_b"]) < TOL
        assert abs(err - g[f"edge{ti}_err"]) <= TOL * abs(g[f"edge{ti}_err"])


def _check_loop(hist, T, valid, g, pre, lo=None, hi=None):
    sl = slice(lo, hi)
    assert len(hist["error"]) == len(g[f"{pre}error"][sl])
    assert np.array_equal(np.array(hist["num_inliers"]), g[f"{pre}num_inliers"][sl])
    assert np.array_equal(np.array(hist["n_inside"]), g[f"{pre}n_inside"][sl])
    assert np.array_equal(np.array(hist["kernel_threshold"], dtype=np.float64), g[f"{pre}kernel_threshold"][sl])
    assert np.array_equal(np.array(hist["dumping_factor"], dtype=np.float64), g[f"{pre}dumping_factor"][sl])
    e_ref = g[f"{pre}error"][sl]
    e = np.array(hist["error"])
    assert np.array_equal(np.isinf(e), np.isinf(e_ref))
    fin = np.isfinite(e_ref)
    assert np.all(np.abs(e[fin] - e_ref[fin]) <= 1e-8 * np.abs(e_ref[fin]) + 1e-18)
    assert rel_err(np.array(hist["T"]), g[f"{pre}T"][sl]) < TOL
    for k, (H, b) in enumerate(zip(hist["H"], hist["b"])):
        Href, bref = g[f"{pre}H"][sl][k], g[f"{pre}b"][sl][k]
        assert rel_err(H, Href) < TOL or np.max(np.abs(Href)) == 0
        assert b_err(b, bref, Href, g[f"{pre}num_inliers"][sl][k], e_ref[k]) < TOL


def test_full_loop_synthetic():
    g = load_golden("ref_synthetic.npz")
    for li in range(int(g["n_loops"])):
        pre = f"loop{li}_"
        params = po.IcpParams(num_iterations=int(g[f"{pre}num_iterations"]))
        T, valid, hist = po.projective_icp(g[f"{pre}uv"], g[f"{pre}pw"], g[f"{pre}pose"], params, record_linear=True)
        _check_loop(hist, T, valid, g, pre)
        assert valid == bool(g[f"{pre}valid"])
        assert rel_err(T, g[f"{pre}T_out"]) < TOL


def test_full_loop_general_camera_matrix():
    """The reference with a camera matrix that has skew and unequal focal lengths (it reads whatever
    data/camera.dat holds, src/Camera.py:20-24)."""
    g = load_golden("ref_general_camera.npz")
    for li in range(int(g["n_loops"])):
        pre = f"loop{li}_"
        params = po.IcpParams(K=g["K"], num_iterations=int(g[f"{pre}num_iterations"]))
        T, valid, hist = po.projective_icp(g[f"{pre}uv"], g[f"{pre}pw"], g[f"{pre}pose"], params, record_linear=True)
        _check_loop(hist, T, valid, g, pre)
        assert valid == bool(g[f"{pre}valid"])
        assert rel_err(T, g[f"{pre}T_out"]) < TOL


def test_full_loop_degenerate_points():
    """The reference on a problem with points on and behind the camera plane and with infinite, NaN
    and huge coordinates or measurements mixed in, from three start poses (12 iterations each)."""
    g = load_golden("ref_degenerate.npz")
    for li in range(int(g["n_loops"])):
        pre = f"loop{li}_"
        params = po.IcpParams(num_iterations=int(g["num_iterations"]))
        with np.errstate(all="ignore"):
            T, valid, hist = po.projective_icp(g["uv"], g["pw"], g[f"{pre}pose"], params, record_linear=True)
        _check_loop(hist, T, valid, g, pre)
        assert valid == bool(g[f"{pre}valid"])
        assert rel_err(T, g[f"{pre}T_out"]) < TOL


def test_gtmap_icp_all_120_frames(bundled):
    """The reference's LM loop on the bundled measurements with the ground-truth map: 1 574 iterations
    over 120 frame pairs, covering zero-inlier iterations and every schedule branch."""
    g = load_golden("ref_gtmap_icp.npz")
    params = po.IcpParams()
    offs = g["iter_offsets"]
    mo = bundled["meas_offsets"]
    for f in range(120):
        uv = bundled["meas_uv"][mo[f + 1]:mo[f + 2]]
        app = bundled["meas_app"][mo[f + 1]:mo[f + 2]]
        i1, i2 = po.associate(app, g["map_appearance"])
        # the simulator's actual_point_id is the world.dat row with the same descriptor
        assert np.array_equal(i2, bundled["meas_actual_id"][mo[f + 1]:mo[f + 2]][i1])
        T, valid, hist = po.projective_icp(uv[i1], g["map_position"][i2], g["start_pose"][f], params, record_linear=True)
        _check_loop(hist, T, valid, g, "", offs[f], offs[f + 1])
        assert valid == bool(g["valid"][f])
        assert rel_err(T, g["T_out"][f]) < TOL
