"""Generate tests/golden/*.npz by running the REFERENCE itself in the build container.

TEST INFRASTRUCTURE, run by hand in the build container only:

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python oracle/gen_golden.py

It imports the reference from /root/reference (never copied, never shipped),
drives its hot path on the bundled data and on seeded synthetic inputs, and
stores inputs + expected outputs as small .npz fixtures.  The GPU box has no
/root/reference; tests read only the committed fixtures.

Two shims are applied before the import because of the container's package set,
not to change behaviour: numpy 2 dropped ``np.Inf`` (used at
src/ProjectiveICP.py:209,263) and OpenCV is absent (imported at module scope by
src/geometry_utils.py:2 but never called by the ICP loop).  ``initialize`` /
``triangulate_points`` need OpenCV and are therefore NOT captured here.
"""
from __future__ import annotations

import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

YAML = dict(num_iterations=250, min_inliers=10, kernel_threshold=1000, min_kernel_threshold=10,
            max_kernel_threshold=2000, dumping_factor=64, min_dumping_factor=1, max_dumping_factor=64)


def import_reference():
    np.Inf = np.inf
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    scratch = tempfile.mkdtemp(prefix="picp_golden_")
    os.symlink(os.path.join(REF, "data"), os.path.join(scratch, "data"))
    os.chdir(scratch)                      # Camera/Data open "data/..." relative to cwd
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    import src.ProjectiveICP as ref_picp_mod
    from src.Camera import Camera
    from src.Data import Data
    from src import data_association as ref_da
    from src import geometry_utils as ref_geo
    return ref_picp_mod, Camera, Data, ref_da, ref_geo


def make_picp(ref_picp_mod, cam, data, **over):
    p = dict(YAML)
    p.update(over)
    return ref_picp_mod.ProjectiveICP(cam, data, p["num_iterations"], p["min_inliers"], p["kernel_threshold"],
                                      p["min_kernel_threshold"], p["max_kernel_threshold"], p["dumping_factor"],
                                      p["min_dumping_factor"], p["max_dumping_factor"], False, False, False, [])


# --------------------------------------------------------------------------
# 1. bundled sequence + published outputs, repacked as arrays
# --------------------------------------------------------------------------


def pack_bundled(cam, data):
    K = np.array(cam.get_camera_matrix())
    traj = np.loadtxt(os.path.join(REF, "data", "trajectory.dat"))
    world = np.loadtxt(os.path.join(REF, "data", "world.dat"))
    offs, cur_id, act_id, uv, app, odom, gt = [0], [], [], [], [], [], []
    for f in range(121):
        m = data.get_measurements_data(f)
        cur_id += m["current_point_id"]
        act_id += m["actual_point_id"]
        uv += m["position"]
        app += m["appearance"]
        offs.append(len(cur_id))
        with open(os.path.join(REF, "data", f"meas-{f:05d}.dat")) as fh:
            for line in fh:
                if line.startswith("gt_pose"):
                    gt.append([float(x) for x in line.split()[1:]])
                elif line.startswith("odom_pose"):
                    odom.append([float(x) for x in line.split()[1:]])
    res = os.path.join(REF, "outputs", "final_results")
    np.savez_compressed(
        os.path.join(OUT, "bundled_sequence.npz"),
        camera_matrix=K, cam_transform=np.array(cam.get_camera_transform()),
        z_range=np.array(cam.get_camera_range()), resolution=np.array(cam.get_camera_resolution()),
        trajectory=traj, world=world,
        meas_offsets=np.array(offs), meas_cur_id=np.array(cur_id), meas_actual_id=np.array(act_id),
        meas_uv=np.array(uv), meas_app=np.array(app), meas_gt_pose=np.array(gt), meas_odom_pose=np.array(odom),
        published_errors=np.loadtxt(os.path.join(res, "errors.dat")),
        published_trajectory=np.loadtxt(os.path.join(res, "estimated_trajectory.dat")),
        published_world_points=np.loadtxt(os.path.join(res, "estimated_world_points.dat")))


# --------------------------------------------------------------------------
# 2. GT-map-driven ICP on all 120 frame pairs
# --------------------------------------------------------------------------


class Recorder:
    """Records the arguments of np.linalg.lstsq (H + lambda*I and -b) and the number of
    pairs the association returned, without editing the reference."""

    def __init__(self, ref_picp_mod):
        self.mod = ref_picp_mod
        self.lstsq_args = []
        self.dx = []
        self.n_assoc = []
        self._lstsq = np.linalg.lstsq
        self._assoc = ref_picp_mod.data_association_on_appearance

    def __enter__(self):
        def lstsq(A, B, rcond=None):
            out = self._lstsq(A, B, rcond=rcond)
            self.lstsq_args.append((np.array(A), np.array(B)))
            self.dx.append(np.array(out[0]))
            return out

        def assoc(*a, **k):
            m = self._assoc(*a, **k)
            self.n_assoc.append(len(m["points_1"]))
            return m

        np.linalg.lstsq = lstsq
        self.mod.data_association_on_appearance = assoc
        return self

    def __exit__(self, *exc):
        np.linalg.lstsq = self._lstsq
        self.mod.data_association_on_appearance = self._assoc


def run_icp_capture(ref_picp_mod, picp, cam, pose, meas, frame):
    picp._ProjectiveICP__current_pose = pose
    cam.set_c_T_w(np.linalg.inv(pose))
    with Recorder(ref_picp_mod) as rec:
        T, valid, hist = picp._ProjectiveICP__projective_ICP(meas, frame)
    n_it = len(hist["T"])
    H = np.zeros((n_it, 6, 6))
    b = np.zeros((n_it, 6))
    dx = np.zeros((n_it, 6))
    k = 0
    lam_before = [picp._ProjectiveICP__base_dumping_factor] + list(hist["dumping_factor"][:-1])
    for it in range(n_it):
        if rec.n_assoc[it] == 0:          # __one_step returned before the solve (:263)
            continue
        A, B = rec.lstsq_args[k]
        H[it] = A - np.eye(6) * lam_before[it]
        b[it] = -B
        dx[it] = rec.dx[k]
        k += 1
    assert k == len(rec.lstsq_args)
    return dict(T=np.array(hist["T"]), error=np.array(hist["error"], dtype=np.float64),
                num_inliers=np.array(hist["num_inliers"], dtype=np.int64),
                kernel_threshold=np.array(hist["kernel_threshold"], dtype=np.float64),
                dumping_factor=np.array(hist["dumping_factor"], dtype=np.float64),
                n_inside=np.array(rec.n_assoc, dtype=np.int64), H=H, b=b, dx=dx,
                T_out=np.array(T), valid=bool(valid))


def gen_gtmap(ref_picp_mod, cam, data):
    gt_poses = data.get_trajectory_data_poses()
    C = np.array(cam.get_camera_transform())
    world = data.get_world_data()
    w_T_cam0 = gt_poses[0] @ C
    cam0_T_w = np.linalg.inv(w_T_cam0)
    wp = np.array(world["position"])
    map_pos = (cam0_T_w[:3, :3] @ wp.T).T + cam0_T_w[:3, 3]
    picp = make_picp(ref_picp_mod, cam, data)
    m = picp.get_map()
    m["position"] = [p for p in map_pos]
    m["appearance"] = [list(a) for a in world["appearance"]]
    m["error"] = [0.0] * len(m["position"])
    out = {"map_position": map_pos, "map_appearance": np.array(world["appearance"])}
    offs = [0]
    keys = ["T", "error", "num_inliers", "kernel_threshold", "dumping_factor", "n_inside", "H", "b", "dx"]
    cat = {k: [] for k in keys}
    start_pose, T_out, valid = [], [], []
    for f in range(120):
        pose = cam0_T_w @ gt_poses[f] @ C            # camera f expressed in the camera-0 frame
        r = run_icp_capture(ref_picp_mod, picp, cam, pose, data.get_measurements_data(f + 1), f)
        for k in keys:
            cat[k].append(r[k])
        offs.append(offs[-1] + len(r["error"]))
        start_pose.append(pose)
        T_out.append(r["T_out"])
        valid.append(r["valid"])
    for k in keys:
        out[k] = np.concatenate(cat[k], axis=0)
    out["iter_offsets"] = np.array(offs)
    out["start_pose"] = np.array(start_pose)
    out["T_out"] = np.array(T_out)
    out["valid"] = np.array(valid)
    np.savez_compressed(os.path.join(OUT, "ref_gtmap_icp.npz"), **out)
    print("gtmap: iterations", offs[-1], "valid", int(np.sum(valid)), "/ 120")


# --------------------------------------------------------------------------
# 3. seeded synthetic problems through __linearize / __one_step / full loop
# --------------------------------------------------------------------------


def synth_problem(rng, M, geo, outlier_frac=0.1, noise=0.5, behind_frac=0.05):
    K = np.array([[180.0, 0, 320], [0, 180.0, 240], [0, 0, 1]])
    T_true = geo.v2T([0.05, -0.03, 0.2, 0.01, -0.02, 0.015])
    z = rng.uniform(1.0, 20.0, M)
    u = rng.uniform(0, 640, M)
    v = rng.uniform(0, 480, M)
    pw = np.stack([(u - 320) / 180 * z, (v - 240) / 180 * z, z], axis=1)
    nb = int(M * behind_frac)
    if nb:
        pw[:nb, 2] *= -1.0                                   # behind the camera
    c_T_w = np.linalg.inv(T_true)
    pc = (c_T_w[:3, :3] @ pw.T).T + c_T_w[:3, 3]
    uv = (K @ pc.T).T
    uv = uv[:, :2] / uv[:, 2:3] + rng.normal(0, noise, (M, 2))
    no = int(M * outlier_frac)
    if no:
        idx = rng.choice(M, no, replace=False)
        uv[idx] = np.stack([rng.uniform(0, 640, no), rng.uniform(0, 480, no)], axis=1)
    return uv, pw


def gen_synthetic(ref_picp_mod, cam, data, geo):
    rng = np.random.default_rng(0)
    out = {}
    cases = []
    sizes = [0, 1, 9, 10, 11, 64, 1000, 20000]
    for M in sizes:
        for thr in (25.0, 1000.0):
            cases.append((M, thr, np.eye(4)))
    cases.append((1000, 1000.0, geo.v2T([0.04, -0.02, 0.15, 0.008, -0.015, 0.01])))
    cases.append((1000, 2.0, geo.v2T([0.05, -0.03, 0.2, 0.01, -0.02, 0.015])))   # near the optimum
    picp = make_picp(ref_picp_mod, cam, data)
    for ci, (M, thr, pose) in enumerate(cases):
        uv, pw = synth_problem(rng, M, geo)
        cam.set_c_T_w(np.linalg.inv(pose))
        if M > 0:
            H, b, n, err = picp._ProjectiveICP__linearize(uv, pw, thr, pose)
        else:
            H, b, n, err = np.zeros((6, 6)), np.zeros(6), 0, np.inf
        step = picp._ProjectiveICP__one_step(uv, pw, pose, thr, 64.0)
        out.update({f"c{ci}_uv": uv, f"c{ci}_pw": pw, f"c{ci}_thr": thr, f"c{ci}_pose": pose,
                    f"c{ci}_H": np.array(H), f"c{ci}_b": np.array(b), f"c{ci}_n": n, f"c{ci}_err": err,
                    f"c{ci}_T1": np.array(step["T"]), f"c{ci}_done": step["computation_done"]})
    out["n_cases"] = len(cases)

    # hand-made edge cases: chi exactly on the threshold, points exactly on the image border,
    # a point with z == 0, and a stage-2 set smaller than min_inliers.
    pose = np.eye(4)
    cam.set_c_T_w(np.eye(4))
    pw = np.array([[0, 0, 1.0], [0, 0, 2.0], [1.0, 0, 1.0], [-320 / 180, 0, 1.0], [320 / 180, 0, 1.0],
                   [0, -240 / 180, 1.0], [0, 240 / 180, 1.0], [0.5, 0.5, 0.0], [0.1, 0.1, -1.0],
                   [0.25, 0.25, 4.0], [0.5, -0.25, 2.0], [0.0, 0.5, 8.0]])
    uv = np.array([[323.0, 244.0], [320.0, 245.0], [500.0, 240.0], [1.0, 240.0], [639.0, 240.0],
                   [320.0, 1.0], [320.0, 479.0], [0.0, 0.0], [0.0, 0.0],
                   [331.25, 251.25 + 3], [365.0 + 4, 217.5], [320.0 + 1, 251.25]])
    for ti, thr in enumerate((25.0, 24.999, 1e6)):
        H, b, n, err = picp._ProjectiveICP__linearize(uv, pw, thr, pose)
        out.update({f"edge{ti}_thr": thr, f"edge{ti}_H": np.array(H), f"edge{ti}_b": np.array(b),
                    f"edge{ti}_n": n, f"edge{ti}_err": err})
    out["edge_uv"], out["edge_pw"], out["n_edge"] = uv, pw, 3

    # full loops (schedules) on synthetic data, YAML parameters, several starts
    loops = [(300, np.eye(4), {}), (300, geo.v2T([0.3, 0.2, -0.4, 0.05, 0.08, -0.06]), {}),
             (40, np.eye(4), {"num_iterations": 30}),
             (5, np.eye(4), {}),
             (300, geo.v2T([3.0, -2.0, 1.0, 0.5, -0.4, 0.6]), {"num_iterations": 60})]
    for li, (M, pose, over) in enumerate(loops):
        uv, pw = synth_problem(rng, M, geo)
        lp = make_picp(ref_picp_mod, cam, data, **over)
        m = lp.get_map()
        app = rng.uniform(-1, 1, (M, 10))
        m["position"] = [p for p in pw]
        m["appearance"] = [list(a) for a in app]
        m["error"] = [0.0] * M
        meas = {"position": [list(p) for p in uv], "appearance": [list(a) for a in app]}
        r = run_icp_capture(ref_picp_mod, lp, cam, pose, meas, 0)
        out.update({f"loop{li}_uv": uv, f"loop{li}_pw": pw, f"loop{li}_pose": pose,
                    f"loop{li}_num_iterations": over.get("num_iterations", YAML["num_iterations"])})
        out.update({f"loop{li}_{k}": v for k, v in r.items()})
    out["n_loops"] = len(loops)
    np.savez_compressed(os.path.join(OUT, "ref_synthetic.npz"), **out)
    print("synthetic: cases", len(cases), "loops", len(loops))


# --------------------------------------------------------------------------
# 4. v2T grid, 5. association
# --------------------------------------------------------------------------


def gen_v2T(geo):
    rng = np.random.default_rng(1)
    v = np.concatenate([rng.uniform(-1, 1, (200, 6)) * np.array([1, 1, 1, 3.2, 3.2, 3.2]),
                        rng.uniform(-1, 1, (200, 6)) * 1e-3,
                        rng.uniform(-1, 1, (100, 6)) * 1e-7,
                        np.zeros((1, 6)), -np.zeros((1, 6))])
    T = np.array([geo.v2T(list(x)) for x in v])
    np.savez_compressed(os.path.join(OUT, "ref_v2T.npz"), v=v, T=T)


def gen_association(ref_da, cam):
    rng = np.random.default_rng(2)
    out = {}
    n1, n2 = 37, 53
    app2 = rng.uniform(-1, 1, (n2, 10))
    app1 = app2[rng.permutation(n2)[:n1]].copy()
    app1[5] = rng.uniform(-1, 1, 10)            # no match
    app1[6, 9] += 1e-12                          # differs in the last component only
    app2[11] = app2[10]                          # duplicate in set_2: two matches for one query
    app1[7] = app2[10]
    app1[8] = app1[7]                            # duplicate in set_1
    app1[9, 3] = 0.0
    app2[20] = app1[9]
    app2[20, 3] = -0.0                           # -0.0 == 0.0
    app1[10, 2] = np.nan
    app2[21] = app1[10]                          # NaN never equal
    pos1 = rng.uniform(0, 640, (n1, 2))
    z = rng.uniform(-2, 10, n2)
    pos2 = np.stack([rng.uniform(-3, 3, n2) * z, rng.uniform(-2, 2, n2) * z, z], axis=1)
    s1 = {"position": [list(p) for p in pos1], "appearance": [list(a) for a in app1]}
    s2 = {"position": [list(p) for p in pos2], "appearance": [list(a) for a in app2]}
    c_T_w = np.eye(4)
    c_T_w[:3, 3] = [0.1, -0.2, 0.3]
    cam.set_c_T_w(c_T_w)
    m0 = ref_da.data_association_on_appearance(s1, s2)
    m2 = ref_da.data_association_on_appearance(s1, s2, projection=2, camera=cam)
    # projection=1 projects set_1 positions, so give it 3-D points on that side
    s1b = {"position": s2["position"][:n1], "appearance": s1["appearance"]}
    m1 = ref_da.data_association_on_appearance(s1b, s2, projection=1, camera=cam)
    out.update(app1=app1, app2=app2, pos1=pos1, pos2=pos2, c_T_w=c_T_w,
               m0_p1=np.array(m0["points_1"]), m0_p2=np.array(m0["points_2"]), m0_app=np.array(m0["appearance"]),
               m2_p1=np.array(m2["points_1"]), m2_p2=np.array(m2["points_2"]),
               m2_proj=np.array(m2["projected_points_2"]), m2_app=np.array(m2["appearance"]),
               m1_p1=np.array(m1["points_1"]), m1_p2=np.array(m1["points_2"]),
               m1_proj=np.array(m1["projected_points_1"]), m1_app=np.array(m1["appearance"]))
    np.savez_compressed(os.path.join(OUT, "ref_association.npz"), **out)
    print("association: matches", len(m0["points_1"]), len(m1["points_1"]), len(m2["points_1"]))


# --------------------------------------------------------------------------
# 6. a camera matrix that is not of pinhole shape (skew, unequal focal lengths, a
#    non-zero entry in the last row): the reference reads whatever data/camera.dat holds
#    (src/Camera.py:20-24, integers), and the device code has a separate instantiation
#    for matrices with non-zero off-diagonal entries.
# --------------------------------------------------------------------------
GENERAL_K = [[180, 3, 320], [-2, 170, 240], [0, 0, 1]]


def gen_general_camera(ref_picp_mod, Camera, Data, geo):
    scratch = tempfile.mkdtemp(prefix="picp_golden_cam_")
    os.makedirs(os.path.join(scratch, "data"))
    for name in os.listdir(os.path.join(REF, "data")):
        if name != "camera.dat":
            os.symlink(os.path.join(REF, "data", name), os.path.join(scratch, "data", name))
    lines = open(os.path.join(REF, "data", "camera.dat")).read().split("\n")
    k = next(i for i, line in enumerate(lines) if line.startswith("camera matrix"))
    for r in range(3):
        lines[k + 1 + r] = " ".join(str(x) for x in GENERAL_K[r])
    with open(os.path.join(scratch, "data", "camera.dat"), "w") as fh:
        fh.write("\n".join(lines))
    back = os.getcwd()
    os.chdir(scratch)
    try:
        cam, data = Camera(), Data()
        K = np.array(cam.get_camera_matrix(), dtype=np.float64)
        assert np.array_equal(K, np.array(GENERAL_K, dtype=np.float64))
        rng = np.random.default_rng(42)
        out = {"K": K}
        T_true = geo.v2T([0.05, -0.03, 0.2, 0.01, -0.02, 0.015])
        c_T_w = np.linalg.inv(T_true)
        loops = [(400, np.eye(4), {}), (37, geo.v2T([0.2, -0.1, 0.1, 0.03, 0.02, -0.04]), {"num_iterations": 40})]
        for li, (M, pose, over) in enumerate(loops):
            z = rng.uniform(1.0, 20.0, M)
            pix = np.stack([rng.uniform(40, 600, M), rng.uniform(40, 440, M), np.ones(M)], axis=1)
            pw = (np.linalg.inv(K) @ pix.T).T * z[:, None]            # in the image under the identity pose
            pc = (c_T_w[:3, :3] @ pw.T).T + c_T_w[:3, 3]
            h = (K @ pc.T).T
            uv = h[:, :2] / h[:, 2:3] + rng.normal(0, 0.5, (M, 2))
            no = M // 10
            idx = rng.choice(M, no, replace=False)
            uv[idx] = np.stack([rng.uniform(0, 640, no), rng.uniform(0, 480, no)], axis=1)
            lp = make_picp(ref_picp_mod, cam, data, **over)
            m = lp.get_map()
            app = rng.uniform(-1, 1, (M, 10))
            m["position"] = [p for p in pw]
            m["appearance"] = [list(a) for a in app]
            m["error"] = [0.0] * M
            meas = {"position": [list(p) for p in uv], "appearance": [list(a) for a in app]}
            r = run_icp_capture(ref_picp_mod, lp, cam, pose, meas, 0)
            out.update({f"loop{li}_uv": uv, f"loop{li}_pw": pw, f"loop{li}_pose": pose,
                        f"loop{li}_num_iterations": over.get("num_iterations", YAML["num_iterations"])})
            out.update({f"loop{li}_{k}": v for k, v in r.items()})
        out["n_loops"] = len(loops)
        np.savez_compressed(os.path.join(OUT, "ref_general_camera.npz"), **out)
        print("general camera: loops", len(loops), [len(out[f"loop{li}_error"]) for li in range(len(loops))])
    finally:
        os.chdir(back)



# --------------------------------------------------------------------------
# 7. degenerate points (non-finite coordinates, camera plane, behind the camera)
# --------------------------------------------------------------------------


def gen_degenerate(ref_picp_mod, cam, data, geo):
    """A regular synthetic problem with points on and behind the camera plane and with infinite, NaN
    and huge coordinates or measurements mixed in, from a rotated start pose: pins what the
    reference's homogeneous division and its comparisons do with them (src/Camera.py:104-119)."""
    rng = np.random.default_rng(41)
    M = 3000
    uv, pw = synth_problem(rng, M, geo)
    big, tiny = 1e300, 5e-324
    odd_pw = np.array([[0.0, 0.0, 0.0], [1.0, 1.0, 0.0], [0.3, -0.2, -4.0], [np.nan, 0.0, 5.0], [0.0, np.inf, 5.0],
                       [0.0, 0.0, np.inf], [big, 0.0, big], [tiny, tiny, 1.0], [-big, big, 2.0], [-np.inf, 1.0, 1.0],
                       [0.0, 0.0, 1.0], [np.nan, np.nan, np.nan]])
    odd_uv = np.array([[320.0, 240.0]] * len(odd_pw))
    odd_uv[5] = [np.nan, 240.0]
    odd_uv[7] = [np.inf, -np.inf]
    where = np.linspace(3, M - 10, len(odd_pw)).astype(int)
    uv[where], pw[where] = odd_uv, odd_pw
    out = {"uv": uv, "pw": pw, "where": where}
    starts = [np.eye(4), geo.v2T([0.02, -0.01, 0.05, 0.02, -0.03, 0.01]), geo.v2T([0.3, 0.2, -0.4, 0.3, -0.2, 0.15])]
    app = rng.uniform(-1, 1, (M, 10))
    meas = {"position": [list(p) for p in uv], "appearance": [list(a) for a in app]}
    with np.errstate(all="ignore"):
        for li, pose in enumerate(starts):
            lp = make_picp(ref_picp_mod, cam, data, num_iterations=12)
            m = lp.get_map()
            m["position"] = [p for p in pw]
            m["appearance"] = [list(a) for a in app]
            m["error"] = [0.0] * M
            r = run_icp_capture(ref_picp_mod, lp, cam, pose, meas, 0)
            out[f"loop{li}_pose"] = pose
            out.update({f"loop{li}_{k}": v for k, v in r.items()})
    out["n_loops"] = len(starts)
    out["num_iterations"] = 12
    np.savez_compressed(os.path.join(OUT, "ref_degenerate.npz"), **out)
    print("degenerate: loops", len(starts), "iterations", [len(out[f"loop{li}_error"]) for li in range(len(starts))],
          "n_inside", [out[f"loop{li}_n_inside"].tolist() for li in range(len(starts))])


def gen_frame_summary(ref_picp_mod, cam, data):
    """What the reference's ``update`` prints for a frame (src/ProjectiveICP.py:117-141) and the Python
    number types of the values in its iteration lists, on three frames of the ground-truth-map run
    (a straight one, two turning ones).  ``update`` goes on into the OpenCV triangulation when the
    frame is valid; that call is replaced by one that keeps no point, which changes nothing that is
    printed.  Also a run with float parameters, where the lists hold floats throughout."""
    import contextlib
    import io
    gt_poses = data.get_trajectory_data_poses()
    C = np.array(cam.get_camera_transform())
    world = data.get_world_data()
    cam0_T_w = np.linalg.inv(gt_poses[0] @ C)
    wp = np.array(world["position"])
    map_pos = (cam0_T_w[:3, :3] @ wp.T).T + cam0_T_w[:3, 3]
    keep_nothing = lambda p0, p1, *a: ([], np.zeros(len(p0), dtype=bool))
    ref_picp_mod.triangulate_points = keep_nothing
    out = {}
    cases = [(0, {}), (16, {}), (70, {}), (16, dict(kernel_threshold=1000.0, min_kernel_threshold=10.0, dumping_factor=64.0))]
    for ci, (f, over) in enumerate(cases):
        picp = make_picp(ref_picp_mod, cam, data, **over)
        m = picp.get_map()
        m["position"] = [p for p in map_pos]
        m["appearance"] = [list(a) for a in world["appearance"]]
        m["error"] = [0.0] * len(m["position"])
        pose = cam0_T_w @ gt_poses[f] @ C
        picp._ProjectiveICP__current_pose = pose
        cam.set_c_T_w(np.linalg.inv(pose))
        captured = {}
        inner = picp._ProjectiveICP__projective_ICP

        def spy(meas, frame, inner=inner, captured=captured):
            r = inner(meas, frame)
            captured["results"] = r[2]
            return r
        picp._ProjectiveICP__projective_ICP = spy
        text = io.StringIO()
        cwd = os.getcwd()
        with contextlib.redirect_stdout(text):
            picp.update(f)                       # makes outputs/frame_XX under the scratch cwd
        assert os.getcwd() == cwd
        r = captured["results"]
        out[f"case{ci}_frame"] = f
        out[f"case{ci}_float_parameters"] = bool(over)
        out[f"case{ci}_text"] = np.array(text.getvalue())
        for k in ("error", "num_inliers", "kernel_threshold", "dumping_factor"):
            out[f"case{ci}_{k}"] = np.array(r[k], dtype=np.float64)
            out[f"case{ci}_{k}_is_int"] = np.array([isinstance(v, (int, np.integer)) for v in r[k]])
    out["n_cases"] = len(cases)
    np.savez_compressed(os.path.join(OUT, "ref_frame_summary.npz"), **out)
    print("frame summary:", [(int(out[f"case{c}_frame"]), len(out[f"case{c}_error"])) for c in range(len(cases))])


def main():
    os.makedirs(OUT, exist_ok=True)
    ref_picp_mod, Camera, Data, ref_da, ref_geo = import_reference()
    if "--only-general-camera" in sys.argv:
        gen_general_camera(ref_picp_mod, Camera, Data, ref_geo)
        return
    cam, data = Camera(), Data()
    if "--only-frame-summary" in sys.argv:
        gen_frame_summary(ref_picp_mod, cam, data)
        return
    if "--only-degenerate" in sys.argv:
        gen_degenerate(ref_picp_mod, cam, data, ref_geo)
        return
    pack_bundled(cam, data)
    gen_v2T(ref_geo)
    gen_association(ref_da, cam)
    gen_synthetic(ref_picp_mod, cam, data, ref_geo)
    gen_gtmap(ref_picp_mod, cam, data)
    gen_general_camera(ref_picp_mod, Camera, Data, ref_geo)
    gen_degenerate(ref_picp_mod, Camera(), Data(), ref_geo)
    gen_frame_summary(ref_picp_mod, Camera(), Data())


if __name__ == "__main__":
    main()
