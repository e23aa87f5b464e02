"""The C ABI from C++ through the adapter templates a TextSLAM maintainer compiles (adapter/tsba_gather.hpp), over plain structs
with the shape of TextSLAM's object graph (tests/cxx/mock_textslam.hpp).

CPU (no GPU needed): tests/cxx/abi_from_cxx builds the object graph of a synthetic problem -- keyframes with vObvPts /
vSceneObv2d[level] / vObvGoodPts / vFrameImg, map points and text planes with their host keyframes, observations and flags --,
runs the adapter's gather (rows B1 of SURVEY.md 8a: optimizer.cc:201-279, :1366-1557) and must reproduce every flat array.
GPU: the same binary then calls the entry point, scatters the result back into the object graph (row O2: optimizer.cc:292-326) and
the graph's parameters / flags must equal what the Python mirror gets from the same flat problem.
"""
import os
import struct
import subprocess
import numpy as np
import pytest

from textslam_amd import synth, abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cxx", "abi_from_cxx")


def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cxx")])


def _write_dump(path, P, state=abi.STATE_LOCAL):
    abi.write_dump(path, P, state)


def _read_out(path):
    out, b = {}, open(path, "rb").read()
    i = 0
    while i < len(b):
        nl, = struct.unpack_from("<I", b, i); i += 4
        name = b[i:i + nl].decode(); i += nl
        dt, cnt = struct.unpack_from("<BQ", b, i); i += 9
        ty = {0: np.float64, 1: np.int32, 2: np.uint8}[dt]
        out[name] = np.frombuffer(b, ty, cnt, i).copy(); i += cnt*np.dtype(ty).itemsize
    return out


def _observed_only(P, kf):
    """PoseOptim / InitBA address the landmarks through the observing frame: F.vObvPts[i] IS observation i (optimizer.cc:1126-1131),
    the planes are the frame's vObvText.  Re-index a synthetic problem that way: keep the points / planes keyframe `kf` observes, in
    observation order."""
    Q = P.copy()
    sel = Q.sobs_kf[0] == kf
    assert sel.all()
    pts = Q.sobs_pt[0].copy()
    assert len(np.unique(pts)) == len(pts)
    new_of = -np.ones(P.n_pt, np.int64); new_of[pts] = np.arange(len(pts))
    Q.rho, Q.pt_ray, Q.pt_host, Q.pt_host_Trw = P.rho[pts], P.pt_ray.reshape(-1, 2)[pts], P.pt_host[pts], P.pt_host_Trw.reshape(-1, 12)[pts]
    for l in range(P.n_levels):
        Q.sobs_pt[l] = new_of[P.sobs_pt[l]].astype(np.int32)
    tx = P.tobs_text.copy()
    assert len(np.unique(tx)) == len(tx) and (P.tobs_kf == kf).all()
    Q.theta = P.theta.reshape(-1, 3)[tx]; Q.text_host = P.text_host[tx]; Q.text_host_Twr = P.text_host_Twr.reshape(-1, 12)[tx]
    Q.text_box_ray = P.text_box_ray.reshape(-1, 8)[tx]
    Q.tobs_text = np.arange(len(tx), dtype=np.int32)
    for l in range(P.n_levels):
        off = P.tfeat_off[l]
        idx = np.concatenate([np.arange(off[j], off[j + 1]) for j in tx]) if len(tx) else np.zeros(0, np.int64)
        Q.tfeat_off[l] = np.concatenate([[0], np.cumsum([off[j + 1] - off[j] for j in tx])]).astype(np.int32)
        Q.tfeat_raw[l], Q.tfeat_uv[l], Q.tfeat_ref[l] = P.tfeat_raw[l][idx], P.tfeat_uv[l].reshape(-1, 2)[idx], P.tfeat_ref[l].reshape(-1, 8)[idx]
    return Q.normalise()


def _cases():
    out = {}
    out["local"] = (synth.tiny(seed=41, n_kf=6, n_pt=120, n_text=5), "local")
    out["global"] = (synth.config_global(n_kf=12, n_pt=300, band=5), "global")
    out["landmarker"] = (synth.landmark_refine(seed=9), "landmarker")
    out["pose"] = (_observed_only(synth.make_problem(1, 200, 6, 13, feats=(8, 6, 4), frozen_frac=1.0, n_out=4, max_targets=1, text_targets=1), 0), "pose")
    P = _observed_only(synth.init_pair(seed=5), 1)
    P.sgood[:] = 1; P.tobs_good[:] = 1; P.tfgood[:] = 1                       # InitBA has no flags
    out["init"] = (P, "init")
    out["theta"] = (synth.landmark_refine(seed=3, n_pt=0, n_text=2), "theta")
    return out


@pytest.mark.parametrize("name", ["local", "global", "landmarker", "pose", "init", "theta"])
def test_gather_reproduces_flat_problem(tmp_path, name):
    _build()
    P, mode = _cases()[name]
    dump, out = str(tmp_path / "p.bin"), str(tmp_path / "o.bin")
    _write_dump(dump, P)
    r = subprocess.run([EXE, dump, mode, out], capture_output=True, text=True, timeout=300)
    assert r.returncode in (0, 3), (r.returncode, r.stdout, r.stderr)
    assert "gather identical" in r.stdout


@pytest.mark.parametrize("case", ["small", "c4"])
def test_sliding_window_gather_with_the_segment_cache_equals_a_fresh_gather(tmp_path, case):
    """optimizer::LocalBundleAdjustment runs on a window that slid by one keyframe (tracking.cc:826-842).  The adapter keeps what every keyframe / text
    plane contributed to the last call (adapter/tsba_gather.hpp: GatherCache -- topology only; parameters and flags are read every call): over a six-window
    slide, after a keyframe gained an observation (keyframe::AddSceneObserv), after flags / parameters / a plane's state changed, and after invalidate(),
    EVERY flat array of the cached gather equals the fresh gather's element for element (so the plan the library builds from them is the same plan)."""
    _build()
    P = synth.config_c4() if case == "c4" else synth.make_problem(n_kf=9, n_pt=400, n_text=8, seed=77, feats=(16, 8, 6))
    dump = tmp_path / "p.bin"
    _write_dump(str(dump), P)
    r = subprocess.run([EXE, str(dump), "slide_check", str(tmp_path / "o.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "gathers identical with and without the cache" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["local", "global", "landmarker", "pose", "init", "theta"])
def test_cxx_solve_and_scatter_matches_python_mirror(tmp_path, name):
    from textslam_amd.optimizer import Optimizer
    _build()
    P, mode = _cases()[name]
    dump, out = str(tmp_path / "p.bin"), str(tmp_path / "o.bin")
    _write_dump(dump, P)
    r = subprocess.run([EXE, dump, mode, out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "gather identical" in r.stdout and "solve + scatter done" in r.stdout
    res = _read_out(out)
    g = Optimizer(0)
    G = P.copy()
    if mode == "theta":
        # the C++ side built its own flat problem (host first, observers, current frame): only the result is comparable
        assert res["cov_valid"][0] == 1 and np.all(np.linalg.eigvalsh(res["cov"].reshape(3, 3)) > 0)
        assert not np.array_equal(res["theta"], P.theta.reshape(-1, 3)[0])
        return
    if mode == "local":
        rep = g.LocalBundleAdjustment(G)
    elif mode == "global":
        rep = g.GlobalBA(G)
    elif mode == "landmarker":
        rep = g.OptimizeLandmarker(G)
    elif mode == "pose":
        rep = g.PoseOptim(G)
    else:
        rep = g.InitBA(G)
    assert res["iters"].tolist() == rep["iters"]
    np.testing.assert_allclose(res["cost1"], rep["cost1"], rtol=1e-5 if mode == "init" else 1e-9)
    # the graph stores poses as matrices: q -> R -> q costs a few ulp on the START pose of the C++ run; two-view InitBA amplifies that
    # along its free scale gauge (tests/test_gpu_parity.py::test_init_ba_first_linearisation_and_gauge_invariants)
    tol = 2e-6 if mode == "init" else 1e-9
    np.testing.assert_allclose(res["pose"], np.asarray(G.pose).reshape(-1), rtol=0, atol=tol)
    np.testing.assert_allclose(res["rho"], np.asarray(G.rho).reshape(-1), rtol=0, atol=tol)
    np.testing.assert_allclose(res["theta"], np.asarray(G.theta).reshape(-1), rtol=0, atol=tol)
    if mode in ("local", "landmarker", "pose"):
        assert np.array_equal(res["sgood"], G.sgood.reshape(-1)) and np.array_equal(res["tobs_good"], G.tobs_good.reshape(-1)) and np.array_equal(res["tfgood"], G.tfgood.reshape(-1))
    assert not np.array_equal(G.rho, P.rho) or mode == "pose"


# ---- loop closing: optimizer::OptimizeSim3 / OptimizeLoop through adapter/tsloop_gather.hpp (tests/cxx/loop_from_cxx.cpp) -------------------
LOOP_EXE = os.path.join(ROOT, "tests", "cxx", "loop_from_cxx")


def _put_records(path, items):
    rec = []
    for name, a, dt in items:
        a = np.ascontiguousarray(a, {0: np.float64, 1: np.int32, 2: np.uint8}[dt]).reshape(-1)
        rec.append(struct.pack("<I", len(name)) + name.encode() + struct.pack("<BQ", dt, a.size) + a.tobytes())
    with open(path, "wb") as f:
        f.write(b"".join(rec))


def _sim3_case():
    m = synth.sim3_matches(seed=6, n=200, outlier_frac=0.15)
    m["inliers"][::17] = 0                                                   # matches an earlier RANSAC pass already rejected
    m["sim0"][:4] *= 1.7                                                     # Sim12.r as handed over is not normalised: the gather does it (optimizer.cc:637-638)
    return m


def _loop_case():
    g = synth.pose_graph(seed=4, n_kf=40)
    n_loop = g["n_loop_edges"]
    rng = np.random.default_rng(11)
    g["pt_host"] = rng.integers(0, 40, 25).astype(np.int32); g["rho"] = rng.uniform(0.1, 0.6, 25)
    g["text_host"] = rng.integers(0, 40, 6).astype(np.int32); g["theta"] = rng.normal(0, 0.3, (6, 3))
    g["norm"] = (g["edge_i"][:-n_loop], g["edge_j"][:-n_loop]); g["loop"] = (g["edge_i"][-n_loop:], g["edge_j"][-n_loop:])
    return g


def _write_loop_dump(path, g):
    mScw = g["conn_sim"][list(g["conn_idx"]).index(g["kf_cur"])]
    _put_records(path, [("est", g["est"], 0), ("conn_idx", g["conn_idx"], 1), ("conn_sim", g["conn_sim"], 0), ("mScw", mScw, 0),
                        ("norm_i", g["norm"][0], 1), ("norm_j", g["norm"][1], 1), ("loop_i", g["loop"][0], 1), ("loop_j", g["loop"][1], 1),
                        ("ids", [g["kf_cur"], g["kf_loop"]], 1), ("pt_host", g["pt_host"], 1), ("rho", g["rho"], 0),
                        ("text_host", g["text_host"], 1), ("theta", g["theta"], 0)])


def _same_sim3_rows(a, b, tol):
    """rows (q | t | s): equal up to the sign of q (q and -q are one rotation; a pose that went through a rotation matrix comes back with w >= 0)"""
    a, b = np.asarray(a).reshape(-1, 8), np.asarray(b).reshape(-1, 8)
    sign = np.where(np.abs(a[:, :4] - b[:, :4]).max(1) <= np.abs(a[:, :4] + b[:, :4]).max(1), 1.0, -1.0)[:, None]
    np.testing.assert_allclose(a[:, :4], sign*b[:, :4], rtol=0, atol=tol)
    np.testing.assert_allclose(a[:, 4:], b[:, 4:], rtol=0, atol=tol)


def _check_loop_gather(res, g):
    """the arrays pack_loop built from the object graph against the flat problem of synth.pose_graph, connection by connection"""
    n = len(g["pose"])
    assert np.array_equal(res["g_fixed"], g["fixed"])
    _same_sim3_rows(res["g_pose"], g["pose"], 1e-12)
    key = lambda i, j: np.asarray(i, np.int64)*n + np.asarray(j, np.int64)
    n_loop = g["n_loop_edges"]; n_norm = len(g["edge_i"]) - n_loop
    assert len(res["g_edge_i"]) == len(g["edge_i"])
    gm = res["g_meas"].reshape(-1, 8)
    for sl in (slice(0, n_norm), slice(n_norm, None)):                       # the normal connections come first (optimizer.cc:788-858), each group in the map's order
        ko, kg = key(g["edge_i"][sl], g["edge_j"][sl]), key(res["g_edge_i"][sl], res["g_edge_j"][sl])
        assert len(np.unique(ko)) == len(ko) and np.array_equal(np.sort(ko), np.sort(kg))
        _same_sim3_rows(gm[sl][np.argsort(kg)], g["meas"][sl][np.argsort(ko)], 1e-12)
    # std::map<keyframe *, std::set<keyframe *>> order with the keyframes in one array: by first key, then by member
    assert np.array_equal(key(res["g_edge_i"][:n_norm], res["g_edge_j"][:n_norm]), np.sort(key(res["g_edge_i"][:n_norm], res["g_edge_j"][:n_norm])))


def test_loop_gather_reproduces_flat_problems(tmp_path):
    _build()
    m = _sim3_case()
    dump, out = str(tmp_path / "s.bin"), str(tmp_path / "so.bin")
    _put_records(dump, [("P1", m["P1"], 0), ("P2", m["P2"], 0), ("uv1", m["uv1"], 0), ("uv2", m["uv2"], 0), ("inliers", m["inliers"], 2), ("sim0", m["sim0"], 0), ("K", m["K"], 0)])
    r = subprocess.run([LOOP_EXE, dump, "sim3", out], capture_output=True, text=True, timeout=300)
    assert r.returncode in (0, 3), (r.returncode, r.stdout, r.stderr)
    assert "gather identical" in r.stdout
    g = _loop_case()
    dump, out = str(tmp_path / "l.bin"), str(tmp_path / "lo.bin")
    _write_loop_dump(dump, g)
    r = subprocess.run([LOOP_EXE, dump, "loop", out], capture_output=True, text=True, timeout=300)
    assert r.returncode in (0, 3), (r.returncode, r.stdout, r.stderr)
    assert "gather done" in r.stdout
    _check_loop_gather(_read_out(out), g)


@pytest.mark.gpu
def test_cxx_loop_closing_solve_and_scatter(tmp_path):
    """OptimizeSim3 / OptimizeLoop from C++: gather -> tsloop_* -> scatter into the object graph, against the Python mirror on the same flat
    arrays and against the reference's map update (optimizer.cc:884-956) done in numpy."""
    from textslam_amd.loop import LoopOptimizer
    _build()
    lo = LoopOptimizer(0)
    m = _sim3_case()
    dump, out = str(tmp_path / "s.bin"), str(tmp_path / "so.bin")
    _put_records(dump, [("P1", m["P1"], 0), ("P2", m["P2"], 0), ("uv1", m["uv1"], 0), ("uv2", m["uv2"], 0), ("inliers", m["inliers"], 2), ("sim0", m["sim0"], 0), ("K", m["K"], 0)])
    r = subprocess.run([LOOP_EXE, dump, "sim3", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "solve + scatter done" in r.stdout, (r.returncode, r.stdout, r.stderr)
    res = _read_out(out)
    n_in, sim, inl, rep = lo.OptimizeSim3(m["P1"], m["uv1"], m["P2"], m["uv2"], m["inliers"], m["sim0"], m["K"])
    assert res["meta"].tolist() == [n_in, rep["iters"], rep["termination"]] and 0 < n_in < len(inl)
    assert np.array_equal(res["inliers"].astype(bool), inl)
    q = sim[:4]/np.linalg.norm(sim[:4])
    np.testing.assert_allclose(res["sim"], np.concatenate([q, sim[4:]]), rtol=0, atol=1e-12)       # Sim12 = (q normalised, t, s), optimizer.cc:683-701
    np.testing.assert_allclose(res["cost1"][0], rep["cost1"], rtol=1e-12)

    g = _loop_case()
    dump, out = str(tmp_path / "l.bin"), str(tmp_path / "lo.bin")
    _write_loop_dump(dump, g)
    r = subprocess.run([LOOP_EXE, dump, "loop", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "solve + scatter done" in r.stdout, (r.returncode, r.stdout, r.stderr)
    res = _read_out(out)
    _check_loop_gather(res, g)
    x, rep = lo.OptimizeLoop(res["g_pose"].reshape(-1, 8), res["g_fixed"], res["g_edge_i"], res["g_edge_j"], res["g_meas"].reshape(-1, 8))
    assert res["meta"].tolist() == [rep["iters"], rep["termination"]] and rep["iters"] > 1
    np.testing.assert_allclose(res["pose_solved"].reshape(-1, 8), x, rtol=0, atol=1e-12)
    np.testing.assert_allclose(res["cost1"][0], rep["cost1"], rtol=1e-12)
    assert rep["cost1"] < 0.1*rep["cost0"]
    # the map update: T_cw = [R(q / |q|) | t / s], rho *= s(host), theta *= s(host)
    for k in range(len(x)):
        w, a, b, c = x[k, :4]/np.linalg.norm(x[k, :4])
        R = np.array([[1 - 2*(b*b + c*c), 2*(a*b - w*c), 2*(a*c + w*b)], [2*(a*b + w*c), 1 - 2*(a*a + c*c), 2*(b*c - w*a)], [2*(a*c - w*b), 2*(b*c + w*a), 1 - 2*(a*a + b*b)]])
        T = res["T34"].reshape(-1, 3, 4)[k]
        np.testing.assert_allclose(T[:, :3], R, rtol=0, atol=1e-12)
        np.testing.assert_allclose(T[:, 3], x[k, 4:7]/x[k, 7], rtol=0, atol=1e-12)
    np.testing.assert_allclose(res["rho"], g["rho"]*x[g["pt_host"], 7], rtol=1e-14)
    np.testing.assert_allclose(res["theta"].reshape(-1, 3), g["theta"]*x[g["text_host"], 7][:, None], rtol=1e-14)
    assert np.abs(x[:, 7] - 1).max() > 1e-3                                   # the scale drift was distributed over the loop


# ---- ORBextractor: the adapter class (adapter/tsorb_extractor_core.hpp) from C++ (tests/cxx/orb_from_cxx.cpp) ------------------------------
ORB_EXE = os.path.join(ROOT, "tests", "cxx", "orb_from_cxx")
ORB_ARGS = (1000, 1.2, 8, 20, 7)                                              # tracking.cc:36: ORBextractor(1000, 1.2, 8, 20, 7)


def _orb_dump(path, pad=0):
    from textslam_amd.orbextractor import synthetic_frame
    img = synthetic_frame(7)
    h, w = img.shape
    buf = np.zeros((h, w + pad), np.uint8); buf[:, :w] = img; buf[:, w:] = 255          # a cv::Mat ROI: step > cols, foreign bytes behind every row
    _put_records(path, [("img", buf, 2), ("wh", [w, h, w + pad], 1), ("args", [ORB_ARGS[0], ORB_ARGS[2], ORB_ARGS[3], ORB_ARGS[4]], 1), ("scale", [ORB_ARGS[1]], 0)])
    return img


def _check_orb_tables(res):
    """the constructor's tables as ORBextractor.cc:415-430 computes them: float entries, the scale factor a double member set from a float"""
    n = ORB_ARGS[2]; s = np.float64(np.float32(ORB_ARGS[1]))
    sf = np.ones(n, np.float32)
    for i in range(1, n):
        sf[i] = np.float32(np.float64(sf[i - 1])*s)
    sig = sf*sf
    expect = np.concatenate([sf, np.float32(1)/sf, sig, np.float32(1)/sig]).astype(np.float64)
    assert np.array_equal(res["tables"], expect)
    assert res["levels"][0] == n and res["scale_factor"][0] == np.float64(np.float32(ORB_ARGS[1]))


def test_orb_adapter_class_tables(tmp_path):
    _build()
    dump, out = str(tmp_path / "o.bin"), str(tmp_path / "oo.bin")
    _orb_dump(dump)
    r = subprocess.run([ORB_EXE, dump, out], capture_output=True, text=True, timeout=300)
    assert r.returncode in (0, 3), (r.returncode, r.stdout, r.stderr)
    _check_orb_tables(_read_out(out))


@pytest.mark.gpu
@pytest.mark.parametrize("pad", [0, 24])
def test_cxx_orb_extractor_matches_python_mirror(tmp_path, pad):
    """operator() of the adapter class from C++ (incl. an image with step > cols) against the Python mirror: bit-exact keypoints and descriptors"""
    from textslam_amd.orbextractor import ORBextractor
    _build()
    dump, out = str(tmp_path / "o.bin"), str(tmp_path / "oo.bin")
    img = _orb_dump(dump, pad)
    r = subprocess.run([ORB_EXE, dump, out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "extract done" in r.stdout, (r.returncode, r.stdout, r.stderr)
    res = _read_out(out)
    _check_orb_tables(res)
    kp, desc = ORBextractor(*ORB_ARGS)(img)
    assert res["n"][0] == len(kp) > 900
    assert np.array_equal(res["kp"].reshape(-1, 6).astype(np.float32), kp) and np.array_equal(res["desc"].reshape(-1, 32), desc)
