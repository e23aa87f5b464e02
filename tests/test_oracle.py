"""CPU tests of the oracle (the checker): self-consistency of the restated residual models, Ceres-LM behaviour and the
cv::fillPoly / CalTextinfo restatement.  The reference has no tests or golden vectors, so these pin the
oracle against independent formulations written here (finite differences, dense numpy solves, brute-force rasterisation)
and against the committed golden fixtures."""
import numpy as np
import pytest

from textslam_amd import synth, abi


def _fd_jacobian(oracle, P, o, level, eps=1e-6):
    """Central differences of the oracle's LITERAL functors w.r.t. the Ceres tangent space (Plus with half-angle)."""
    base = oracle.evaluate(P, o, level, jac=True)
    ns, nt = base["ns"], base["nt"]

    def resid(Q):
        return oracle.evaluate(Q, o, level, jac=False)["resid"]

    def quat_plus(q, d):
        n = np.linalg.norm(d)
        z = np.array([1.0, 0, 0, 0]) if n == 0 else np.concatenate([[np.cos(n)], np.sin(n) / n * d])
        return np.array([z[0]*q[0] - z[1]*q[1] - z[2]*q[2] - z[3]*q[3],
                         z[0]*q[1] + z[1]*q[0] + z[2]*q[3] - z[3]*q[2],
                         z[0]*q[2] - z[1]*q[3] + z[2]*q[0] + z[3]*q[1],
                         z[0]*q[3] + z[1]*q[2] - z[2]*q[1] + z[3]*q[0]])

    cols = {}
    for k in range(P.n_kf):
        for a in range(6):
            out = []
            for sgn in (+1, -1):
                Q = P.copy()
                pose = Q.pose.reshape(-1, 7)
                if a < 3:
                    d = np.zeros(3); d[a] = sgn * eps
                    pose[k, :4] = quat_plus(pose[k, :4], d)
                else:
                    pose[k, 4 + a - 3] += sgn * eps
                out.append(resid(Q))
            cols[("pose", k, a)] = (out[0] - out[1]) / (2 * eps)
    return base, cols


def test_scene_jacobian_matches_finite_differences(oracle_lib):
    P = synth.tiny(seed=5, n_kf=4, n_pt=40, n_text=0, n_levels=1)
    o = abi.options_local(); o.use_text = 0; o.filter_good = 0; o.n_passes = 1; o.levels[0] = 0
    base, cols = _fd_jacobian(oracle_lib, P, o, 0)
    ns = base["ns"]
    assert ns > 20
    J = base["jac_scene"]
    # map block -> (kf, host)
    kfs, hosts = [], []
    for s in range(P.sobs_kf[0].size):
        kf, pt = P.sobs_kf[0][s], P.sobs_pt[0][s]
        if P.pt_host[pt] == kf:
            continue
        kfs.append(kf); hosts.append(P.pt_host[pt])
    worst = 0.0
    for b in range(ns):
        for a in range(6):
            fd_t = cols[("pose", kfs[b], a)][2*b:2*b+2]
            worst = max(worst, np.max(np.abs(fd_t - J[b, :, a])) / (1 + np.max(np.abs(J[b, :, a]))))
            if hosts[b] >= 0:
                fd_h = cols[("pose", hosts[b], a)][2*b:2*b+2]
                worst = max(worst, np.max(np.abs(fd_h - J[b, :, 6 + a])) / (1 + np.max(np.abs(J[b, :, 6 + a]))))
    assert worst < 2e-7, worst
    # inverse depth column
    eps = 1e-7
    for j in range(P.n_pt):
        if P.pt_host[j] < 0:
            continue
        Qp, Qm = P.copy(), P.copy()
        Qp.rho[j] += eps; Qm.rho[j] -= eps
        fd = (oracle_lib.evaluate(Qp, o, 0, jac=False)["resid"] - oracle_lib.evaluate(Qm, o, 0, jac=False)["resid"]) / (2 * eps)
        for b in np.nonzero(fd[:2*ns].reshape(-1, 2).any(1))[0]:
            assert np.allclose(fd[2*b:2*b+2], J[b, :, 12], rtol=1e-5, atol=1e-5)


def test_text_analytic_vs_ceres_numeric_diff(oracle_lib):
    """Analytic bilinear Jacobian (the HIP path's choice) vs the reference's NumericDiffCostFunction<CENTRAL>."""
    P = synth.tiny()
    o = abi.options_local()
    oa = oracle_lib.evaluate(P, o, 0)
    o.text_jacobian = 1
    on = oracle_lib.evaluate(P, o, 0)
    Ja, Jn = oa["jac_text"], on["jac_text"]
    assert Ja.shape[0] > 50
    scale = np.abs(Ja).max()
    # away from pixel-edge crossings the two agree to ~1e-6 relative; edge-straddling stencils are rare
    rel = np.abs(Ja - Jn) / scale
    assert np.median(rel) < 1e-9
    assert np.mean(rel < 1e-6) > 0.995


def test_lm_converges_to_truth_noise_free(oracle_lib):
    P = synth.make_problem(8, 400, 0, seed=3, noise_px=0.0, outlier_frac=0.0, frozen_frac=0.1, n_levels=1)
    o = abi.options_local(); o.use_text = 0; o.n_passes = 1; o.levels[0] = 0; o.its[0] = 50
    Q = P.copy()
    rep = oracle_lib.solve(Q, o)
    assert rep["cost1"][0] < 1e-12 * max(1.0, rep["cost0"][0])
    assert np.abs(Q.pose - P.truth["pose"]).max() < 1e-8
    used = np.zeros(P.n_pt, bool)
    act = (P.sgood[P.sobs_flag[0]] == 1) & (P.pt_host[P.sobs_pt[0]] != P.sobs_kf[0])
    used[P.sobs_pt[0][act]] = True
    m = (P.pt_host >= 0) & used
    assert np.abs(Q.rho[m] / P.truth["rho"][m] - 1).max() < 1e-6


def test_reduced_system_matches_dense_normal_equations(oracle_lib):
    """Schur complement of the oracle vs an independent dense numpy assembly of J^T J from the oracle's Jacobians."""
    P = synth.tiny(seed=9, n_kf=5, n_pt=50, n_text=3)
    o = abi.options_local(); o.n_passes = 1; o.levels[0] = 0
    rs = oracle_lib.reduced_system(P, o, 0, 1e4)
    ev = oracle_lib.evaluate(P, o, 0)
    free = rs["free_idx"]
    nf = rs["nf"]
    # parameter layout: free poses (6 each), then rho of used points, then theta of used planes
    blocks = []
    for s in range(P.sobs_kf[0].size):
        kf, pt = int(P.sobs_kf[0][s]), int(P.sobs_pt[0][s])
        if not P.sgood[P.sobs_flag[0][s]] or P.pt_host[pt] == kf:
            continue
        blocks.append(("s", kf, int(P.pt_host[pt]), pt))
    for t in range(P.n_tobs):
        kf, j = int(P.tobs_kf[t]), int(P.tobs_text[t])
        if not P.tobs_good[t] or P.text_host[j] == kf:
            continue
        for f in range(P.tfeat_off[0][j], P.tfeat_off[0][j+1]):
            if P.tfgood[P.tobs_fgood_off[t] + P.tfeat_raw[0][f]]:
                blocks.append(("t", kf, int(P.text_host[j]), j))
    ns, nt = ev["ns"], ev["nt"]
    assert len(blocks) == ns + nt
    pts = sorted({b[3] for b in blocks if b[0] == "s" and b[2] >= 0})
    txs = sorted({b[3] for b in blocks if b[0] == "t" and b[2] >= 0})
    off_pt = {j: 6*nf + i for i, j in enumerate(pts)}
    off_tx = {j: 6*nf + len(pts) + 3*i for i, j in enumerate(txs)}
    n = 6*nf + len(pts) + 3*len(txs)
    H = np.zeros((n, n)); g = np.zeros(n)
    const = {k for k in range(P.n_kf) if free[k] < 0}
    for bi, b in enumerate(blocks):
        if b[0] == "s":
            r = ev["resid"][2*bi:2*bi+2]; J = ev["jac_scene"][bi]; delta = o.huber_scene; nl = 1
        else:
            k = bi - ns
            r = ev["resid"][2*ns + 8*k: 2*ns + 8*k + 8]; J = ev["jac_text"][k]; delta = o.huber_text; nl = 3
        if b[2] < 0 and b[1] in const:
            continue                                    # all parameter blocks constant
        s2 = r @ r
        w = np.sqrt(delta / np.sqrt(s2)) if s2 > delta**2 else 1.0
        r = r * w; J = J * w
        cols = []
        if free[b[1]] >= 0:
            cols += [(6*free[b[1]] + a, a) for a in range(6)]
        if b[2] >= 0 and free[b[2]] >= 0:
            cols += [(6*free[b[2]] + a, 6 + a) for a in range(6)]
        if b[2] >= 0:
            base = off_pt[b[3]] if b[0] == "s" else off_tx[b[3]]
            cols += [(base + a, 12 + a) for a in range(nl)]
        idx = np.array([c[0] for c in cols]); jc = np.array([c[1] for c in cols])
        Jc = J[:, jc]
        H[np.ix_(idx, idx)] += Jc.T @ Jc
        g[idx] += Jc.T @ r
    # Ceres LM damping in Jacobi-scaled coordinates, mapped back: Lambda_k = clamp(s^2 H_kk) / (radius s^2)
    d = np.diag(H).copy()
    sc = 1.0 / (1.0 + np.sqrt(d))
    lam = np.clip(sc**2 * d, o.min_diagonal, o.max_diagonal) / (1e4 * sc**2)
    Hd = H + np.diag(lam)
    npz = 6*nf
    S = Hd[:npz, :npz] - Hd[:npz, npz:] @ np.linalg.solve(Hd[npz:, npz:], Hd[npz:, :npz])
    gr = g[:npz] - Hd[:npz, npz:] @ np.linalg.solve(Hd[npz:, npz:], g[npz:])
    assert np.allclose(rs["S"], S, rtol=1e-9, atol=1e-9 * np.abs(S).max())
    assert np.allclose(rs["g"], gr, rtol=1e-9, atol=1e-9 * np.abs(gr).max())


def test_fillpoly_against_bruteforce(oracle_lib):
    """Convex quads: the restated fillPoly mask must contain every pixel strictly inside and no pixel farther than
    one pixel from the polygon (boundary pixels follow Bresenham / fixed-point rules)."""
    rng = np.random.default_rng(0)
    w, h = 64, 48
    for _ in range(50):
        c = np.array([rng.uniform(15, w - 15), rng.uniform(12, h - 12)])
        ang = np.sort(rng.uniform(0, 2*np.pi, 4))
        rad = rng.uniform(5, 11)                      # points on a circle: always a convex quad
        if np.min(np.diff(np.concatenate([ang, [ang[0] + 2*np.pi]]))) < 0.5:
            continue
        pts = np.stack([c[0] + rad*np.cos(ang), c[1] + rad*np.sin(ang)], 1).astype(int)
        m = oracle_lib.fillpoly4(w, h, pts)
        ys, xs = np.mgrid[0:h, 0:w]
        inside = np.ones((h, w), bool); dist_out = np.zeros((h, w))
        sign = None
        for i in range(4):
            a, b = pts[i], pts[(i + 1) % 4]
            e = b - a
            if not e.any():
                continue
            cr = (e[0]*(ys - a[1]) - e[1]*(xs - a[0])) / np.hypot(*e)
            if sign is None:
                cen = pts.mean(0)
                sign = 1.0 if (e[0]*(cen[1] - a[1]) - e[1]*(cen[0] - a[0])) >= 0 else -1.0
            inside &= cr*sign > 0.75
            dist_out = np.maximum(dist_out, -cr*sign)
        assert np.all(m[inside] == 1)
        assert np.all(dist_out[m == 1] <= 1.0 + 1e-9)
        for x, y in pts:
            assert m[y, x] == 1


def test_musigma_matches_numpy_on_mask(oracle_lib):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (48, 64)).astype(np.uint8)
    corners = np.array([[10.3, 8.7], [50.9, 10.2], [48.1, 35.5], [12.6, 30.0]])
    ok, mu, sg = oracle_lib.musigma(img, corners)
    m = oracle_lib.fillpoly4(64, 48, corners.astype(int)).astype(bool)
    x0, x1 = int(np.floor(corners[:, 0].min())), int(np.ceil(corners[:, 0].max()))
    y0, y1 = int(np.floor(corners[:, 1].min())), int(np.ceil(corners[:, 1].max()))
    box = np.zeros_like(m); box[y0:y1+1, x0:x1+1] = True
    vals = img[m & box].astype(float)
    assert ok == 1
    assert mu == pytest.approx(vals.mean(), rel=1e-14)
    assert sg == pytest.approx(vals.std(ddof=1), rel=1e-13)
    # degenerate: quad outside the image -> invalid, sigma = 0
    ok, mu, sg = oracle_lib.musigma(img, corners + 500)
    assert ok == 0 and sg == 0


def test_outlier_flags_and_gauge(oracle_lib):
    P = synth.tiny(seed=21, n_kf=6, n_pt=200, n_text=4)
    o = abi.options_local()
    Q = P.copy()
    rep = oracle_lib.solve(Q, o)
    # the first three participating keyframes are constant (STATE == LOCAL, optimizer.cc:1571-1588)
    assert np.array_equal(Q.pose.reshape(-1, 7)[:3], P.pose.reshape(-1, 7)[:3])
    assert not np.array_equal(Q.pose.reshape(-1, 7)[3:], P.pose.reshape(-1, 7)[3:])
    # flags only ever go from good to bad, and gross outliers are caught
    assert np.all(Q.sgood <= P.sgood)
    assert sum(rep["n_bad_scene"]) > 0
    # frozen landmarks are untouched
    assert np.array_equal(Q.rho[P.pt_host < 0], P.rho[P.pt_host < 0])
    assert np.array_equal(Q.theta[P.text_host < 0], P.theta[P.text_host < 0])


def test_empty_and_ragged_inputs(oracle_lib):
    # no text at all, one keyframe, no observations at some levels
    P = synth.make_problem(1, 50, 0, seed=2, frozen_frac=1.0, n_levels=1)
    o = abi.options_pose(); o.use_text = 0; o.n_passes = 1; o.levels[0] = 0
    Q = P.copy(); rep = oracle_lib.solve(Q, o)
    assert rep["n_tblock"] == [0] and rep["n_sblock"][0] > 0
    # a window whose flags are all bad: nothing to optimise, parameters unchanged
    P2 = synth.tiny(seed=4)
    P2.sgood[:] = 0; P2.tobs_good[:] = 0
    Q2 = P2.copy(); rep2 = oracle_lib.solve(Q2, abi.options_local())
    assert rep2["iters"] == [0, 0, 0]
    assert np.array_equal(Q2.pose, P2.pose)


def test_landmark_only_and_theta_covariance(oracle_lib):
    """R5 / R9: every pose constant.  theta covariance = inverse of the theta block of J^T J (ceres::Covariance)."""
    P = synth.landmark_refine(seed=3)
    o = abi.options_theta(); o.use_text = 1
    Q = P.copy()
    rep = oracle_lib.solve(Q, o)
    assert np.array_equal(Q.pose, P.pose)                                  # nothing but landmarks moves
    assert not np.array_equal(Q.theta, P.theta)
    assert all(c1 <= c0 for c0, c1 in zip(rep["cost0"], rep["cost1"]))
    rc, cov = oracle_lib.theta_cov(Q, o, 0, 1)
    assert rc == 0
    ev = oracle_lib.evaluate(Q, o, 0)
    # rebuild the information matrix of plane 1 from the per-block Jacobians
    k = 0; V = np.zeros((3, 3))
    for t in range(Q.n_tobs):
        kf, j = int(Q.tobs_kf[t]), int(Q.tobs_text[t])
        if Q.text_host[j] == kf:
            continue
        nf = Q.tfeat_off[0][j + 1] - Q.tfeat_off[0][j]
        for _ in range(nf):
            if j == 1:
                Jl = ev["jac_text"][k][:, 12:15]; V += Jl.T @ Jl
            k += 1
    assert k == ev["nt"]
    assert np.allclose(cov, np.linalg.inv(V), rtol=1e-9)


def test_init_ba_style_problem(oracle_lib):
    """R4 / R8 (optimizer::InitBA): host keyframe constant at its pose, second keyframe + rho + theta free, 4 levels."""
    P = synth.init_pair(seed=5)
    o = abi.options_init()
    Q = P.copy()
    rep = oracle_lib.solve(Q, o)
    assert rep["n_passes"] == 4 and np.array_equal(Q.pose.reshape(-1, 7)[0], P.pose.reshape(-1, 7)[0])
    err0 = np.abs(P.pose.reshape(-1, 7)[1, 4:] - P.truth["pose"][1, 4:]).max()
    err1 = np.abs(Q.pose.reshape(-1, 7)[1, 4:] - P.truth["pose"][1, 4:]).max()
    assert rep["cost1"][-1] < rep["cost0"][0] and err1 < err0


def test_label_image_against_point_in_polygon(oracle_lib):
    oracle = oracle_lib
    """oracle.label_image (ShowBAReproj_TextBox -> TextBoxWithFill): background -1, labels = rank of the text observation in
    the keyframe, later quads on top; checked against an independent even-odd point-in-polygon test away from the edges."""
    P = synth.tiny(seed=23, n_kf=4, n_pt=30, n_text=5, text_targets=3)
    lvl = 0
    s = P.struct()
    w, h = int(s.img_w[lvl]), int(s.img_h[lvl])
    K = np.array([s.K[0], s.K[1], s.K[2], s.K[3]])
    for kf in range(P.n_kf):
        lab = oracle.label_image(P, kf, lvl)
        assert lab.shape == (h, w) and lab.dtype == np.float32
        obs = [t for t in range(P.n_tobs) if P.tobs_kf[t] == kf]
        assert set(np.unique(lab)).issubset({-1.0} | {float(r) for r in range(len(obs))})
        # corners of every observed plane with plain numpy
        quads = []
        for t in obs:
            j = int(P.tobs_text[t]); host = int(P.text_host[j])
            def Rt(p):
                q = p[:4]/np.linalg.norm(p[:4]); w_, x, y, z = q
                R = np.array([[1-2*(y*y+z*z), 2*(x*y-w_*z), 2*(x*z+w_*y)], [2*(x*y+w_*z), 1-2*(x*x+z*z), 2*(y*z-w_*x)], [2*(x*z-w_*y), 2*(y*z+w_*x), 1-2*(x*x+y*y)]])
                return R, p[4:7]
            Rc, tc = Rt(P.pose[kf])
            if host >= 0:
                Rr, tr = Rt(P.pose[host]); Rcr = Rc @ Rr.T; tcr = tc - Rcr @ tr
            else:
                T = P.text_host_Twr[j].reshape(3, 4); Rcr = Rc @ T[:, :3]; tcr = Rc @ T[:, 3] + tc
            th = P.theta[j]; c = []
            for b in range(4):
                m = np.array([P.text_box_ray[j, b, 0], P.text_box_ray[j, b, 1], 1.0])
                X = Rcr @ m/(-(m @ th)) + tcr
                c.append((K[0]*X[0]/X[2] + K[2], K[1]*X[1]/X[2] + K[3]))
            quads.append(np.trunc(np.array(c)))
        # sample pixels: inside quad r (even-odd) and at least 1.5 px from all its edges => label >= r; far outside all => -1
        rng = np.random.default_rng(kf)
        for _ in range(400):
            x, y = int(rng.integers(0, w)), int(rng.integers(0, h))
            def inside_margin(q):
                inside = False; dmin = 1e9
                for i in range(4):
                    x0, y0 = q[i - 1]; x1, y1 = q[i]
                    if (y0 > y) != (y1 > y) and x < (x1 - x0)*(y - y0)/(y1 - y0 + 1e-30) + x0:
                        inside = not inside
                    d = np.array([x1 - x0, y1 - y0]); L2 = d @ d
                    tt = 0.0 if L2 == 0 else np.clip(((x - x0)*d[0] + (y - y0)*d[1])/L2, 0, 1)
                    dmin = min(dmin, np.hypot(x - (x0 + tt*d[0]), y - (y0 + tt*d[1])))
                return inside, dmin
            top = -1
            ambiguous = False
            for r, q in enumerate(quads):
                ins, dm = inside_margin(q)
                if dm < 1.5: ambiguous = True
                elif ins: top = r
            if not ambiguous:
                assert lab[y, x] == float(top), (kf, x, y, lab[y, x], top)


def test_lm_optimum_matches_scipy_least_squares(oracle_lib):
    """Independent optimiser on the oracle's literal residual functors: scipy's trust-region least squares (a different LM
    implementation, dense finite-difference Jacobian, ambient quaternion parameters) reaches the same minimum as the oracle's
    Ceres-style LM with the analytic tangent-space Jacobians -- noisy scene observations, hosts inside and outside the window,
    loss switched off (scipy's robust losses act per scalar residual, Ceres' per block).  Scene blocks only: the solver holds
    mu / sigma of a text pair fixed within a pass (optimizer.cc:1490-1504) while a plain re-evaluation recomputes them."""
    from scipy.optimize import least_squares
    P = synth.tiny(seed=11, n_kf=5, n_pt=80, n_text=0)
    o = abi.options_local(); o.use_text = 0
    o.n_passes = 1; o.levels[0] = 0; o.its[0] = 60
    o.huber_scene = 1e9; o.huber_text = 1e9; o.outlier_scene = 0; o.outlier_text = 0
    o.function_tolerance = 1e-15; o.parameter_tolerance = 1e-15
    Q = P.copy()
    oracle_lib.solve(Q, o)
    free = [int(k) for k in np.nonzero(oracle_lib.reduced_system(P, o, 0, 1e4)["free_idx"] >= 0)[0]]      # gauge as the solver fixes it
    assert 1 <= len(free) < P.n_kf

    pts = np.nonzero(P.pt_host >= 0)[0]; txs = np.nonzero(P.text_host >= 0)[0]       # frozen-host landmarks are constants (R3 / R7)

    def unpack(x, R):
        n = 0
        for k in free:
            R.pose[k] = x[n:n+7]; n += 7
        R.rho[pts] = x[n:n+pts.size]; n += pts.size
        R.theta[txs] = x[n:n+3*txs.size].reshape(-1, 3)

    def pack(R):
        return np.concatenate([np.concatenate([R.pose[k] for k in free]), R.rho[pts], R.theta[txs].reshape(-1)])

    W = P.copy()

    def resid(x):
        unpack(x, W)
        return oracle_lib.evaluate(W, o, 0, jac=False)["resid"].copy()

    c_oracle = 0.5*np.sum(resid(pack(Q))**2)
    sol = least_squares(resid, pack(P), method="trf", x_scale="jac", xtol=1e-14, ftol=1e-14, gtol=1e-12, max_nfev=400)
    c_scipy = 0.5*np.sum(sol.fun**2)
    c0 = 0.5*np.sum(resid(pack(P))**2)
    assert c_oracle < 0.9*c0                                             # the problem is not trivial
    assert abs(c_oracle - c_scipy) <= 2e-6*c_scipy, (c0, c_oracle, c_scipy)
    # and the oracle's solution is stationary for the independent optimiser
    sol2 = least_squares(resid, pack(Q), method="trf", x_scale="jac", xtol=1e-14, ftol=1e-14, gtol=1e-12, max_nfev=100)
    assert 0.5*np.sum(sol2.fun**2) >= c_oracle*(1 - 1e-7)


def test_band_storage_solver_matches_dense(oracle_lib):
    """Maps of thousands of keyframes go through band storage of H_pp / S and a band Cholesky (the dense reduced system would be 7 GB at
    5000 keyframes -- what bench.py's cpu_baseline runs); forced here at 120 keyframes: same LM trajectory, same parameters to the bit."""
    P = synth.config_global(n_kf=120, n_pt=4000, band=8)
    o = abi.options_global(); o.its[0] = 5
    A, B = P.copy(), P.copy()
    ra = oracle_lib.solve(A, o)
    oracle_lib.set_band_threshold(10)
    try:
        rb = oracle_lib.solve(B, o)
    finally:
        oracle_lib.set_band_threshold(400)
    assert ra["iters"] == rb["iters"] and ra["accepted"] == rb["accepted"] and ra["cost1"] == rb["cost1"]
    assert np.array_equal(A.pose, B.pose) and np.array_equal(A.rho, B.rho)
    assert ra["accepted"][0] >= 3 and ra["cost1"][0] < ra["cost0"][0]


# ---- block-sparse normal equations + pluggable exact linear solver (the oracle at thousands of keyframes: tests/test_gpu_fullsize_oracle.py)
def test_block_sparse_reduced_system_equals_the_dense_one():
    """tsba_oracle_reduced_blocks against tsba_oracle_reduced_system on a map with long-range points: the same sums in the same order, so the
    lower block triangle agrees to the last bit; g and the cost are identical."""
    from textslam_amd import synth, abi
    import oracle
    P = synth.config_global(n_kf=60, n_pt=1500, band=8, far_frac=0.03)
    o = abi.options_global()
    rs = oracle.reduced_system(P, o, 0, o.initial_radius)
    rb = oracle.reduced_blocks(P, o, 0, o.initial_radius)
    assert rb["nf"] == rs["nf"] and np.array_equal(rb["free_idx"], rs["free_idx"])
    assert rb["cost"] == rs["cost"] and np.array_equal(rb["g"], rs["g"])
    n = 6*rb["nf"]
    assert np.all(rb["br"] >= rb["bc"]) and (rb["br"] - rb["bc"]).max() > 12          # long-range blocks are there
    A = oracle.blocks_to_sparse(n, rb["br"], rb["bc"], rb["val"]).toarray()
    assert np.array_equal(np.tril(A), np.tril(rs["S"]))
    assert np.array_equal(A, A.T)
    nz = np.zeros((rb["nf"], rb["nf"]), bool); nz[rb["br"], rb["bc"]] = True
    dense_nz = np.abs(rs["S"]).reshape(rb["nf"], 6, rb["nf"], 6).max(axis=(1, 3)) > 0
    assert np.array_equal(np.tril(dense_nz) & ~nz, np.zeros_like(nz))                 # every non-zero block of the dense matrix is in the list


@pytest.mark.parametrize("kw,expect", [(dict(n_kf=200, n_pt=4000, band=8, far_frac=0.03), "gmres"), (dict(n_kf=200, n_pt=4000, band=8, closures=2), "banded"),
                                        (dict(n_kf=200, n_pt=4000, band=8, loop=True), "banded")])
def test_lm_loop_on_block_sparse_equations_with_a_plugged_solver(kw, expect):
    """The LM loop through block-sparse storage + oracle.sparse_solver (banded Cholesky under a reverse Cuthill-McKee order, or GMRES with the
    band's LU as preconditioner where no order gives a band) reproduces the built-in dense Cholesky path: same decisions, costs to 1e-12."""
    from textslam_amd import synth, abi
    import oracle
    P = synth.config_global(**kw)
    o = abi.options_global(); o.its[0] = 8
    R1 = P.copy(); rep1, tr1 = oracle.solve_traced(R1, o)
    oracle.SOLVER_LOG.clear()
    oracle.set_sparse_solver(lambda A, b: oracle.sparse_solver(A, b, band_limit=150 if expect == "gmres" else 700))
    try:
        R2 = P.copy(); rep2, tr2 = oracle.solve_traced(R2, o)
    finally:
        oracle.set_sparse_solver(None)
    assert oracle.SOLVER_LOG and all(e[0] == expect for e in oracle.SOLVER_LOG), oracle.SOLVER_LOG[:3]
    assert rep1["iters"] == rep2["iters"] and rep1["accepted"] == rep2["accepted"] and rep1["termination"] == rep2["termination"]
    assert np.array_equal(tr1[0][:, 3], tr2[0][:, 3])
    np.testing.assert_allclose(tr1[0][:, 0], tr2[0][:, 0], rtol=1e-12)
    np.testing.assert_allclose(R1.pose, R2.pose, rtol=0, atol=1e-9)
    assert len(tr1[0]) == rep1["iters"][0] and tr1[0][-1][3] in (0.0, 1.0, 2.0)
