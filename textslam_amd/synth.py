"""Seeded synthetic BA problems: the workloads of bench.py and the parity tests.

There is no dataset and no Ceres / OpenCV in the image, so every configuration is generated:
camera K = yaml/GeneralMotion.yaml:12-15, 640x480; keyframes on a gentle arc; inverse-depth
points hosted in their first observer; planar text patches with a band-limited texture rendered
into every keyframe through the true plane-induced homography, so that photometric residuals
vanish at ground truth.  Pure numpy, deterministic for a given seed.
"""
import numpy as np
from .abi import BAProblem, MAX_LEVELS

K_GENERAL_MOTION = np.array([384.396254546, 382.825746531, 315.635886103, 249.182929809])
SEED = 20240926
TAP_DX = np.array([0, 2, 1, 0, -1, -2, -1, 0], np.float64)      # tool.cc:1550-1557 (INTERVAL8)
TAP_DY = np.array([0, 0, -1, -2, -1, 0, 1, 2], np.float64)
W, H = 640, 480


def _rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def _rodrigues(v):
    th = np.linalg.norm(v)
    if th < 1e-15:
        return np.eye(3)
    k = v / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def _R_to_q(R):
    """Eigen's Quaternion(Matrix3) (w,x,y,z), normalised -- what optimizer.cc:84-90 feeds Ceres."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        q = np.array([w, (R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        v = np.zeros(3)
        v[i] = 0.5 * s
        s = 0.5 / s
        w = (R[k, j] - R[j, k]) * s
        v[j] = (R[j, i] + R[i, j]) * s
        v[k] = (R[k, i] + R[i, k]) * s
        q = np.array([w, v[0], v[1], v[2]])
    return q / np.linalg.norm(q)


def _cam(k, step=0.1, yaw_deg=0.35):
    """T_cw of camera k (k may be negative: hosts that already left the window)."""
    c = np.array([step * k, 0.02 * np.sin(0.3 * k), 0.03 * np.cos(0.2 * k)])
    Rwc = _rot_y(np.deg2rad(yaw_deg) * k)
    Rcw = Rwc.T
    return Rcw, -Rcw @ c


def _cam_ring(k, n, step=0.1):
    """T_cw of camera k of a CLOSED trajectory: n cameras on a circle (arc length `step` between neighbours) looking outward, so that
    the last keyframes see what the first ones saw -- the map right after a loop closure, which is when TextSLAM runs GlobalBA
    (loopClosing.cc:589)."""
    phi = 2.0*np.pi*k/n
    radius = step*n/(2.0*np.pi)
    c = np.array([radius*np.sin(phi), 0.02*np.sin(0.3*k), radius*np.cos(phi)])
    Rwc = _rot_y(phi)                       # optical axis (0,0,1) -> (sin phi, 0, cos phi): outward
    Rcw = Rwc.T
    return Rcw, -Rcw @ c


def _cam_lollipop(k, n, k0, step=0.1):
    """T_cw of camera k of a trajectory that runs straight for k0 keyframes and then once around a circle back to keyframe k0: a loop closure
    with a TAIL before the loop (the usual case: the loop does not start at the first keyframe)."""
    if k >= k0:
        return _cam_ring(k - k0, n - k0, step)
    R0, t0 = _cam_ring(0, n - k0, step)
    c0 = -R0.T @ t0                         # centre of the loop's first camera; the tail arrives along the tangent (+x at phi = 0)
    c = c0 + np.array([-(k0 - k)*step, 0.02*np.sin(0.3*k) - 0.02*np.sin(0.0), 0.0])
    return R0, -R0 @ c


def _lobes(n, K):
    """Keyframe ranges of a trajectory with K separate loop closures: a straight tail, then K loops, each closing on its own first keyframe,
    joined by straight connectors.  Returns [(start, end)] of the loops."""
    t = m = max(30, n//(4*K + 1))
    l = (n - t - (K - 1)*m)//K
    return [(t + i*(l + m), t + i*(l + m) + l) for i in range(K)]


def _cam_lobes(k, n, K, step=0.1):
    """T_cw of camera k of a trajectory that closes K separate loops (several loop closures in one session: the reference runs GlobalBA after
    each, loopClosing.cc:587-591, on a map that keeps the earlier ones): tail -> loop 0 -> connector -> loop 1 -> ...; every loop is a circle
    tangent to the base line at its junction, cameras looking outward as in _cam_ring; the connectors leave the junction along the tangent."""
    lobes = _lobes(n, K)
    l = lobes[0][1] - lobes[0][0]
    radius = step*l/(2.0*np.pi)
    m = lobes[1][0] - lobes[0][1] if K > 1 else 0
    xj = lambda i: i*m*step                                   # junction of loop i on the base line (x axis, z = radius)
    wob = 0.02*np.sin(0.3*k)
    R0, _ = _cam_ring(0, l, step)
    for i, (s, e) in enumerate(lobes):
        if s <= k < e:
            Rcw, t = _cam_ring(k - s, l, step)
            c = -Rcw.T @ t + np.array([xj(i), 0.0, 0.0])
            c[1] = wob
            return Rcw, -Rcw @ c
        if k < s:                                            # tail (i == 0) or the connector before loop i
            c = np.array([xj(i) - (s - k)*step + (0.5*step if i > 0 else 0.0), wob, radius])
            return R0, -R0 @ c
    c = np.array([xj(K - 1) + (k - lobes[-1][1] + 0.5)*step, wob, radius])      # after the last loop
    return R0, -R0 @ c


def _bilinear(img, u, v):
    h, w = img.shape
    uf, vf = np.floor(u).astype(int), np.floor(v).astype(int)
    ok = (uf >= 0) & (vf >= 0) & (np.ceil(u) < w) & (np.ceil(v) < h)
    uf0, vf0 = np.clip(uf, 0, w - 2), np.clip(vf, 0, h - 2)
    a, b = u - uf, v - vf
    I = ((1 - a) * (1 - b) * img[vf0, uf0] + a * (1 - b) * img[vf0, uf0 + 1]
         + (1 - a) * b * img[vf0 + 1, uf0] + a * b * img[vf0 + 1, uf0 + 1])
    return np.where(ok, I, 0.0)


def _quad_stats(img, corners):
    """mean / sample-std of the pixels whose centres lie inside a convex quad (host-side reference statistics)."""
    h, w = img.shape
    x0, x1 = int(max(0, np.floor(corners[:, 0].min()))), int(min(w - 1, np.ceil(corners[:, 0].max())))
    y0, y1 = int(max(0, np.floor(corners[:, 1].min()))), int(min(h - 1, np.ceil(corners[:, 1].max())))
    if x1 < x0 or y1 < y0:
        return 0.0, 0.0
    xs, ys = np.meshgrid(np.arange(x0, x1 + 1), np.arange(y0, y1 + 1))
    inside = np.ones(xs.shape, bool)
    sign = None
    for i in range(4):
        a, b = corners[i], corners[(i + 1) % 4]
        cr = (b[0] - a[0]) * (ys - a[1]) - (b[1] - a[1]) * (xs - a[0])
        if sign is None:
            sign = 1.0 if np.sum(cr > 0) >= np.sum(cr < 0) else -1.0
        inside &= (cr * sign >= 0)
    vals = img[y0:y1 + 1, x0:x1 + 1][inside].astype(np.float64)
    if vals.size < 2:
        return 0.0, 0.0
    return float(vals.mean()), float(vals.std(ddof=1))


def _pyr_down(img):
    h, w = img.shape[-2] // 2 * 2, img.shape[-1] // 2 * 2
    a = img[..., :h, :w].astype(np.uint16)
    s = a[..., 0::2, 0::2] + a[..., 0::2, 1::2] + a[..., 1::2, 0::2] + a[..., 1::2, 1::2]
    return ((s + 2) // 4).astype(np.uint8)


class _Plane:
    pass


def make_problem(n_kf=20, n_pt=5000, n_text=100, seed=SEED, feats=(64, 24, 12), max_targets=5, text_targets=5,
                 frozen_frac=0.1, outlier_frac=0.05, noise_px=0.5, perturb=True, n_levels=3,
                 band=None, far_frac=0.0, n_out=3, rot_deg=0.5, trans_m=0.02, lm_rel=0.05, self_obs=True, n_fixed=3, kf_initial=None, loop=False, loop_at=0, closures=0,
                 drop_outlier_points=False, perturb_in_camera=False):
    """Build a synthetic window (local BA / pose-only / global BA depending on the arguments).

    n_kf == 1 with frozen_frac == 1 gives the pose-only problem (every landmark hosted outside).
    n_text == 0 skips image synthesis (global BA, scene points only, as the reference's GlobalBA).
    drop_outlier_points: the map as the reference hands it to GlobalBA -- optimizer::GlobalBA takes map::GetAllMapPoints(false)
    (src/optimizer.cc:337-341, src/map.cc:36-47: only points with FLAG_BAD == false) and every local BA has set FLAG_BAD on a point with ANY
    observation it flagged (tracking::mpPtsCondUpdate, src/tracking.cc:2215-2230): every point with a gross-outlier observation or an
    observation whose good flag is already down leaves the problem with all its observations (PyrGlobalBA skips them: IdxRho < 0,
    optimizer.cc:1738-1743).  The random stream is the same with and without it.
    perturb_in_camera: the pose perturbation as a motion of the CAMERA, T_cw' = (dR, dt) T_cw: the camera centre moves by ~trans_m wherever the keyframe is.  The
    default perturbs R_cw and t_cw separately, which turns a camera 500 m from the origin around the ORIGIN (0.2 degrees = 1.7 m at keyframe 5000 against a
    step of 0.1 m between keyframes): a start no loop correction would leave behind, and the reason the 5000-keyframe chains need hundreds of LM iterations.
    """
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = K_GENERAL_MOTION
    P = BAProblem()
    P.K = K_GENERAL_MOTION.copy()
    P.n_levels = n_levels if n_text > 0 else 1
    band = band or n_kf
    # cameras: window 0..n_kf-1, outside hosts -1..-n_out
    cams = {k: ((_cam_lollipop(k, n_kf, loop_at) if loop_at > 0 else _cam_ring(k, n_kf)) if loop else _cam(k)) for k in range(-n_out, n_kf)}
    lobes = _lobes(n_kf, closures) if closures > 0 else []
    if closures > 0:                                          # several separate loop closures (frozen hosts outside the map are not used with it)
        cams = {k: _cam_lobes(max(k, 0), n_kf, closures) for k in range(-n_out, n_kf)}
    Rcw = np.stack([cams[k][0] for k in range(n_kf)])
    tcw = np.stack([cams[k][1] for k in range(n_kf)])

    def project(k, Xw):
        R, t = cams[k]
        Xc = Xw @ R.T + t
        z = Xc[..., 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            u = fx * Xc[..., 0] / z + cx
            v = fy * Xc[..., 1] / z + cy
        return u, v, z

    # ------------------------------------------------------------------ scene points
    frozen = rng.random(n_pt) < frozen_frac
    if n_kf == 1:
        frozen[:] = True
    host = np.where(frozen, -1 - rng.integers(0, n_out, n_pt), rng.integers(0, max(1, n_kf - 1), n_pt)).astype(np.int64)
    u0 = rng.uniform(16, W - 16, n_pt)
    v0 = rng.uniform(16, H - 16, n_pt)
    rho = rng.uniform(0.125, 0.5, n_pt)
    far = rng.random(n_pt) < far_frac
    rho[far] = rng.uniform(0.01, 0.02, far.sum())
    ray = np.stack([(u0 - cx) / fx, (v0 - cy) / fy], 1)
    Xh = np.concatenate([ray, np.ones((n_pt, 1))], 1) / rho[:, None]
    Xw = np.empty((n_pt, 3))
    for k in np.unique(host):
        m = host == k
        R, t = cams[int(k)]
        Xw[m] = (Xh[m] - t) @ R          # X_w = R^T (X_c - t)
    per_kf = [[] for _ in range(n_kf)]    # (pt, u, v)
    gross = np.zeros(n_pt, bool)          # points with a gross-outlier observation (what a local BA's outlier pass flags)
    order = np.arange(n_pt)
    for j in order:
        h = int(host[j])
        if far[j]:
            cand = rng.choice(n_kf, size=min(n_kf, 3 * max_targets), replace=False)
            cand.sort()
        elif h >= 0 and closures > 0:
            nxt = h + 1 + np.arange(band)
            lo = [le for le in lobes if le[0] <= h < le[1] and h + band >= le[1]]
            if lo and j % 2:                                  # the end of a loop: every second landmark is seen again by the loop's first keyframes (the closure),
                nxt = np.where(nxt < lo[0][1], nxt, lo[0][0] + (nxt - lo[0][1]))     # the others by the keyframes that follow (the trajectory goes on)
            cand = np.sort(nxt[nxt < n_kf])
        elif h >= 0 and loop:
            nxt = h + 1 + np.arange(min(band, n_kf - loop_at - 1))                  # the ring closes: the last keyframes observe the first ones' landmarks
            cand = np.sort(np.where(nxt < n_kf, nxt, loop_at + (nxt - n_kf)))      # (a loop that starts at keyframe loop_at: its end runs into keyframe loop_at)
        elif h >= 0:
            cand = np.arange(h + 1, min(n_kf, h + 1 + band))
        else:
            cand = np.arange(0, min(n_kf, band))
        got = 0
        if h >= 0 and self_obs and (j % 2 == 0):
            per_kf[h].append((j, u0[j], v0[j]))          # host observes its own point (skipped: host == target)
        for k in cand:
            if k == h:
                continue
            u, v, z = project(int(k), Xw[j])
            if z > 0.1 and 16 <= u < W - 16 and 16 <= v < H - 16:
                nu, nv = rng.normal(0, noise_px, 2)
                if rng.random() < outlier_frac:
                    nu += rng.choice([-1, 1]) * rng.uniform(8, 20)
                    nv += rng.choice([-1, 1]) * rng.uniform(8, 20)
                    gross[j] = True
                per_kf[int(k)].append((j, u + nu, v + nv))
                got += 1
                if got >= max_targets:
                    break
    kf_off = np.zeros(n_kf + 1, np.int64)
    for k in range(n_kf):
        kf_off[k + 1] = kf_off[k] + len(per_kf[k])
    P.sgood = np.ones(int(kf_off[-1]), np.uint8)
    if P.sgood.size > 10:
        P.sgood[rng.integers(0, P.sgood.size, max(1, P.sgood.size // 100))] = 0
    bad_pt = gross.copy()
    if drop_outlier_points:
        for k in range(n_kf):
            for i, (j, u, v) in enumerate(per_kf[k]):
                if not P.sgood[kf_off[k] + i]:
                    bad_pt[j] = True
    for l in range(P.n_levels):
        kk, pp, ff, uv = [], [], [], []
        for k in range(n_kf):
            for i, (j, u, v) in enumerate(per_kf[k]):
                if l > 0 and (i % (2 ** l)) != 0:      # coarser levels see a subset (tool::GetPyramidPts)
                    continue
                if drop_outlier_points and bad_pt[j]:  # FLAG_BAD points are not in GetAllMapPoints(false)
                    continue
                kk.append(k); pp.append(j); ff.append(kf_off[k] + i); uv.append((u, v))
        P.sobs_kf[l], P.sobs_pt[l], P.sobs_flag[l] = np.array(kk, np.int32), np.array(pp, np.int32), np.array(ff, np.int32)
        P.sobs_uv0[l] = np.array(uv, np.float64).reshape(-1, 2)
    P.pt_ray = ray
    P.pt_host = np.where(host >= 0, host, -1).astype(np.int32)
    Trw = np.zeros((n_pt, 12))
    for j in np.nonzero(host < 0)[0]:
        R, t = cams[int(host[j])]
        Trw[j] = np.concatenate([R, t[:, None]], 1).reshape(-1)
    P.pt_host_Trw = Trw

    # ------------------------------------------------------------------ text planes
    planes = []
    tfrozen = rng.random(n_text) < frozen_frac
    if n_kf == 1:
        tfrozen[:] = True
    a_half, b_half = 0.4, 0.15                        # 0.8 m x 0.3 m text patches
    gnx = max(1, int(np.ceil(np.sqrt(max(n_text, 1) * 4.0 / 3.0))))
    gny = max(1, int(np.ceil(max(n_text, 1) / gnx)))
    cells = rng.permutation(gnx * gny)
    for j in range(n_text):
        for _try in range(100):
            pl = _Plane()
            pl.host = int(-1 - rng.integers(0, n_out)) if tfrozen[j] else int(rng.integers(0, max(1, n_kf - text_targets)))
            cxi, cyi = cells[j] % gnx, cells[j] // gnx     # jittered grid in the host image limits patch overlap
            uc = 80 + (W - 160) * (cxi + 0.5 + rng.uniform(-0.2, 0.2)) / gnx
            vc = 60 + (H - 120) * (cyi + 0.5 + rng.uniform(-0.2, 0.2)) / gny
            z0 = rng.uniform(3.0, 6.0)
            m0 = np.array([(uc - cx) / fx, (vc - cy) / fy, 1.0])
            tilt = np.deg2rad(rng.uniform(0, 30))
            ang = rng.uniform(0, 2 * np.pi)
            n = _rodrigues(tilt * np.array([np.cos(ang), np.sin(ang), 0.0])) @ np.array([0, 0, -1.0])
            X0 = m0 * z0
            d = -float(n @ X0)
            pl.n, pl.d, pl.X0 = n, d, X0
            pl.theta = n / d
            e1 = np.array([1.0, 0, 0]) - n[0] * n
            e1 /= np.linalg.norm(e1)
            e2 = np.cross(n, e1)
            pl.e1, pl.e2 = e1, e2
            cs = np.array([[-a_half, -b_half], [a_half, -b_half], [a_half, b_half], [-a_half, b_half]])
            Xc = X0 + cs[:, :1] * e1 + cs[:, 1:] * e2
            pl.box_ray = Xc[:, :2] / Xc[:, 2:]
            bu, bv = fx * pl.box_ray[:, 0] + cx, fy * pl.box_ray[:, 1] + cy
            if bu.min() < 12 or bu.max() > W - 12 or bv.min() < 12 or bv.max() > H - 12:
                continue
            nf = 6
            pl.amp = rng.uniform(10, 22, nf)
            fr = rng.uniform(2.0, 7.0, nf)
            fa = rng.uniform(0, 2 * np.pi, nf)
            pl.f1, pl.f2 = fr * np.cos(fa), fr * np.sin(fa)
            pl.ph = rng.uniform(0, 2 * np.pi, nf)
            pl.base = rng.uniform(100, 150)
            # observers
            Rh, th = cams[pl.host]
            Xw_c = (Xc - th) @ Rh
            obs = []
            cand = range(n_kf) if pl.host < 0 else range(pl.host + 1, n_kf)
            for k in cand:
                u, v, z = project(k, Xw_c)
                if np.all(z > 0.1) and u.min() >= 8 and u.max() < W - 8 and v.min() >= 8 and v.max() < H - 8:
                    obs.append(k)
                    if len(obs) >= text_targets:
                        break
            if len(obs) == 0:
                continue
            pl.obs = obs
            planes.append(pl)
            break
        else:
            raise RuntimeError("could not place text plane")

    def texture(pl, s, t):
        val = np.full(s.shape, pl.base)
        for i in range(pl.amp.size):
            val = val + pl.amp[i] * np.sin(2 * np.pi * (pl.f1[i] * s + pl.f2[i] * t) + pl.ph[i])
        return val

    if n_text > 0:
        # ---- images: render every plane into every camera (window + outside hosts)
        imgs = {}
        for k in range(-n_out, n_kf):
            img = 110.0 + rng.normal(0, 4.0, (H, W))
            Rk, tk = cams[k]
            for pl in planes:
                if k != pl.host and k not in pl.obs:
                    continue                         # a patch is painted only where the problem uses it (limits overlap)
                Rh, th = cams[pl.host]
                Rhk = Rh @ Rk.T                      # X_h = Rhk X_k + thk
                thk = th - Rhk @ tk
                ext = np.array([[-1.25 * a_half, -1.4 * b_half], [1.25 * a_half, -1.4 * b_half],
                                [1.25 * a_half, 1.4 * b_half], [-1.25 * a_half, 1.4 * b_half]])
                Xe = pl.X0 + ext[:, :1] * pl.e1 + ext[:, 1:] * pl.e2
                Xk = (Xe - thk) @ Rhk                # X_k = Rhk^T (X_h - thk)
                if np.any(Xk[:, 2] < 0.1):
                    continue
                uu, vv = fx * Xk[:, 0] / Xk[:, 2] + cx, fy * Xk[:, 1] / Xk[:, 2] + cy
                x0, x1 = int(max(0, np.floor(uu.min()))), int(min(W - 1, np.ceil(uu.max())))
                y0, y1 = int(max(0, np.floor(vv.min()))), int(min(H - 1, np.ceil(vv.max())))
                if x1 < x0 or y1 < y0:
                    continue
                xs, ys = np.meshgrid(np.arange(x0, x1 + 1, dtype=np.float64), np.arange(y0, y1 + 1, dtype=np.float64))
                m = np.stack([(xs - cx) / fx, (ys - cy) / fy, np.ones_like(xs)], -1)
                Rm = m @ Rhk.T
                den = Rm @ pl.n
                with np.errstate(divide="ignore", invalid="ignore"):
                    z = -(pl.d + pl.n @ thk) / den
                Xh_ = Rm * z[..., None] + thk
                s = (Xh_ - pl.X0) @ pl.e1
                t = (Xh_ - pl.X0) @ pl.e2
                ins = (z > 0.1) & (np.abs(s) <= 1.25 * a_half) & (np.abs(t) <= 1.4 * b_half)
                val = texture(pl, s, t) + rng.normal(0, 0.25, s.shape)
                sub = img[y0:y1 + 1, x0:x1 + 1]
                sub[ins] = val[ins]
            imgs[k] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        pyr = {k: [imgs[k]] for k in imgs}
        for k in imgs:
            for l in range(1, P.n_levels):
                pyr[k].append(_pyr_down(pyr[k][l - 1]))
        for l in range(P.n_levels):
            P.img[l] = np.stack([pyr[k][l] for k in range(n_kf)])

        # ---- reference features (mapText::GetObjectInfo, tool::GetNeighbour)
        P.theta = np.stack([pl.theta for pl in planes])
        P.text_host = np.array([pl.host if pl.host >= 0 else -1 for pl in planes], np.int32)
        P.text_box_ray = np.stack([pl.box_ray for pl in planes])
        Twr = np.zeros((n_text, 12))
        for j, pl in enumerate(planes):
            if pl.host < 0:
                R, t = cams[pl.host]
                Twr[j] = np.concatenate([R.T, (-R.T @ t)[:, None]], 1).reshape(-1)
        P.text_host_Twr = Twr
        F0 = feats[0]
        gx = int(np.ceil(np.sqrt(F0 * 4)))
        gy = int(np.ceil(F0 / gx))
        feat_uv0 = []
        for pl in planes:
            ii = np.arange(F0)
            s = (-0.9 + 1.8 * ((ii % gx) + 0.5 + rng.uniform(-0.3, 0.3, F0)) / gx) * a_half
            t = (-0.8 + 1.6 * ((ii // gx) + 0.5 + rng.uniform(-0.3, 0.3, F0)) / gy) * b_half
            X = pl.X0 + s[:, None] * pl.e1 + t[:, None] * pl.e2
            feat_uv0.append(np.stack([fx * X[:, 0] / X[:, 2] + cx, fy * X[:, 1] / X[:, 2] + cy], 1))
        for l in range(P.n_levels):
            sc = 0.5 ** l
            Kl = K_GENERAL_MOTION * sc
            off, raw, uv, ref = [0], [], [], []
            nl = min(feats[l] if l < len(feats) else feats[-1], F0)
            for j, pl in enumerate(planes):
                sel = np.arange(F0) if l == 0 else np.sort(rng.choice(F0, nl, replace=False))
                him = pyr[pl.host][l].astype(np.float64)
                corners = np.stack([Kl[0] * pl.box_ray[:, 0] + Kl[2], Kl[1] * pl.box_ray[:, 1] + Kl[3]], 1)
                mu, sg = _quad_stats(pyr[pl.host][l], corners)
                if sg == 0:
                    sg = 1.0
                c = feat_uv0[j][sel] * sc
                tu = c[:, :1] + TAP_DX[None, :]
                tv = c[:, 1:] + TAP_DY[None, :]
                I = _bilinear(him, tu, tv)
                raw.append(sel); uv.append(c); ref.append((I - mu) / sg)
                off.append(off[-1] + sel.size)
            P.tfeat_off[l] = np.array(off, np.int32)
            P.tfeat_raw[l] = np.concatenate(raw).astype(np.int32)
            P.tfeat_uv[l] = np.concatenate(uv)
            P.tfeat_ref[l] = np.concatenate(ref)
        # ---- text observations, KF-major
        tk, tt = [], []
        for k in range(n_kf):
            for j, pl in enumerate(planes):
                if k in pl.obs or (self_obs and k == pl.host and j % 2 == 0):
                    tk.append(k); tt.append(j)
        P.tobs_kf, P.tobs_text = np.array(tk, np.int32), np.array(tt, np.int32)
        P.tobs_good = np.ones(len(tk), np.uint8)
        P.tobs_fgood_off = (np.arange(len(tk) + 1) * F0).astype(np.int32)
        P.tfgood = np.ones(len(tk) * F0, np.uint8)
        if P.tfgood.size > 100:
            P.tfgood[rng.integers(0, P.tfgood.size, P.tfgood.size // 50)] = 0
        if len(tk) > 20:
            P.tobs_good[rng.integers(0, len(tk))] = 0
    else:
        P.theta = np.zeros((0, 3))
        P.text_host = np.zeros(0, np.int32)
        P.text_box_ray = np.zeros((0, 4, 2))
        P.text_host_Twr = np.zeros((0, 12))

    # ------------------------------------------------------------------ parameters: truth (+) perturbation
    pose_true = np.zeros((n_kf, 7))
    for k in range(n_kf):
        pose_true[k, :4] = _R_to_q(Rcw[k])
        pose_true[k, 4:] = tcw[k]
    P.truth = {"pose": pose_true.copy(), "rho": rho.copy(), "theta": P.theta.copy()}
    pose = pose_true.copy()
    rho_i = rho.copy()
    theta_i = P.theta.copy()
    if perturb:
        first_free = 0 if n_kf == 1 else min(n_fixed, n_kf)
        for k in range(first_free, n_kf):
            ax = rng.normal(size=3)
            ax /= np.linalg.norm(ax)
            dR = _rodrigues(np.deg2rad(rot_deg) * ax)
            pose[k, :4] = _R_to_q(dR @ Rcw[k])
            dv = rng.normal(size=3)
            pose[k, 4:] = (dR @ tcw[k] if perturb_in_camera else tcw[k]) + trans_m * dv / np.linalg.norm(dv)
        movable = P.pt_host >= 0
        rho_i[movable] *= 1 + rng.uniform(-lm_rel, lm_rel, movable.sum())
        if n_text > 0:
            mv = P.text_host >= 0
            theta_i[mv] *= 1 + rng.uniform(-lm_rel, lm_rel, (mv.sum(), 1))
    P.pose, P.rho, P.theta = pose, rho_i, theta_i
    ki = np.zeros(n_kf, np.uint8)
    ki[:min(2, n_kf)] = 1 if n_kf > 1 else 0
    P.kf_initial = ki if kf_initial is None else np.asarray(kf_initial, np.uint8)
    return P.normalise()


# ---- the named configurations of SURVEY.md 8(d) --------------------------------------------------
def config_c1(seed=SEED):
    """C1 plumbing: 10 KF / 2000 pts / 20 planes."""
    return make_problem(10, 2000, 20, seed, text_targets=4)


def config_c3(seed=SEED):
    """C3 pose-only: 1 frame, 3000 scene blocks (frozen landmarks) + 200 text blocks (25 planes x 8 features)."""
    return make_problem(1, 3000, 25, seed, feats=(8, 6, 4), frozen_frac=1.0, n_out=6, max_targets=1, text_targets=1)


def config_c4(seed=SEED):
    """C4 = the headline local-BA window: 20 KF x 5000 pts x 100 text planes (64 features each at level 0)."""
    return make_problem(20, 5000, 100, seed)


def config_global(n_kf=500, n_pt=50000, seed=SEED, max_targets=8, band=12, far_frac=0.0, loop=False, loop_at=0, closures=0, drop_outlier_points=False, perturb_in_camera=False):
    """C5 / C6: scene-only global BA (the reference's GlobalBA ignores text, optimizer.cc:1707).  loop: closed trajectory (the map right
    after a loop closure: the co-visibility graph is a ring, not a band)."""
    return make_problem(n_kf, n_pt, 0, seed, max_targets=max_targets, frozen_frac=0.0, band=band, far_frac=far_frac,
                        n_levels=1, rot_deg=0.2, trans_m=0.01, loop=loop, loop_at=loop_at, closures=closures,
                        drop_outlier_points=drop_outlier_points, perturb_in_camera=perturb_in_camera)


def tiny(seed=7, n_kf=5, n_pt=60, n_text=4, **kw):
    """Small problem for unit tests (seconds on the CPU oracle)."""
    kw.setdefault("feats", (12, 8, 6))
    kw.setdefault("text_targets", 3)
    return make_problem(n_kf, n_pt, n_text, seed, **kw)


def init_pair(seed=SEED, n_pt=300, n_text=3):
    """optimizer::InitBA: two keyframes, every landmark hosted in the first (constant) one, 4 pyramid levels."""
    return make_problem(2, n_pt, n_text, seed, feats=(32, 16, 10, 6), n_levels=4, frozen_frac=0.0, max_targets=1, text_targets=1,
                        n_fixed=1, kf_initial=[1, 0], self_obs=False)


def landmark_refine(seed=SEED, n_kf=5, n_pt=150, n_text=4):
    """optimizer::OptimizeLandmarker / ThetaOptimMultiFs: every pose constant (and at its true value), landmarks perturbed."""
    return make_problem(n_kf, n_pt, n_text, seed, feats=(24, 12, 8, 6), n_levels=4, frozen_frac=0.0, text_targets=3,
                        n_fixed=n_kf, kf_initial=np.ones(n_kf, np.uint8))


def sim3_matches(seed=SEED, n=300, outlier_frac=0.1, noise_px=0.5, scale=1.07):
    """optimizer::OptimizeSim3 input: n 3D-2D matches between two keyframes related by a Sim3 (loop closure with scale drift).
    Returns dict(P1, uv1, P2, uv2, inliers, sim0 (perturbed initial Sim12), sim_true, K)."""
    rng = np.random.default_rng(seed)
    K = np.array([384.396254546, 382.825746531, 315.635886103, 249.182929809])
    ang = np.deg2rad(7.0); ax = np.array([0.2, 1.0, 0.1]); ax /= np.linalg.norm(ax)
    q = np.concatenate([[np.cos(ang/2)], np.sin(ang/2)*ax]); t = np.array([0.25, -0.05, 0.1])

    def R_of(qq):
        w, x, y, z = qq/np.linalg.norm(qq)
        return np.array([[1 - 2*(y*y + z*z), 2*(x*y - w*z), 2*(x*z + w*y)], [2*(x*y + w*z), 1 - 2*(x*x + z*z), 2*(y*z - w*x)], [2*(x*z - w*y), 2*(y*z + w*x), 1 - 2*(x*x + y*y)]])
    R = R_of(q)
    P2 = np.stack([rng.uniform(-1.5, 1.5, n), rng.uniform(-1.0, 1.0, n), rng.uniform(2.5, 7.0, n)], 1)
    P1 = (scale*(R @ P2.T)).T + t

    def proj(P):
        return np.stack([K[0]*P[:, 0]/P[:, 2] + K[2], K[1]*P[:, 1]/P[:, 2] + K[3]], 1)
    uv1 = proj(P1) + rng.normal(0, noise_px, (n, 2)); uv2 = proj(P2) + rng.normal(0, noise_px, (n, 2))
    bad = rng.random(n) < outlier_frac
    uv1[bad] += rng.uniform(8, 30, (int(bad.sum()), 2))*rng.choice([-1, 1], (int(bad.sum()), 2))
    # the map points carry some depth error on each side (posObv comes from each keyframe's own map)
    P1n = P1*(1 + rng.normal(0, 0.004, (n, 1))); P2n = P2*(1 + rng.normal(0, 0.004, (n, 1)))
    dq = np.concatenate([[1.0], rng.normal(0, 0.01, 3)]); dq /= np.linalg.norm(dq)
    q0 = np.array([dq[0]*q[0] - dq[1:] @ q[1:], *(dq[0]*q[1:] + q[0]*dq[1:] + np.cross(dq[1:], q[1:]))])
    sim0 = np.concatenate([q0, t + rng.normal(0, 0.03, 3), [scale*1.04]])
    return {"P1": P1n, "uv1": uv1.astype(np.float32), "P2": P2n, "uv2": uv2.astype(np.float32), "inliers": np.ones(n, np.uint8),
            "sim0": sim0, "sim_true": np.concatenate([q, t, [scale]]), "K": K}


def _q_mul(a, b):
    return np.array([a[0]*b[0] - a[1:] @ b[1:], *(a[0]*b[1:] + b[0]*a[1:] + np.cross(a[1:], b[1:]))])


def _q_rot(q, v):
    uv = 2*np.cross(q[1:], v)
    return v + q[0]*uv + np.cross(q[1:], uv)


def _sim_mul(A, B):
    """(q, t, s) composition A o B: x -> sA RA (sB RB x + tB) + tA."""
    return np.concatenate([_q_mul(A[:4], B[:4]), A[7]*_q_rot(A[:4], B[4:7]) + A[4:7], [A[7]*B[7]]])


def _sim_inv(A):
    qi = np.array([A[0], -A[1], -A[2], -A[3]])
    return np.concatenate([qi, _q_rot(qi, -A[4:7]/A[7]), [1.0/A[7]]])


def pose_graph(seed=SEED, n_kf=40, window=3, loop_at=2, scale_drift=1.06):
    """optimizer::OptimizeLoop input: a trajectory with accumulated drift, covisibility connections that hold at the drifted
    estimate, and loop connections between the last keyframes (corrected by the loop Sim3) and the neighbourhood of keyframe
    `loop_at`.  Returns dict(pose [n, 8], fixed, edge_i, edge_j, meas [m, 8])."""
    rng = np.random.default_rng(seed)
    def pose_of(k, drift):
        a = 2*np.pi*k/n_kf*(1.0 + drift*0.02)
        q = np.array([np.cos(a/2), 0.0, np.sin(a/2), 0.0]); q = _q_mul(q, np.concatenate([[1.0], drift*0.002*k*np.array([0.3, 0.1, -0.2])])); q /= np.linalg.norm(q)
        c = np.array([3*np.cos(a), 0.05*np.sin(3*a), 3*np.sin(a)])*(1.0 + drift*0.004*k)           # camera centre, drifting outwards
        return np.concatenate([q, -_q_rot(q, c), [1.0]])                                             # T_cw
    true = [pose_of(k, 0.0) for k in range(n_kf)]
    est = [pose_of(k, 1.0) for k in range(n_kf)]
    ei, ej, meas = [], [], []
    for i in range(n_kf):                                                                             # NormConnections: both directions occur in the reference
        for j in range(i + 1, min(n_kf, i + 1 + window)):
            for (a, b) in ((i, j), (j, i)):
                ei.append(a); ej.append(b); meas.append(_sim_mul(est[b], _sim_inv(est[a])))          # Sji = Sjw * Siw^-1
    # loop: the last keyframe group gets the corrected Sim3 (vConnectKFs / mScw), connected to the loop group
    ini = [e.copy() for e in est]
    cur_group = list(range(n_kf - 3, n_kf)); loop_group = [loop_at, loop_at + 1, loop_at + 2]
    corr = {}
    for c in cur_group:
        rel = _sim_mul(true[c], _sim_inv(true[loop_at]))                                            # what the loop detection measures
        S = _sim_mul(rel, est[loop_at]); S = np.concatenate([S[:4], S[4:7]*scale_drift, [scale_drift]])   # Sim3 with the scale the drifted map has
        corr[c] = S; ini[c] = S.copy()
    for c in cur_group:
        for l in loop_group:
            ei.append(l); ej.append(c); meas.append(_sim_mul(corr[c], _sim_inv(ini[l])))
    fixed = np.zeros(n_kf, np.uint8); fixed[[0, 1, loop_at]] = 1                                    # optimizer.cc:861-869
    pose = np.array(ini) + 0.0
    n_loop = 3*3
    return {"pose": pose, "fixed": fixed, "edge_i": np.array(ei, np.int32), "edge_j": np.array(ej, np.int32), "meas": np.array(meas),
            # the object-graph form the reference's caller holds (loopClosing::CorrectLoop): the drifted keyframe poses (mTcw), the corrected
            # Sim3 of the current keyframe's neighbourhood (vConnectKFs; mScw = the current keyframe's), the normal and the loop connections,
            # KF = the last keyframe, LoopKF = keyframe loop_at
            "est": np.array(est), "conn_idx": np.array(cur_group, np.int32), "conn_sim": np.array([corr[c] for c in cur_group]),
            "n_loop_edges": n_loop, "kf_cur": n_kf - 1, "kf_loop": loop_at}


def _q_to_R(q):
    w, x, y, z = q/np.linalg.norm(q)
    return np.array([[1 - 2*(y*y + z*z), 2*(x*y - w*z), 2*(x*z + w*y)],
                     [2*(x*y + w*z), 1 - 2*(x*x + z*z), 2*(y*z - w*x)],
                     [2*(x*z - w*y), 2*(y*z + w*x), 1 - 2*(x*x + y*y)]])


def window_of(Q, k0, n, kf_ids=None):
    """The sliding window [k0, k0 + n) of a longer sequence Q (tracking.cc:828-842: LocalBundleAdjustment on the last 20 keyframes, once per new
    keyframe): the window's keyframes with their observations; a landmark hosted in a keyframe outside the window is frozen in that host (its
    pose taken from Q, as optimizer.cc:219-262 does); landmark indices stay those of Q.  kf_ids: identities of Q's keyframes (the plane cache
    of the context, tsba_problem.kf_id)."""
    Q.normalise()
    W_ = Q.copy()
    pose = Q.pose.reshape(-1, 7)
    W_.pose = pose[k0:k0 + n].copy()
    W_.kf_initial = Q.kf_initial[k0:k0 + n].copy()
    inside = lambda h: (h >= k0) & (h < k0 + n)
    # scene points
    ph = Q.pt_host.astype(np.int64)
    Trw = Q.pt_host_Trw.reshape(-1, 12).copy()
    for j in np.nonzero((ph >= 0) & ~inside(ph))[0]:
        R = _q_to_R(pose[ph[j], :4]); Trw[j] = np.concatenate([R, pose[ph[j], 4:7, None]], 1).reshape(-1)
    W_.pt_host = np.where(inside(ph), ph - k0, -1).astype(np.int32)
    W_.pt_host_Trw = Trw
    kf0 = Q.sobs_kf[0]
    f_lo, f_hi = int(np.searchsorted(kf0, k0, "left")), int(np.searchsorted(kf0, k0 + n, "left"))
    assert np.array_equal(Q.sobs_flag[0], np.arange(Q.sgood.size)) and np.all(np.diff(kf0) >= 0)
    W_.sgood = Q.sgood[f_lo:f_hi].copy()
    for l in range(Q.n_levels):
        m = inside(Q.sobs_kf[l].astype(np.int64))
        W_.sobs_kf[l] = (Q.sobs_kf[l][m] - k0).astype(np.int32); W_.sobs_pt[l] = Q.sobs_pt[l][m].copy()
        W_.sobs_flag[l] = (Q.sobs_flag[l][m] - f_lo).astype(np.int32); W_.sobs_uv0[l] = Q.sobs_uv0[l].reshape(-1, 2)[m].copy()
        if Q.img[l] is not None:
            W_.img[l] = Q.img[l][k0:k0 + n].copy()
    # text planes
    th = Q.text_host.astype(np.int64)
    Twr = Q.text_host_Twr.reshape(-1, 12).copy()
    for j in np.nonzero((th >= 0) & ~inside(th))[0]:
        R = _q_to_R(pose[th[j], :4]); t = pose[th[j], 4:7]; Twr[j] = np.concatenate([R.T, (-R.T @ t)[:, None]], 1).reshape(-1)
    W_.text_host = np.where(inside(th), th - k0, -1).astype(np.int32)
    W_.text_host_Twr = Twr
    m = inside(Q.tobs_kf.astype(np.int64))
    W_.tobs_kf = (Q.tobs_kf[m] - k0).astype(np.int32); W_.tobs_text = Q.tobs_text[m].copy(); W_.tobs_good = Q.tobs_good[m].copy()
    off = Q.tobs_fgood_off
    keep = np.nonzero(m)[0]
    W_.tobs_fgood_off = np.concatenate([[0], np.cumsum(off[keep + 1] - off[keep])]).astype(np.int32)
    W_.tfgood = np.concatenate([Q.tfgood[off[t]:off[t + 1]] for t in keep]) if keep.size else np.zeros(0, np.uint8)
    W_.kf_id = None if kf_ids is None else np.asarray(kf_ids, np.int64)[k0:k0 + n].copy()
    W_.truth = {}
    return W_.normalise()
