"""CPU checks of the BA-pyramid / reference-feature restatement (oracle/tsframe_oracle.c): the image operators
against an independent scipy formulation of the same OpenCV semantics, the feature selection and the INTERVAL8 sampling against
straightforward Python loops; libtsframe.so loads and exports every symbol include/tsframe.h declares (no compute without a GPU)."""
import ctypes as C
import math
import os
import re
import numpy as np
import pytest
from scipy import ndimage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _img(seed, h=97, w=131):
    rng = np.random.default_rng(seed)
    base = ndimage.gaussian_filter(rng.normal(0, 1, (h, w)), 2.0)
    return np.clip(128 + 400*base + rng.normal(0, 6, (h, w)), 0, 255).astype(np.uint8)


def test_pyrdown_sobel_addweighted_against_scipy(oracle_lib):
    for (h, w) in ((97, 131), (120, 160), (31, 18)):
        img = _img(h, h, w)
        pyr = oracle_lib.frame_pyramid(img, 4)
        k = np.array([1, 4, 6, 4, 1], np.int64)
        cur = img.astype(np.int64)
        for l in range(4):
            if l > 0:                                            # cv::pyrDown: Gaussian 5x5 / 256, REFLECT_101, even samples, (x + 128) >> 8
                t = ndimage.correlate1d(ndimage.correlate1d(cur, k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
                cur = ((t[::2, ::2] + 128) >> 8)
                assert np.array_equal(pyr[l][0], cur.astype(np.uint8))
            sx = ndimage.correlate(cur, np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]]), mode="mirror")
            sy = ndimage.correlate(cur, np.array([[-1, -2, -1], [0, 0, 0], [1, 2, 1]]), mode="mirror")
            gx, gy = np.clip(sx, 0, 255).astype(np.uint8), np.clip(sy, 0, 255).astype(np.uint8)      # CV_8U destination: saturate
            assert np.array_equal(pyr[l][2], gx) and np.array_equal(pyr[l][3], gy)
            g = np.rint(gx.astype(np.float32)*np.float32(0.5) + gy.astype(np.float32)*np.float32(0.5)).astype(np.uint8)   # cvRound: ties to even
            assert np.array_equal(pyr[l][1], g)


def _bilinear(img, u, v):
    h, w = img.shape
    x0, y0, x1, y1 = math.floor(u), math.floor(v), math.ceil(u), math.ceil(v)
    if x0 < 0 or y0 < 0 or x1 >= w or y1 >= h:
        return 0.0, False
    a, b = u - x0, v - y0
    xr, yb = min(x0 + 1, w - 1), min(y0 + 1, h - 1)
    return ((1.0 - a)*(1.0 - b))*float(img[y0, x0]) + (a*(1.0 - b))*float(img[y0, xr]) + ((1.0 - a)*b)*float(img[yb, x0]) + (a*b)*float(img[yb, xr]), True


@pytest.mark.parametrize("mode", [0, 1])
def test_pyramid_pts_against_python_loops(oracle_lib, mode):
    img = _img(5, 240, 320)
    pyr = oracle_lib.frame_pyramid(img, 4)
    rng = np.random.default_rng(8)
    box = (60.0, 50.0, 250.0, 130.0)
    n = 180
    if mode == 0:
        xy = np.stack([rng.uniform(box[0] + 0.5, box[2] - 0.5, n), rng.uniform(box[1] + 0.5, box[3] - 0.5, n)], 1).astype(np.float32)
    else:
        xy = np.stack([rng.uniform(1, 318, n), rng.uniform(1, 238, n)], 1).astype(np.float32)
    xy[::17] = np.rint(xy[::17])                                   # integral coordinates: the zero-weight reads
    inv = [1.0, 0.5, 0.25, 0.125]
    got = oracle_lib.frame_pyramid_pts(mode, xy, box if mode == 0 else None, pyr, inv)
    off = got["level_off"]
    assert off[0] == 0 and off[1] == n and np.array_equal(got["idx"][:n], np.arange(n))
    for l in range(1, 4):
        s = inv[l]; h, w = pyr[l][0].shape
        ncell = int(n*s*s + (100 if mode == 0 else 500))
        if mode == 0:
            x0, y0, x1, y1 = [b*s for b in box]; WH = (x1 - x0)/(y1 - y0)
        else:
            x0 = y0 = 0.0; x1, y1 = float(w), float(h); WH = w/h
        ch, cw = int(math.sqrt(ncell/WH)), int(math.sqrt(ncell*WH))
        fx, fy = (x1 - x0)/cw, (y1 - y0)/ch
        sel = {}
        for j in range(n):
            pu, pv = float(xy[j, 0])*s, float(xy[j, 1])*s
            g, _ = _bilinear(pyr[l][1], pu, pv)
            m, q = int(math.floor((pu - x0)/fx + 0.5)), int(math.floor((pv - y0)/fy + 0.5))       # round(): half away from zero (values are >= 0)
            m, q = (cw - 1 if m == cw else m), (ch - 1 if q == ch else q)
            if g > 0 or mode == 1:
                sel[(m, q)] = j                                    # the reference never updates MAX: the last one wins
        exp = [sel[(i3, i4)] for i3 in range(cw) for i4 in range(ch) if (i3, i4) in sel]
        sl = slice(off[l], off[l + 1])
        assert list(got["idx"][sl]) == exp
        for k, j in zip(range(off[l], off[l + 1]), exp):
            I, ok = _bilinear(pyr[l][0], float(xy[j, 0])*s, float(xy[j, 1])*s)
            assert got["u"][k] == float(xy[j, 0])*s and got["inten"][k] == I and bool(got["in"][k]) == ok
        assert 0 < len(exp) <= n


def test_neighbours_against_python_loops(oracle_lib):
    img = _img(9, 60, 80)
    rng = np.random.default_rng(2)
    uv = np.stack([rng.uniform(-1, 81, 120), rng.uniform(-1, 61, 120)], 1)
    uv[:10] = np.rint(uv[:10])
    mu, sg = 117.25, 23.5
    I, N, inn = oracle_lib.frame_neighbours(img, uv, mu, sg)
    dx, dy = [0, 2, 1, 0, -1, -2, -1, 0], [0, 0, -1, -2, -1, 0, 1, 2]           # tool.cc:1551-1558
    for j in range(len(uv)):
        for k in range(8):
            e, ok = _bilinear(img, uv[j, 0] + dx[k], uv[j, 1] + dy[k])
            assert I[j, k] == e and N[j, k] == (e - mu)/sg
        assert bool(inn[j]) == ok                                  # feat->IN is overwritten by every tap: the last one stays


def test_libtsframe_exports_every_declared_symbol():
    import __graft_entry__ as ge
    so = os.path.join(ROOT, "textslam_amd", "libtsframe.so")
    if not os.path.exists(so):
        ge.build()
    lib = C.CDLL(so)
    names = sorted(set(re.findall(r"\b(tsframe_[a-z_0-9]+)\s*\(", open(os.path.join(ROOT, "include", "tsframe.h")).read())))
    assert len(names) >= 9
    for nme in names:
        assert hasattr(lib, nme), f"{nme} declared in include/tsframe.h but not exported"
    from textslam_amd import frame
    assert sorted(frame.EXPORTED_SYMBOLS) == names


def test_frame_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from textslam_amd.frame import Frame, FrameError
    with pytest.raises(FrameError):
        Frame(0)


def test_box_pixels_oracle_properties(oracle_lib):
    """tool::GetBoxAllPixs restatement: an integer rectangle is filled inclusively, the order is row-major, a quad outside the image
    yields what the clamped bounding box still sees of it (nothing), a convex quad agrees with a half-plane test away from its boundary."""
    rng = np.random.default_rng(5)
    img = rng.integers(0, 255, (120, 160), dtype=np.uint8)
    u, v, I, N = oracle_lib.frame_box_pixels(img, [(20, 30), (60, 30), (60, 50), (20, 50)], 100.0, 20.0)
    assert len(u) == 41*21 and u.min() == 20 and u.max() == 60 and v.min() == 30 and v.max() == 50
    key = v.astype(np.int64)*160 + u
    assert np.all(np.diff(key) > 0)                                     # row-major, no duplicates
    assert np.array_equal(I, img[v, u].astype(np.float64)) and np.array_equal(N, (I - 100.0)/20.0)
    assert len(oracle_lib.frame_box_pixels(img, [(-50, -60), (-10, -60), (-10, -20), (-50, -20)], 0.0, 1.0)[0]) == 0
    quad = np.array([(20.3, 30.7), (90.2, 25.1), (95.9, 60.0), (18.0, 66.5)])
    u, v, _, _ = oracle_lib.frame_box_pixels(img, quad, 0.0, 1.0)
    got = np.zeros(img.shape, bool); got[v, u] = True
    q = np.trunc(quad)                                                  # cv::Point truncation
    yy, xx = np.mgrid[0:120, 0:160]
    d = np.full(img.shape, np.inf)
    for i in range(4):                                                  # signed distance to each edge (clockwise in image coordinates)
        a, b = q[i], q[(i + 1) % 4]
        e = b - a
        d = np.minimum(d, ((xx - a[0])*e[1] - (yy - a[1])*e[0])/np.hypot(*e)*-1.0)
    assert got[d > 1.5].all() and not got[d < -1.5].any()
