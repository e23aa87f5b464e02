"""GPU parity of the BA-pyramid / reference-feature front-end (libtsframe.so through the C ABI) against the CPU oracle: bit-exact
planes (integer arithmetic), bit-exact fp64 samples (no FMA contraction on either side), identical selections and orders."""
import numpy as np
import pytest
from scipy import ndimage

pytestmark = pytest.mark.gpu


def _img(seed, h=480, w=640):
    rng = np.random.default_rng(seed)
    base = ndimage.gaussian_filter(rng.normal(0, 1, (h, w)), 3.0)
    return np.clip(128 + 600*base + rng.normal(0, 5, (h, w)), 0, 255).astype(np.uint8)


@pytest.fixture(scope="module")
def fr():
    from textslam_amd.frame import Frame
    return Frame(0)


@pytest.mark.parametrize("shape", [(480, 640), (97, 131), (33, 18)])
def test_pyramid_planes_bit_exact(fr, oracle_lib, shape):
    img = _img(shape[0], *shape)
    fr.GetPyrMat(img, 4)
    ref = oracle_lib.frame_pyramid(img, 4)
    for l in range(4):
        assert fr.level_shape(l) == ref[l][0].shape
        for which in range(4):
            assert np.array_equal(fr.level(l, which), ref[l][which]), (l, which)
        assert fr.level_device_ptr(l) != 0


@pytest.mark.parametrize("mode", [0, 1])
def test_pyramid_pts_bit_exact(fr, oracle_lib, mode):
    img = _img(3)
    fr.GetPyrMat(img, 4)
    pyr = oracle_lib.frame_pyramid(img, 4)
    rng = np.random.default_rng(4)
    box = (200.0, 150.0, 460.0, 260.0)
    inv = [1.0, 0.5, 0.25, 0.125]
    for n in (0, 1, 300, 2500):
        if mode == 0:
            xy = np.stack([rng.uniform(box[0] + 0.5, box[2] - 0.5, n), rng.uniform(box[1] + 0.5, box[3] - 0.5, n)], 1).astype(np.float32)
            got = fr.GetPyramidPts(xy, box[:2], box[2:], inv)
        else:
            xy = np.stack([rng.uniform(0, 639, n), rng.uniform(0, 479, n)], 1).astype(np.float32)
            got = fr.GetPyramidPtsScene(xy, inv)
        xy[::13] = np.rint(xy[::13])
        if mode == 0:
            got = fr.GetPyramidPts(xy, box[:2], box[2:], inv)
        else:
            got = fr.GetPyramidPtsScene(xy, inv)
        ref = oracle_lib.frame_pyramid_pts(mode, xy, box if mode == 0 else None, pyr, inv)
        for k in ("level_off", "idx", "u", "v", "inten", "in"):
            assert np.array_equal(got[k], ref[k]), (n, k)
        if n >= 300:
            assert ref["level_off"][2] - ref["level_off"][1] > 20        # the coarse levels keep a real subset


def test_neighbours_bit_exact_and_edges(fr, oracle_lib):
    from textslam_amd.frame import FrameError
    img = _img(6)
    fr.GetPyrMat(img, 3)
    pyr = oracle_lib.frame_pyramid(img, 3)
    rng = np.random.default_rng(7)
    for l in range(3):
        h, w = pyr[l][0].shape
        uv = np.stack([rng.uniform(-2, w + 2, 700), rng.uniform(-2, h + 2, 700)], 1)      # includes taps outside the image (IN = false, 0)
        uv[:40] = np.rint(uv[:40])
        I, N, inn = fr.CalNormvec(l, uv, 101.5, 37.25)
        Io, No, ino = oracle_lib.frame_neighbours(pyr[l][0], uv, 101.5, 37.25)
        assert np.array_equal(I, Io) and np.array_equal(N, No) and np.array_equal(inn, ino)
        assert 0 < inn.sum() < len(inn)
    with pytest.raises(FrameError):
        fr.CalNormvec(0, np.zeros((1, 2)), 100.0, 0.0)                # tool::CalNormvec returns false on std == 0
    I, N, inn = fr.CalNormvec(0, np.zeros((0, 2)), 1.0, 1.0)
    assert I.shape == (0, 8)


QUADS = {
    "plain": [(200.3, 150.7), (460.2, 141.1), (455.9, 260.0), (204.0, 266.5)],
    "integer_rect": [(100, 50), (180, 50), (180, 90), (100, 90)],
    "clipped": [(-40.5, -30.2), (700.7, 25.1), (650.9, 560.0), (18.0, 366.5)],
    "bow_tie": [(100.0, 100.0), (300.0, 220.0), (300.0, 100.0), (100.0, 220.0)],
    "point": [(77.2, 33.9), (77.4, 33.1), (77.9, 33.5), (77.0, 33.0)],
    "outside": [(-50.0, -60.0), (-10.0, -60.0), (-10.0, -20.0), (-50.0, -20.0)],
    "sliver": [(10.2, 400.9), (630.8, 402.1), (630.1, 403.3), (10.9, 401.7)],
    "whole_image": [(0, 0), (639, 0), (639, 479), (0, 479)],
}


@pytest.mark.parametrize("name", sorted(QUADS))
def test_box_pixels_bit_exact(fr, oracle_lib, name):
    """tool::GetBoxAllPixs: same pixels in the same (row-major) order, same intensities, on levels 0 and 1."""
    img = _img(11)
    fr.GetPyrMat(img, 3)
    pyr = oracle_lib.frame_pyramid(img, 3)
    for level, s in ((0, 1.0), (1, 0.5)):
        quad = np.asarray(QUADS[name], np.float64)*s
        got = fr.GetBoxAllPixs(level, quad, 117.25, 31.5, K=(500.0*s, 505.0*s, 320.0*s, 240.0*s))
        u, v, I, N = oracle_lib.frame_box_pixels(pyr[level][0], quad, 117.25, 31.5)
        assert np.array_equal(got["u"], u) and np.array_equal(got["v"], v), (name, level, len(u), len(got["u"]))
        assert np.array_equal(got["featureInten"], I) and np.array_equal(got["featureNInten"], N)
        assert got["ray"].shape == (len(u), 3)
    if name == "integer_rect":
        assert len(fr.GetBoxAllPixs(0, QUADS[name], 0.0, 1.0)["u"]) == 81*41
    if name == "whole_image":
        assert len(fr.GetBoxAllPixs(0, QUADS[name], 0.0, 1.0)["u"]) == 640*480


def test_box_pixels_sigma_zero_and_capacity(fr, oracle_lib):
    """The reference divides by std without a guard: inf / nan must come out the same; a too small capacity is an error that reports the count."""
    import ctypes as C
    img = _img(12); img[60:70, 60:70] = 100
    fr.GetPyrMat(img, 1)
    quad = [(50.5, 50.5), (90.5, 52.0), (88.0, 85.0), (52.0, 80.0)]
    got = fr.GetBoxAllPixs(0, quad, 100.0, 0.0)
    u, v, I, N = oracle_lib.frame_box_pixels(img, quad, 100.0, 0.0)
    np.testing.assert_array_equal(got["featureNInten"], N)
    assert np.isnan(N).any() and np.isinf(N).any()
    n = C.c_int32(0); cap = 10
    uu = np.zeros(cap, np.int32); vv = np.zeros(cap, np.int32); a = np.zeros(cap); b = np.zeros(cap)
    ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    q = np.ascontiguousarray(quad, np.float64)
    rc = fr.lib.tsframe_box_pixels(fr.ctx, 0, q.ctypes.data_as(dp), 100.0, 1.0, cap, C.byref(n), uu.ctypes.data_as(ip), vv.ctypes.data_as(ip), a.ctypes.data_as(dp), b.ctypes.data_as(dp))
    assert rc == -1 and n.value == len(u) and b"capacity" in fr.lib.tsframe_last_error(fr.ctx)
    q[0, 0] = np.nan
    assert fr.lib.tsframe_box_pixels(fr.ctx, 0, q.ctypes.data_as(dp), 100.0, 1.0, 0, C.byref(n), None, None, None, None) == -1
    assert fr.lib.tsframe_box_pixels(fr.ctx, 5, q.ctypes.data_as(dp), 100.0, 1.0, 0, C.byref(n), None, None, None, None) == -3


def test_local_ba_on_resident_pyramid_planes():
    """tsframe -> tsba without a host round trip: LocalBundleAdjustment reading the pyramid planes frame::GetPyrMat left in HBM
    (tsba_options.img_on_device) takes bit-for-bit the same trajectory as with the same planes handed over as host images."""
    from textslam_amd import synth
    from textslam_amd.abi import options_local
    from textslam_amd.frame import Frame
    from textslam_amd.optimizer import Optimizer
    P = synth.tiny(seed=9, n_kf=6, n_pt=120, n_text=5)
    nl = P.n_levels
    frames = [Frame(0) for _ in range(P.n_kf)]
    for k, f in enumerate(frames):
        f.GetPyrMat(P.img[0][k], nl)
    for l in range(nl):
        assert frames[0].level_shape(l) == P.img[l][0].shape
        P.img[l] = np.stack([f.level(l) for f in frames])              # the planes the device holds, as host images
    A, B = P.copy(), P.copy()
    for l in range(nl):
        B.img_dev[l] = [f.level_device_ptr(l) for f in frames]
    opt = Optimizer(0)
    ra = opt.LocalBundleAdjustment(A)
    ob = options_local(); ob.img_on_device = 1
    rb = opt.LocalBundleAdjustment(B, options=ob)
    assert ra["iters"] == rb["iters"] and ra["cost1"] == rb["cost1"] and sum(ra["n_tblock"]) > 0
    assert np.array_equal(A.pose, B.pose) and np.array_equal(A.rho, B.rho) and np.array_equal(A.theta, B.theta)
    assert np.array_equal(A.tfgood, B.tfgood) and np.array_equal(A.sgood, B.sgood)
