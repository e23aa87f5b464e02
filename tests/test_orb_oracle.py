"""CPU tests of the ORB oracle (oracle/tsorb_oracle.c): each OpenCV behaviour it restates is checked against an
independent formulation (brute-force FAST definition, float bilinear / Gaussian, numpy arctan2) -- the reference ships
no tests and OpenCV is not installed, so this is what pins the checker."""
import numpy as np
import pytest

from textslam_amd.orbextractor import synthetic_frame


def test_constructor_constants(oracle_lib):
    sf, nfl, umax, gk = oracle_lib.orb_params(1000, 1.2, 8)
    assert nfl.tolist() == [217, 181, 151, 126, 105, 87, 73, 60] and nfl.sum() == 1000
    assert np.allclose(sf, 1.2 ** np.arange(8), rtol=1e-6)
    assert umax.tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert gk.tolist() == [18, 34, 49, 55, 49, 34, 18]
    _, nfl3, _, _ = oracle_lib.orb_params(3000, 1.2, 8)
    assert nfl3.sum() == 3000


def test_pyramid_sizes_and_border(oracle_lib):
    img = synthetic_frame(3)
    sizes = [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]
    for l, (w, h) in enumerate(sizes):
        lev = oracle_lib.orb_level(img, l)
        assert lev.shape == (h + 38, w + 38)
        inner = lev[19:19 + h, 19:19 + w]
        assert np.array_equal(lev[19:19 + h, 0:19], inner[:, 19:0:-1])            # BORDER_REFLECT_101
        assert np.array_equal(lev[0:19, 19:19 + w], inner[19:0:-1, :])
        assert np.array_equal(lev[19 + h:, 19:19 + w], inner[h - 2:h - 21:-1, :])
    assert np.array_equal(oracle_lib.orb_level(img, 0)[19:-19, 19:-19], img)


def test_resize_close_to_float_bilinear(oracle_lib):
    img = synthetic_frame(5)
    a = oracle_lib.orb_level(img, 0)[19:-19, 19:-19].astype(np.float64)
    b = oracle_lib.orb_level(img, 1)[19:-19, 19:-19].astype(np.float64)
    h, w = b.shape
    sx, sy = a.shape[1] / w, a.shape[0] / h
    xs = np.clip((np.arange(w) + 0.5) * sx - 0.5, 0, a.shape[1] - 1)
    ys = np.clip((np.arange(h) + 0.5) * sy - 0.5, 0, a.shape[0] - 1)
    x0 = np.floor(xs).astype(int); y0 = np.floor(ys).astype(int)
    x1 = np.minimum(x0 + 1, a.shape[1] - 1); y1 = np.minimum(y0 + 1, a.shape[0] - 1)
    fx = xs - x0; fy = ys - y0
    ref = ((1 - fy)[:, None] * ((1 - fx) * a[y0][:, x0] + fx * a[y0][:, x1]) + fy[:, None] * ((1 - fx) * a[y1][:, x0] + fx * a[y1][:, x1]))
    assert np.abs(b - ref).max() <= 1.0            # fixed-point (11-bit) vs float bilinear


def test_blur_close_to_float_gaussian(oracle_lib):
    img = synthetic_frame(6)
    b = oracle_lib.orb_level(img, 0, blurred=True).astype(np.float64)
    k = np.exp(-0.5 * (np.arange(7) - 3.0) ** 2 / 4.0); k /= k.sum()
    pad = np.pad(img.astype(np.float64), 3, mode="reflect")
    tmp = sum(k[i] * pad[:, i:i + img.shape[1]] for i in range(7))
    ref = sum(k[i] * tmp[i:i + img.shape[0], :] for i in range(7))
    assert np.abs(b - ref * (257.0 / 256.0) ** 2).max() <= 1.0      # Q8 kernel sums to 257 (OpenCV 3.3 8-bit path)


def _fast_bruteforce(img, t):
    """FAST-9/16 by definition: score = max threshold at which the pixel is still a corner; 3x3 strict NMS."""
    h, w = img.shape
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    im = img.astype(np.int32)
    c = im[3:h - 3, 3:w - 3]
    d = np.stack([c - im[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in ring])     # v - p
    d2 = np.concatenate([d, d[:8]])
    best = np.full(c.shape, -10**6)
    for s in range(16):
        arc = d2[s:s + 9]
        best = np.maximum(best, np.maximum(arc.min(0), (-arc).min(0)))
    score = np.where(best > t, best - 1, 0)
    full = np.zeros((h, w), np.int32); full[3:h - 3, 3:w - 3] = score
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = full[y, x]
            if s > 0:
                nb = full[y - 1:y + 2, x - 1:x + 2].copy(); nb[1, 1] = -1
                if s > nb.max():
                    out.append((x, y, s))
    return np.array(out, np.float32).reshape(-1, 3)


@pytest.mark.parametrize("t", [20, 7])
def test_fast_matches_definition(oracle_lib, t):
    img = np.ascontiguousarray(synthetic_frame(8)[40:240, 100:400])
    a = oracle_lib.orb_fast(img, t)
    b = _fast_bruteforce(img, t)
    assert len(a) > 5 and a.shape == b.shape and np.array_equal(a, b)


def test_fast_atan2(oracle_lib):
    rng = np.random.default_rng(2)
    y, x = rng.normal(size=2000).astype(np.float32) * 1000, rng.normal(size=2000).astype(np.float32) * 1000
    got = np.array([oracle_lib.orb_lib().tsorb_oracle_atan2(float(a), float(b)) for a, b in zip(y, x)])
    ref = np.degrees(np.arctan2(y.astype(np.float64), x.astype(np.float64))) % 360.0
    err = np.abs(got - ref); err = np.minimum(err, 360 - err)
    assert err.max() < 0.02                       # cv::fastAtan2: ~0.01 degree polynomial


def test_extract_properties(oracle_lib):
    img = synthetic_frame(9)
    kp, desc = oracle_lib.orb_extract(img)
    assert 990 <= len(kp) <= 1040 and desc.shape == (len(kp), 32)
    octave = kp[:, 5].astype(int)
    assert np.all(np.diff(octave) >= 0)                                   # level-major order
    assert np.all(kp[:, 2] == np.floor(31 * (1.2 ** octave).astype(np.float32)))      # size = 31 * scale^level (int)
    sc = (np.float32(1.2) ** octave).astype(np.float32)
    assert np.all(kp[:, 0] >= 16 * sc - 1e-3) and np.all(kp[:, 0] <= 640) and np.all(kp[:, 1] <= 480)
    assert np.all((kp[:, 3] >= 0) & (kp[:, 3] <= 360))
    # determinism and insensitivity to the stride
    kp2, desc2 = oracle_lib.orb_extract(img.copy())
    assert np.array_equal(kp, kp2) and np.array_equal(desc, desc2)
    # a featureless image yields nothing
    kp0, _ = oracle_lib.orb_extract(np.full((480, 640), 77, np.uint8))
    assert len(kp0) == 0


def test_descriptor_follows_rotation(oracle_lib):
    """Steered BRIEF: rotating the image by 180 degrees maps a keypoint's descriptor onto the descriptor of the rotated
    keypoint (same bits, since the pattern is rotated with the patch orientation)."""
    img = synthetic_frame(10)
    kp, desc = oracle_lib.orb_extract(img, nlevels=1, nfeatures=300)
    rot = np.ascontiguousarray(img[::-1, ::-1])
    kp_r, desc_r = oracle_lib.orb_extract(rot, nlevels=1, nfeatures=300)
    pos = {(float(x), float(y)): i for i, (x, y) in enumerate(kp_r[:, :2])}
    matched = close = 0
    for i, (x, y) in enumerate(kp[:, :2]):
        j = pos.get((639.0 - x, 479.0 - y))
        if j is None:
            continue
        matched += 1
        ham = int(np.unpackbits(desc[i] ^ desc_r[j]).sum())
        close += ham <= 40
    assert matched > 50 and close / matched > 0.9


def test_match_oracle_against_bruteforce(oracle_lib):
    oracle = oracle_lib
    """orb_match (GetFeaturesInArea + DescriptorDistance restatement) vs a brute-force numpy formulation: same candidate SETS,
    same distances, best = first minimum in the oracle's candidate order, second best = runner-up distance."""
    rng = np.random.default_rng(5)
    n, nq = 700, 300
    kp = np.zeros((n, 6), np.float32)
    kp[:, 0] = rng.uniform(-5, 645, n); kp[:, 1] = rng.uniform(-5, 485, n); kp[:, 5] = rng.integers(0, 8, n)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    qxy = np.stack([rng.uniform(0, 640, nq), rng.uniform(0, 480, nq)], 1).astype(np.float32)
    qr = rng.uniform(3, 40, nq).astype(np.float32)
    qlev = np.stack([rng.integers(-1, 3, nq), rng.integers(-1, 6, nq)], 1).astype(np.int32)
    qdesc = desc[rng.integers(0, n, nq)] ^ (rng.integers(0, 256, (nq, 32), dtype=np.uint8) & rng.integers(0, 256, (nq, 32), dtype=np.uint8) & rng.integers(0, 256, (nq, 32), dtype=np.uint8))
    bounds = (0.0, 640.0, 0.0, 480.0)
    out = oracle.orb_match(kp, desc, bounds, qxy, qr, qlev, qdesc, max_cand=256)
    iw, ih = 64/640.0, 48/480.0
    px = np.round((kp[:, 0].astype(np.float64) - 0.0)*iw); py = np.round((kp[:, 1].astype(np.float64))*ih)   # np.round is half-even: avoid ties
    ingrid = (px >= 0) & (px < 64) & (py >= 0) & (py < 48)
    popc = np.array([bin(v).count("1") for v in range(256)])
    for q in range(nq):
        x, y, r = qxy[q, 0], qxy[q, 1], qr[q]
        c0x = max(0, int(np.floor((float(x) - float(r))*iw))); c1x = min(63, int(np.ceil((float(x) + float(r))*iw)))
        c0y = max(0, int(np.floor((float(y) - float(r))*ih))); c1y = min(47, int(np.ceil((float(y) + float(r))*ih)))
        m = ingrid & (px >= c0x) & (px <= c1x) & (py >= c0y) & (py <= c1y)
        mn, mx = qlev[q]
        if mn > 0 or mx >= 0:
            m &= kp[:, 5] >= mn
            if mx >= 0: m &= kp[:, 5] <= mx
        m &= (np.abs(kp[:, 0] - x) < r) & (np.abs(kp[:, 1] - y) < r)
        want = np.nonzero(m)[0]
        cnt = out["cand_cnt"][q]
        assert cnt == want.size
        got = out["cand_idx"][q, :cnt]
        assert sorted(got.tolist()) == want.tolist()
        d = popc[desc[got] ^ qdesc[q]].sum(1) if cnt else np.zeros(0, int)
        assert np.array_equal(d, out["cand_dist"][q, :cnt])
        # reference order: cell column outer, cell row inner, index inside the cell
        key = [(int(px[i]), int(py[i]), int(i)) for i in got]
        assert key == sorted(key)
        if cnt:
            k = int(np.argmin(d))                             # first minimum
            assert out["best_idx"][q] == got[k] and out["best_dist"][q] == d[k]
            rest = np.delete(d, k)
            assert out["best_dist2"][q] == (rest.min() if rest.size else 2147483647)
        else:
            assert out["best_idx"][q] == -1 and out["best_dist"][q] == 2147483647
