"""GPU parity of the ORB front-end (libtsorb.so through the C ABI) against the CPU oracle: bit-exact keypoint coordinates,
octaves, responses and 256-bit descriptors; angles identical (both sides evaluate cv::fastAtan2 with the same fp32 roundings)."""
import os
import numpy as np
import pytest

from textslam_amd.orbextractor import synthetic_frame

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def orb():
    from textslam_amd.orbextractor import ORBextractor
    return ORBextractor(1000, 1.2, 8, 20, 7, device=0)


def _same(a, b):
    kp_a, d_a = a; kp_b, d_b = b
    assert kp_a.shape == kp_b.shape
    assert np.array_equal(kp_a[:, [0, 1, 2, 4, 5]], kp_b[:, [0, 1, 2, 4, 5]])       # x, y, size, response, octave: bit-exact
    assert np.array_equal(kp_a[:, 3], kp_b[:, 3])                                  # angle
    assert np.array_equal(d_a, d_b)                                                # 256-bit descriptors


def test_stages_bit_exact(orb, oracle_lib):
    img = synthetic_frame(21)
    orb.extract_batch(img)
    for l in range(8):
        assert np.array_equal(orb.debug_level(0, l), oracle_lib.orb_level(img, l))                          # S1 pyramid + border
        assert np.array_equal(orb.debug_level(0, l, True), oracle_lib.orb_level(img, l, blurred=True))      # S5 blur


@pytest.mark.parametrize("shape", [(480, 640), (240, 320), (480, 752)])
def test_pyramid_in_one_launch_is_the_chained_pyramid(oracle_lib, shape):
    """Up to four frames take k_pyramid_one (a level's tile formed inside one workgroup from a base level several levels up: two launches, or one for
    every level from the input image), larger batches a launch per level (tsorb_debug_pyramid forces any of them, and moves the split level): the same
    bytes on every level, frame included, and the same features."""
    from textslam_amd.orbextractor import ORBextractor
    h, w = shape
    imgs = np.stack([np.ascontiguousarray(np.tile(synthetic_frame(60 + s), (1, 2))[:h, :w]) for s in range(6)])
    ex = ORBextractor(1000, 1.2, 8, 20, 7, device=0)
    try:
        for mode, batch in ((1, imgs[:1]), (1, imgs), (2, imgs[:1]), (2, imgs[:3]), (3, imgs[:1]), (3, imgs), (0, imgs[:1]), (-1, imgs[:2]), (-1, imgs), (102, imgs[:1]), (105, imgs[:2]), (103, imgs[:1])):
            ex.debug_pyramid(mode)                  # (100 + s: the split level of the two launches; the shape stays as set before)
            if mode >= 100: ex.debug_pyramid(1)
            res = ex.extract_batch(batch)
            for f in sorted({0, len(batch) - 1}):
                for l in range(8):
                    assert np.array_equal(ex.debug_level(f, l), oracle_lib.orb_level(batch[f], l)), (mode, len(batch), f, l)
                _same(res[f], oracle_lib.orb_extract(batch[f]))
    finally:
        ex.close()


def test_batch_matches_oracle(orb, oracle_lib):
    imgs = np.stack([synthetic_frame(30 + s) for s in range(6)])
    res = orb.extract_batch(imgs)
    for f in range(len(imgs)):
        _same(res[f], oracle_lib.orb_extract(imgs[f]))
    # operator() form and batch independence
    _same(orb(imgs[3]), res[3])


def test_init_frame_extractor_3000(oracle_lib):
    """tracking.cc:38-39 builds a second extractor with 3 x nfeatures for the two initialisation frames."""
    from textslam_amd.orbextractor import ORBextractor
    ex = ORBextractor(3000, 1.2, 8, 20, 7, device=0)
    img = synthetic_frame(41)
    _same(ex(img), oracle_lib.orb_extract(img, nfeatures=3000, cap=8192))
    assert ex.GetFeaturesPerLevel().sum() == 3000 and ex.GetLevels() == 8


@pytest.mark.parametrize("nfeatures,scale,nlevels,shape", [(500, 1.5, 5, (480, 640)), (2000, 1.1, 8, (480, 640)), (1200, 1.2, 8, (720, 1280)), (300, 2.0, 3, (360, 480)), (1000, 1.2, 1, (480, 640))])
def test_other_extractor_parameters(oracle_lib, nfeatures, scale, nlevels, shape):
    """Other pyramids and feature counts than TextSLAM's yaml files use -- the tile sizes of the few-frames pyramid, its split level, the quadtree's closed-form
    generations (their number follows the level's feature count; a 16:9 frame starts from two nodes and takes the loop) all depend on them: a single frame
    (the few-frames launch plan) and the same frame as a launch per stage and level, both against the oracle, pyramid levels included."""
    from textslam_amd.orbextractor import ORBextractor
    h, w = shape
    img = np.ascontiguousarray(np.tile(synthetic_frame(90 + nlevels), (2, 3))[:h, :w])
    ref = oracle_lib.orb_extract(img, nfeatures=nfeatures, scale=scale, nlevels=nlevels, cap=8192)
    ex = ORBextractor(nfeatures, scale, nlevels, 20, 7, device=0)
    try:
        for mode in (-1, 0):
            ex.debug_pyramid(mode)
            _same(ex(img), ref)
            for l in range(nlevels):
                assert np.array_equal(ex.debug_level(0, l), oracle_lib.orb_level(img, l, scale=scale, nlevels=nlevels)), (mode, l)
    finally:
        ex.close()


def test_edge_cases(orb, oracle_lib):
    flat = np.full((480, 640), 128, np.uint8)                      # no corners anywhere: empty output
    kp, desc = orb(flat)
    assert len(kp) == 0 and desc.shape == (0, 32)
    low = (synthetic_frame(50).astype(np.int32) // 12 + 100).astype(np.uint8)     # low contrast: the per-cell fallback to threshold 7 fires
    _same(orb(low), oracle_lib.orb_extract(low))
    small = np.ascontiguousarray(synthetic_frame(51)[:240, :320])   # another resolution
    _same(orb(small), oracle_lib.orb_extract(small))
    # a handful of corners, some of them alone in their quarter of the image, clusters that stay together for several splits: the quadtree's closed-form
    # generations end early (a pass that changes nothing, one-key nodes at every depth) -- and one corner only, and two
    for k, boxes in enumerate(([(100, 100), (104, 300), (400, 500), (404, 508), (408, 516), (60, 600)], [(200, 320)], [(200, 320), (206, 330)])):
        sparse = np.full((480, 640), 40, np.uint8)
        for (y, x) in boxes:
            sparse[y:y + 9, x:x + 9] = 220
        _same(orb(sparse), oracle_lib.orb_extract(sparse))
    noise = np.random.default_rng(5).integers(0, 256, (480, 640)).astype(np.uint8)        # corners everywhere: quadtree under load
    before = orb.debug_fallbacks()
    _same(orb(noise), oracle_lib.orb_extract(noise, cap=8192))
    assert orb.debug_fallbacks() == before + 1          # more candidates than the LDS quadtree holds: the serial pass behind the first synchronisation
    _same(orb(small), oracle_lib.orb_extract(small))    # (and the next frame is none the worse for it)
    assert orb.debug_fallbacks() == before + 1


def test_against_golden_fixture(orb):
    g = np.load(os.path.join(GOLD, "orb_frame.npz"))
    img = synthetic_frame(int(g["seed"]))
    kp, desc = orb(img)
    assert np.array_equal(kp, g["kp"]) and np.array_equal(desc, g["desc"])


def test_full_batch_properties(orb):
    """BASELINE config 2: 64 frames 640x480 -- size-independent properties at full batch size."""
    imgs = np.stack([synthetic_frame(200 + s) for s in range(64)])
    res = orb.extract_batch(imgs)
    res2 = orb.extract_batch(imgs[::-1].copy())
    for f in (0, 17, 63):
        _same(res[f], res2[63 - f])                                # a frame's result does not depend on its batch position
        kp, desc = res[f]
        assert 990 <= len(kp) <= 1040 and np.all(np.diff(kp[:, 5]) >= 0)


def test_large_batches_take_the_split_detector_and_agree(orb, oracle_lib):
    """From 24 frames on the detector runs as two launches (levels with cells up to 34 px: two waves per cell on a 40 x 40 tile; the others through the
    general instance); below, one launch of the general instance.  Same keypoints and descriptors either way, and against the oracle -- including a frame
    of noise (candidate lists full) and a low-contrast one (the per-cell fallback threshold)."""
    imgs = [synthetic_frame(300 + s) for s in range(30)]
    imgs.append(np.random.default_rng(6).integers(0, 256, (480, 640)).astype(np.uint8))
    imgs.append((synthetic_frame(331).astype(np.int32) // 12 + 100).astype(np.uint8))
    imgs = np.stack(imgs)
    res = orb.extract_batch(imgs)
    for f in (0, 13, 29, 30, 31):
        _same(res[f], orb.extract_batch(imgs[f:f + 1])[0])
    for f in (5, 30, 31):
        _same(res[f], oracle_lib.orb_extract(imgs[f], cap=8192))
    try:                                                          # the shapes kept for A/B runs (tsorb_debug_fast_shape): the same output
        for shape in (0, 1, 3, 2):
            orb.debug_fast_shape(shape)
            alt = orb.extract_batch(imgs)
            for f in (0, 7, 30, 31):
                _same(alt[f], res[f])
            _same(orb.extract_batch(imgs[3:4])[0], res[3])        # (the split on a single frame)
        orb.debug_fast_shape(-1)
        orb.debug_pyramid(200)                                    # orientation and blur as launches of their own (one launch by default on a batch)
        apart = orb.extract_batch(imgs)
        for f in (0, 7, 30, 31):
            _same(apart[f], res[f])
    finally:
        orb.debug_fast_shape(-1); orb.debug_pyramid(201)


@pytest.mark.gpu
def test_match_search_parity_on_resident_frames(orb, oracle_lib):
    """tsorb_match_* vs the oracle, bit-exact: candidates in the reference's order, Hamming distances, best / second best --
    searching in a frame of the resident batch (features never leave the device) and in an explicit feature set."""
    oracle, ex = oracle_lib, orb
    imgs = np.stack([synthetic_frame(40), synthetic_frame(41)])
    (kpA, dA), (kpB, dB) = ex.extract_batch(imgs)
    bounds = (0.0, 640.0, 0.0, 480.0)
    rng = np.random.default_rng(9)
    nq = kpA.shape[0]
    qxy = (kpA[:, :2] + rng.normal(0, 3.0, (nq, 2))).astype(np.float32)
    qr = np.where(rng.random(nq) < 0.5, 15.0, 40.0).astype(np.float32)
    oct_ = kpA[:, 5].astype(np.int32)
    qlev = np.stack([oct_ - 1, oct_ + 1], 1).astype(np.int32)
    ref = oracle.orb_match(kpB, dB, bounds, qxy, qr, qlev, dA, max_cand=8)
    ex.match_set_frame(1, bounds)
    got = ex.match_search(qxy, qr, qlev, dA, max_cand=8)
    for k in ("cand_cnt", "best_idx", "best_dist", "best_dist2", "cand_idx", "cand_dist"):
        assert np.array_equal(got[k], ref[k]), k
    assert (ref["cand_cnt"] > 0).mean() > 0.5 and ref["cand_cnt"].max() > 8        # both the common and the truncated case occur
    # windows of many cells and many features (a wave takes 64 cells, then 64 features at a time), a window that leaves the grid, truncated candidate lists, no lists at all
    qxy2 = qxy[:96].copy(); qxy2[:8] = [[-500.0, -500.0], [5.0, 5.0], [639.0, 479.0], [320.0, 240.0], [2000.0, 100.0], [0.0, 479.0], [639.0, 0.0], [320.0, -30.0]]
    qr2 = np.where(np.arange(96) % 3 == 0, 300.0, 120.0).astype(np.float32)
    for mc in (4, 0, 256):
        ref3 = oracle.orb_match(kpB, dB, bounds, qxy2, qr2, qlev[:96], dA[:96], max_cand=mc)
        got3 = ex.match_search(qxy2, qr2, qlev[:96], dA[:96], max_cand=mc)
        for k in ("cand_cnt", "best_idx", "best_dist", "best_dist2", "cand_idx", "cand_dist"):
            assert np.array_equal(got3[k], ref3[k]), (mc, k)
    assert ref3["cand_cnt"].max() > 128                                                # more features than two rounds of a wave
    # explicit feature set, no level check (keyframe::GetFeaturesInArea)
    ref2 = oracle.orb_match(kpA, dA, bounds, qxy, qr, np.full((nq, 2), -1, np.int32), dA, max_cand=32)
    ex.match_set_features(kpA, dA, bounds)
    got2 = ex.match_search(qxy, qr, None, dA, max_cand=32)
    for k in ("cand_cnt", "best_idx", "best_dist", "best_dist2", "cand_idx", "cand_dist"):
        assert np.array_equal(got2[k], ref2[k]), k
