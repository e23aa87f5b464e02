"""The kernels whose workgroups hand results over INSIDE one launch (value polling: k_solve_back on windows, k_sv_cre_tree / k_sv_tree_back /
k_cre_back_tree on large maps, the k_lin_mid roles) -- made safe and observable:

  * they are chosen only where the whole grid is resident at once (occupancy x compute units, checked per context); a device too small for the grid
    (tsba_debug_options.assume_cus) takes the launch-per-step path and computes the same bits;
  * a wait that runs into its bound is counted (tsba_report.poll_timeouts) on top of failing the linear solve;
  * a second context that keeps the device busy (an ORB extractor looping on a 16-frame batch from another host thread, as TextSLAM's tracking thread does
    beside a local / global BA) changes nothing: identical LM traces, identical parameters, zero time-outs.

The reference is deterministic (optimizer.cc:1599,1838: num_threads = 1)."""
import threading
import numpy as np
import pytest

from textslam_amd import synth, abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from textslam_amd.optimizer import Optimizer
    g = Optimizer(0)
    yield g
    g.close()


def _solve(gpu, P, o, n_passes):
    gpu.upload(P, o)
    rep = gpu.solve()
    G = gpu.download(P.copy())
    return rep, G, [gpu.lm_trace(ps) for ps in range(n_passes)]


def _same(a, b):
    ra, Ga, ta = a; rb, Gb, tb = b
    assert ra["iters"] == rb["iters"] and ra["accepted"] == rb["accepted"] and ra["termination"] == rb["termination"]
    assert ra["cost1"] == rb["cost1"]
    assert np.array_equal(Ga.pose, Gb.pose) and np.array_equal(Ga.rho, Gb.rho) and np.array_equal(Ga.theta, Gb.theta)
    for x, y in zip(ta, tb):
        assert np.array_equal(x, y, equal_nan=True)


class _Busy:
    """An ORB extractor of its own context (own stream) running batches back to back on another host thread."""
    def __init__(self, n_frames=16):
        from textslam_amd.orbextractor import ORBextractor, synthetic_frame
        self.ex = ORBextractor(device=0)
        self.ex.upload(np.stack([synthetic_frame(s) for s in range(n_frames)]))
        self.stop = threading.Event(); self.runs = 0
        self.th = threading.Thread(target=self._loop, daemon=True)

    def _loop(self):
        while not self.stop.is_set():
            self.ex.run(); self.runs += 1

    def __enter__(self):
        self.th.start()
        return self

    def __exit__(self, *a):
        self.stop.set(); self.th.join(timeout=30)
        self.ex.close()


class _BusyBA:
    """A second bundle-adjustment context solving a 700-keyframe map over and over on another host thread (TextSLAM's loop-closing thread runs GlobalBA while
    the mapping thread runs local windows, loopClosing.cc:587-591 / tracking.cc:826-842): workgroups of 768 threads with 100+ KB of LDS, the kind that keeps
    a polling launch's workgroups off the device."""
    def __init__(self):
        from textslam_amd.optimizer import Optimizer
        self.g = Optimizer(0)
        P = synth.config_global(n_kf=700, n_pt=14000, band=8); o = abi.options_global(); o.its[0] = 20
        self.g.upload(P, o)
        self.stop = threading.Event(); self.runs = 0; self.reps = set()
        self.th = threading.Thread(target=self._loop, daemon=True)

    def _loop(self):
        while not self.stop.is_set():
            rep = self.g.solve(); self.runs += 1
            self.reps.add((tuple(rep["iters"]), tuple(rep["accepted"]), tuple(rep["cost1"]), rep["poll_timeouts"]))

    def __enter__(self):
        self.th.start()
        return self

    def __exit__(self, *a):
        self.stop.set(); self.th.join(timeout=60)
        self.g.close()


def test_window_and_maps_beside_a_second_bundle_adjustment(gpu, map_cache):
    """Local windows and 5000-keyframe maps on one context while a second context solves a 700-keyframe map in a loop: both sides bit-identical to their runs
    alone, no time-outs."""
    Pw, ow = synth.config_c4(), abi.options_local()
    alone_w = _solve(gpu, Pw, ow, ow.n_passes)
    maps = {}
    for name, kw in (("open_chain", {}), ("long_range", dict(far_frac=0.01))):
        P = map_cache(n_kf=5000, n_pt=70000, band=10, **kw); o = abi.options_global(); o.its[0] = 5
        maps[name] = (P, o, _solve(gpu, P, o, 1))
    with _BusyBA() as busy:
        while busy.runs < 2:
            pass
        for _ in range(15):
            r = _solve(gpu, Pw, ow, ow.n_passes)
            assert r[0]["poll_timeouts"] == 0
            _same(r, alone_w)
        for name, (P, o, alone) in maps.items():
            for _ in range(6):
                r = _solve(gpu, P, o, 1)
                assert r[0]["poll_timeouts"] == 0 and r[0]["pcg_unconverged"] == 0, (name, r[0])
                _same(r, alone)
        n = busy.runs
    assert n > 4 and len(busy.reps) == 1 and next(iter(busy.reps))[3] == 0, (n, busy.reps)      # the other side: one and the same result every time


def test_orb_extraction_beside_a_busy_bundle_adjustment(oracle_lib):
    """The other direction: the ORB extractor's output while a bundle adjustment loops on the device -- keypoints and descriptors bit-exact (vs the oracle)."""
    from textslam_amd.orbextractor import ORBextractor, synthetic_frame
    imgs = np.stack([synthetic_frame(80 + s) for s in range(8)])
    ex = ORBextractor(1000, 1.2, 8, 20, 7, device=0)
    ref = ex.extract_batch(imgs)
    kp_o, d_o = oracle_lib.orb_extract(imgs[0])
    assert np.array_equal(ref[0][0][:, :2], kp_o[:, :2]) and np.array_equal(ref[0][1], d_o)
    with _BusyBA() as busy:
        while busy.runs < 2:
            pass
        r0, k = busy.runs, 0
        while k < 25 or (busy.runs <= r0 + 1 and k < 4000):          # at least 25 batches, and until the other side has finished two more solves beside them
            res = ex.extract_batch(imgs)                             # (a batch takes ~1 ms, a solve of the other side ~15 ms alone and several times that when more
            for (ka, da), (kb, db) in zip(res, ref):                 # contexts share the device: a fixed count of 25 batches once ended before its first solve did)
                assert np.array_equal(ka, kb) and np.array_equal(da, db)
            k += 1
        assert busy.runs > r0 + 1                                    # (the other side did run meanwhile)
    ex.close()


def test_window_solve_beside_a_busy_context(gpu):
    P, o = synth.config_c4(), abi.options_local()
    alone = _solve(gpu, P, o, o.n_passes)
    assert alone[0]["poll_timeouts"] == 0 and alone[0]["solver_path"] == 0
    with _Busy() as busy:
        while busy.runs < 2:
            pass
        for _ in range(20):
            r0 = busy.runs
            beside = _solve(gpu, P, o, o.n_passes)
            assert beside[0]["poll_timeouts"] == 0, beside[0]
            _same(beside, alone)
        assert busy.runs > r0 or busy.runs > 2                      # (the other context did run meanwhile)


def test_long_range_map_beside_a_busy_context(gpu, map_cache):
    P = map_cache(n_kf=5000, n_pt=70000, band=10, far_frac=0.01); o = abi.options_global(); o.its[0] = 6
    alone = _solve(gpu, P, o, 1)
    assert alone[0]["poll_timeouts"] == 0 and alone[0]["pcg_unconverged"] == 0 and alone[0]["pcg_iterations"] > 0
    with _Busy() as busy:
        while busy.runs < 2:
            pass
        for _ in range(3):
            beside = _solve(gpu, P, o, 1)
            assert beside[0]["poll_timeouts"] == 0 and beside[0]["pcg_unconverged"] == 0, beside[0]
            _same(beside, alone)
        assert busy.runs > 2


@pytest.mark.parametrize("kw", [dict(n_kf=5000, n_pt=70000, band=10), dict(n_kf=5000, n_pt=70000, band=10, loop=True), dict(n_kf=5000, n_pt=70000, band=10, closures=2),
                                dict(n_kf=500, n_pt=50000, band=12)], ids=["open_chain", "ring", "two_closures", "c5"])
def test_direct_map_solves_beside_a_busy_context(gpu, map_cache, kw):
    """The direct solves of the large maps (partitioned band solver + cyclic reduction; on the 5000-keyframe chain the separator back-substitution is one polling
    launch, k_cre_back_tree; two closures: + the low-rank correction)."""
    P = map_cache(**kw); o = abi.options_global(); o.its[0] = 6
    alone = _solve(gpu, P, o, 1)
    assert alone[0]["poll_timeouts"] == 0
    with _Busy() as busy:
        while busy.runs < 2:
            pass
        for _ in range(25):                                          # (round 5: 1 run in 15 differed before the siblings of a cyclic-reduction pivot waited for each other's loads, k_cre_elim)
            beside = _solve(gpu, P, o, 1)
            assert beside[0]["poll_timeouts"] == 0, beside[0]
            _same(beside, alone)


def test_pose_optim_beside_busy_contexts(gpu):
    """PoseOptim (tracking, once per frame) runs a pass's LM steps in ONE launch whose workgroups poll each other's sums (k_pose_pass): beside an ORB extractor
    and beside a second bundle adjustment looping on other host threads -- what TextSLAM's mapping and loop-closing threads do to the tracking thread -- the
    same bits as alone, no time-outs; with a device assumed too small for the launch (assume_cus = 1) the launch-per-step path takes over."""
    P, o = synth.config_c3(), abi.options_pose()
    alone = _solve(gpu, P, o, o.n_passes)
    assert alone[0]["poll_timeouts"] == 0 and sum(alone[0]["iters"]) > 0
    for busy_cls in (_Busy, _BusyBA):
        with busy_cls() as busy:
            while busy.runs < 2:
                pass
            for _ in range(40):
                r = _solve(gpu, P, o, o.n_passes)
                assert r[0]["poll_timeouts"] == 0, r[0]
                _same(r, alone)
    try:
        gpu.debug_set(assume_cus=1)
        small = _solve(gpu, P, o, o.n_passes)
    finally:
        gpu.debug_set()
    assert small[0]["poll_timeouts"] == 0 and small[0]["iters"] == alone[0]["iters"] and small[0]["accepted"] == alone[0]["accepted"]
    np.testing.assert_allclose(small[1].pose, alone[1].pose, rtol=0, atol=1e-12)       # (another compilation of the same arithmetic: last-bit differences in the sums)


@pytest.mark.parametrize("case", ["c4", "window_31", "pose", "open_chain", "long_range", "ring", "c5", "dense_100"])
def test_results_do_not_depend_on_what_other_kernels_left_in_lds(gpu, map_cache, case):
    """What a kernel finds in LDS is whatever the last workgroup on that compute unit left there: this context's own kernels when the device is otherwise idle
    (the same bytes every run -- a read of never-written LDS goes unnoticed), another context's next to it.  tsba_debug_options.lds_poison fills the LDS of every
    compute unit with NaNs / 1e300 / 0x5a bytes before EVERY launch of the solve: same traces, same parameters, bit for bit."""
    its = None
    if case == "c4":
        P, o = synth.config_c4(), abi.options_local()
    elif case == "window_31":
        P, o = synth.make_problem(n_kf=31, n_pt=1800, n_text=15, seed=71, feats=(16, 8, 6)), abi.options_local()
    elif case == "pose":
        P, o = synth.config_c3(), abi.options_pose()
    elif case == "c5":
        P, o, its = map_cache(n_kf=500, n_pt=50000, band=12), abi.options_global(), 4
    elif case == "dense_100":
        P, o, its = synth.config_global(n_kf=100, n_pt=3000, band=100), abi.options_global(), 4
    else:
        kw = dict(open_chain={}, long_range=dict(far_frac=0.01), ring=dict(loop=True))[case]
        P, o, its = map_cache(n_kf=5000, n_pt=70000, band=10, **kw), abi.options_global(), 3
    if its:
        o.its[0] = its
    runs = []
    try:
        for pz in (0, 1, 2, 3):
            gpu.debug_set(lds_poison=pz)
            runs.append(_solve(gpu, P, o, o.n_passes))
    finally:
        gpu.debug_set()
    for r in runs[1:]:
        assert r[0]["poll_timeouts"] == 0
        _same(r, runs[0])


@pytest.mark.parametrize("case", ["window", "long_range", "open_chain"])
def test_a_device_too_small_for_the_grid_takes_the_launch_per_step_path(gpu, map_cache, case):
    """assume_cus = 4: no polling launch fits (30 workgroups of k_solve_back, 127 + 128 of the separator tree) -- the solves go through k_solve_t + k_back +
    k_decide / the launch per level and give the same bits (windows) / the same LM run to the tolerance the per-level tests state (maps: the product form
    of the separator back substitution rounds differently from the substitution, tests/test_gpu_global.py)."""
    if case == "window":
        P, o, npass = synth.config_c4(), abi.options_local(), 3
    else:
        P = map_cache(n_kf=5000, n_pt=70000, band=10, **({"far_frac": 0.01} if case == "long_range" else {})); o = abi.options_global(); o.its[0] = 4; npass = 1
    ref = _solve(gpu, P, o, npass)
    try:
        gpu.debug_set(assume_cus=4)
        small = _solve(gpu, P, o, npass)
    finally:
        gpu.debug_set()
    assert small[0]["poll_timeouts"] == 0
    if case == "window":
        _same(small, ref)
    else:
        assert small[0]["iters"] == ref[0]["iters"] and small[0]["accepted"] == ref[0]["accepted"]
        assert abs(small[0]["cost1"][0] - ref[0]["cost1"][0]) <= 1e-6*ref[0]["cost1"][0]
        assert np.array_equal(small[2][0][:, 3], ref[2][0][:, 3])                     # the same decisions
