"""GPU parity tests of the global-BA paths (BASELINE configs 3 and 4) -- every solver the bench lines run through, under a checker.

  * partitioned band solver with the separator system by block cyclic reduction (tsba_bandcr.h), forced through tsba_debug_set:
    first LM step against scipy's banded Cholesky on the downloaded band, full trajectory against the sequential-separator and
    the single-workgroup streaming solvers;
  * config 4 at full size (5000 KF / ~500 k scene blocks): first step against scipy.linalg.solveh_banded, size-independent
    properties of the solve (cost decrease, unit quaternions, gauge, bit-reproducibility);
  * config 3 WITH text planes: parity with the oracle at 60 KF / 40 planes, properties at 500 KF x 50 k points x 1000 planes;
  * the landmark shards of the multi-GPU path on ONE device: partial S, g of nshard = 2, 3 sum to the unsharded system, and the
    whole N-rank solve (in-process communicator, one thread per rank) reproduces the 1-rank solve.
"""
import threading
import os
import sys
import numpy as np
import pytest
from scipy.linalg import solveh_banded

from textslam_amd import synth, abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from textslam_amd.optimizer import Optimizer
    g = Optimizer(0)
    yield g
    g.close()


def _free_rows(free):
    kf = np.nonzero(free)[0]
    return np.concatenate([np.arange(6*k, 6*k + 6) for k in kf])


def _check_first_step_banded(gpu, o):
    rb = gpu.reduced_band(o.initial_radius)
    assert np.abs(rb["ab"][0]).min() > 0                              # (the band solvers leave S intact)
    ref = -solveh_banded(rb["ab"], rb["g"], lower=True)
    got = rb["dp_rows"]
    assert got.size == ref.size == rb["n"] == 6*int(np.count_nonzero(rb["free"]))
    err = np.abs(got - ref).max()/np.abs(ref).max()
    assert err <= 1e-8, err
    return rb


def _same_trajectory(rep1, rep2, G1, G2, atol=1e-9):
    assert rep1["iters"] == rep2["iters"] and rep1["accepted"] == rep2["accepted"] and rep1["termination"] == rep2["termination"]
    np.testing.assert_allclose(rep1["cost1"], rep2["cost1"], rtol=1e-9)
    np.testing.assert_allclose(G1.pose, G2.pose, rtol=0, atol=atol)
    np.testing.assert_allclose(G1.rho, G2.rho, rtol=0, atol=atol)


@pytest.mark.parametrize("n_kf,band,parts,cr", [(600, 6, 8, 2), (900, 9, 17, 2), (1100, 12, 11, 2), (1500, 10, 27, 2), (1300, 7, 40, 2), (1500, 13, 9, 2), (900, 9, 17, 3), (900, 9, 17, 4)])
def test_cyclic_reduction_separator_solver(gpu, n_kf, band, parts, cr):
    """The cyclic-reduction separator solver forced at sizes where the cost model would pick the sequential separator solve: separators of
    6 .. 12 pose blocks, 7 .. 39 of them (odd and even counts, not powers of two).  cr = 2: one launch per level (tsba_bandcre.h:
    k_cre_elim / k_cre_back, several workgroups per pivot at the lower levels); cr = 3: the pivot / update / back kernels of round 1
    (tsba_bandcr.h), kept for A/B runs; cr = 4: as 2 with the separator system assembled by k_bandp_border + k_bandp_sep
    instead of the fused k_bandp_sepf."""
    P = synth.config_global(n_kf=n_kf, n_pt=40*n_kf, band=band)
    o = abi.options_global(); o.its[0] = 5
    try:
        gpu.debug_set(band_parts=parts, sep_solver=cr)
        gpu.upload(P, o)
        info = gpu.solver_info()
        assert info["band_stream"] == 1 and info["sep_cr"] == 1 and info["interiors"] == parts, info
        _check_first_step_banded(gpu, o)
        G1 = P.copy(); rep1 = gpu.GlobalBA(G1, options=o)
        assert rep1["accepted"][0] >= 3 and rep1["termination"][0] != 5 and rep1["cost1"][0] < rep1["cost0"][0]
        gpu.debug_set(band_parts=parts, sep_solver=1)                   # same interiors, separator system by the streaming solver
        gpu.upload(P, o); assert gpu.solver_info()["sep_cr"] == 0
        G2 = P.copy(); rep2 = gpu.GlobalBA(G2, options=o)
        _same_trajectory(rep1, rep2, G1, G2)
        gpu.debug_set(band_parts=1)                                     # single-workgroup streaming solver
        G3 = P.copy(); rep3 = gpu.GlobalBA(G3, options=o)
        _same_trajectory(rep1, rep3, G1, G3)
    finally:
        gpu.debug_set()


def test_small_pair_linearisation_variant(gpu):
    """k_linearize<FULL, 4> (four (target, host) pairs per wave, chosen when a pair holds <= 24 scene blocks) against the
    one-pair-per-wave kernel on the same map: the per-pair sums are formed in a different order, nothing else changes."""
    P = synth.config_global(n_kf=400, n_pt=6000, band=10)
    o = abi.options_global(); o.its[0] = 6
    try:
        gpu.upload(P, o)
        assert gpu.solver_info()["small_pairs"] == 1
        G1 = P.copy(); rep1 = gpu.GlobalBA(G1, options=o)
        gpu.debug_set(no_small_pairs=1)
        gpu.upload(P, o); assert gpu.solver_info()["small_pairs"] == 0
        G2 = P.copy(); rep2 = gpu.GlobalBA(G2, options=o)
        _same_trajectory(rep1, rep2, G1, G2, atol=1e-10)
    finally:
        gpu.debug_set()


def test_schur_four_blocks_per_wave_variant(gpu):
    """k_schur_quad (large maps: four S blocks per wave, 16 lanes each) against one wave per block (k_schur_t<1>): the reduced system
    of the first linearisation to 1e-13 (the partial sums are grouped differently), then the same LM trajectory -- on a scene-only map
    and on a map with text planes (the 3 x 3 plane blocks take the second loop of the kernel)."""
    cases = [(synth.config_global(n_kf=400, n_pt=8000, band=10), 0),
             (_text_map(150, 3000, 60, 9, (24, 12, 8), 10, 1), 1)]
    for P, use_text in cases:
        o = abi.options_global(); o.its[0] = 6; o.use_text = use_text
        try:
            gpu.upload(P, o); ra = gpu.reduced_band(o.initial_radius)
            G1 = P.copy(); rep1 = gpu.GlobalBA(G1, options=o)
            gpu.debug_set(no_schur_quad=1)
            gpu.upload(P, o); rb = gpu.reduced_band(o.initial_radius)
            G2 = P.copy(); rep2 = gpu.GlobalBA(G2, options=o)
        finally:
            gpu.debug_set()
        assert np.abs(ra["ab"] - rb["ab"]).max() <= 1e-13*np.abs(rb["ab"]).max() and np.abs(ra["ab"]).max() > 0
        np.testing.assert_allclose(ra["g"], rb["g"], rtol=0, atol=1e-13*np.abs(rb["g"]).max())
        _same_trajectory(rep1, rep2, G1, G2, atol=1e-10)
        assert rep1["accepted"][0] >= 3


def test_c6_full_size_global_ba(gpu):
    """BASELINE config 4 on one GPU, the instance bench.py --workload global_ba times (5000 KF x 70 k points, ~500 k scene blocks):
    the path the cost model picks (64 interiors, cyclic-reduction separator solve, one-wave Schur blocks, k_pose_sums,
    k_gauge_par, four pairs per wave), first LM step against scipy's banded Cholesky, then size-independent properties."""
    P = synth.config_global(n_kf=5000, n_pt=70000, band=10)
    o = abi.options_global()
    gpu.upload(P, o)
    info = gpu.solver_info()
    assert info["band_stream"] == 1 and info["sep_cr"] == 1 and info["interiors"] >= 32 and info["small_pairs"] == 1 and info["large_map"] == 1, info
    rb = _check_first_step_banded(gpu, o)
    assert rb["n"] == 6*(5000 - 2)                                    # gauge: the two initial keyframes are constant (optimizer.cc:1825-1830)
    rep = gpu.solve(); A = gpu.download(P.copy())
    assert rep["iters"][0] == 20 or rep["termination"][0] in (1, 2, 3)
    assert rep["accepted"][0] >= 5 and rep["termination"][0] != 5
    assert rep["cost1"][0] < 0.5*rep["cost0"][0]
    assert 480000 < rep["n_sblock"][0] < 540000
    q = A.pose.reshape(-1, 7)
    assert np.allclose(np.linalg.norm(q[:, :4], axis=1), 1.0, atol=1e-12)
    assert np.array_equal(q[:2], P.pose.reshape(-1, 7)[:2]) and not np.array_equal(q[2:], P.pose.reshape(-1, 7)[2:])
    assert np.all(np.isfinite(A.rho)) and np.mean(A.rho > 0) > 0.99       # (a few points behind outlier observations may flip)
    rep2 = gpu.solve(); B = gpu.download(P.copy())                     # restart from the uploaded state: bit-reproducible
    assert rep2["iters"] == rep["iters"] and rep2["cost1"] == rep["cost1"]
    assert np.array_equal(A.pose, B.pose) and np.array_equal(A.rho, B.rho)


def _text_map(n_kf, n_pt, n_text, seed, feats, band, n_levels):
    return synth.make_problem(n_kf=n_kf, n_pt=n_pt, n_text=n_text, seed=seed, feats=feats, max_targets=8, text_targets=5,
                              frozen_frac=0.0, band=band, n_levels=n_levels, rot_deg=0.2, trans_m=0.01)


def test_global_ba_with_text_planes_parity(gpu, oracle_lib):
    """BASELINE config 3 keeps its 1 k text planes in the global BA (the reference's GlobalBA switches them off, optimizer.cc:1707:
    options.use_text = 1 is the superset).  60 KF / 3000 points / 40 planes against the oracle: same LM trajectory."""
    P = _text_map(60, 3000, 40, 5, (32, 16, 8), 10, 3)
    o = abi.options_global(); o.use_text = 1; o.its[0] = 8
    G, R = P.copy(), P.copy()
    rg = gpu.GlobalBA(G, options=o)
    ro = oracle_lib.solve(R, o)
    assert rg["iters"] == ro["iters"] and rg["accepted"] == ro["accepted"] and rg["termination"] == ro["termination"]
    assert rg["n_sblock"] == ro["n_sblock"] and rg["n_tblock"] == ro["n_tblock"] and rg["n_tblock"][0] > 1000
    np.testing.assert_allclose(rg["cost0"], ro["cost0"], rtol=1e-11)
    np.testing.assert_allclose(rg["cost1"], ro["cost1"], rtol=1e-9)
    np.testing.assert_allclose(G.pose, R.pose, rtol=0, atol=1e-8)
    np.testing.assert_allclose(G.rho, R.rho, rtol=0, atol=1e-8)
    np.testing.assert_allclose(G.theta, R.theta, rtol=0, atol=1e-8)


def test_c5_full_size_global_ba_with_text(gpu):
    """BASELINE config 3 at full size: 500 KF x 50 k points x 1000 text planes, full Schur LM on one GPU (no oracle at this size:
    first step against scipy's banded Cholesky + size-independent properties)."""
    P = _text_map(500, 50000, 1000, 7, (64, 24, 12), 12, 1)
    o = abi.options_global(); o.use_text = 1
    gpu.upload(P, o)
    info = gpu.solver_info()
    assert info["band_stream"] == 1 and info["interiors"] > 1, info
    _check_first_step_banded(gpu, o)
    rep = gpu.solve(); A = gpu.download(P.copy())
    assert rep["n_tblock"][0] > 200000 and rep["n_sblock"][0] > 250000
    assert rep["accepted"][0] >= 5 and rep["termination"][0] != 5 and rep["cost1"][0] < 0.5*rep["cost0"][0]
    assert np.allclose(np.linalg.norm(A.pose.reshape(-1, 7)[:, :4], axis=1), 1.0, atol=1e-12)
    assert np.all(np.isfinite(A.theta)) and not np.array_equal(A.theta, P.theta)
    rep2 = gpu.solve(); B = gpu.download(P.copy())
    assert rep2["cost1"] == rep["cost1"] and np.array_equal(A.pose, B.pose) and np.array_equal(A.theta, B.theta)


# ------------------------------------------------------------------------------------------------ loop closures
@pytest.mark.parametrize("mode", ["ring", "reorder"])
def test_loop_closure_map_parity(gpu, oracle_lib, mode):
    """GlobalBA runs right after a loop closure (loopClosing.cc:589): the last keyframes share landmarks with the first, the
    co-visibility graph is a RING, and in keyframe order the envelope of S is the whole matrix.  Two ways through it: the ghost-row
    partition (keyframe order, the closure blocks behind the last pose, band of the open chain) and the reverse Cuthill-McKee order of
    the rows of S (band = twice the local one; what a closure that is not end-to-start takes).  120-keyframe ring against the oracle
    (which knows nothing of either)."""
    P = synth.config_global(n_kf=120, n_pt=4000, band=8, loop=True)
    o = abi.options_global(); o.its[0] = 8
    try:
        gpu.debug_set(no_ring=1 if mode == "reorder" else 0)
        gpu.upload(P, o)
        info = gpu.solver_info()
        if mode == "ring":
            assert info["ring"] == 1 and info["kf_reordered"] == 0 and info["band_rows"] <= 6*(8 + 3) and info["interiors"] == 4, info
        else:
            assert info["ring"] == 0 and info["kf_reordered"] == 1 and info["band_storage"] == 1 and info["band_rows"] <= 6*3*8, info
        G, R = P.copy(), P.copy()
        rg = gpu.GlobalBA(G, options=o); ro = oracle_lib.solve(R, o)
        assert rg["iters"] == ro["iters"] and rg["accepted"] == ro["accepted"] and rg["termination"] == ro["termination"]
        np.testing.assert_allclose(rg["cost1"], ro["cost1"], rtol=1e-9)
        np.testing.assert_allclose(G.pose, R.pose, rtol=0, atol=1e-8)
        np.testing.assert_allclose(G.rho, R.rho, rtol=0, atol=1e-8)
    finally:
        gpu.debug_set()


@pytest.mark.parametrize("n_kf,band", [(600, 8), (1500, 8)])
def test_loop_closure_map_band_solvers(gpu, n_kf, band):
    """The same on maps that take the partitioned band solver: first LM step against scipy's banded Cholesky in the reordered row space,
    and (600 keyframes) the whole trajectory against the keyframe-order solve (tsba_debug_set no_kf_reorder: the wide-band /
    dense multi-workgroup Cholesky on a (6 n_kf)^2 matrix -- what a loop closure cost before)."""
    P = synth.config_global(n_kf=n_kf, n_pt=30*n_kf, band=band, loop=True)
    o = abi.options_global(); o.its[0] = 5
    try:
        gpu.debug_set(no_ring=1)                                  # (the ghost-row partition of ring maps has its own test below)
        gpu.upload(P, o)
        info = gpu.solver_info()
        assert info["kf_reordered"] == 1 and info["band_stream"] == 1 and info["band_rows"] <= 6*3*band, info
        rb = _check_first_step_banded(gpu, o)
        assert not np.all(np.diff(rb["rowblk"][rb["rowblk"] >= 0]) > 0)       # the order really is not the keyframe order
        G1 = P.copy(); rep1 = gpu.GlobalBA(G1, options=o)
        assert rep1["accepted"][0] >= 3 and rep1["termination"][0] != 5
        if n_kf <= 600:
            gpu.debug_set(no_kf_reorder=1)
            gpu.upload(P, o)
            info = gpu.solver_info()
            assert info["kf_reordered"] == 0 and info["band_rows"] > 6*(n_kf - 40), info
            G2 = P.copy(); rep2 = gpu.GlobalBA(G2, options=o)
            _same_trajectory(rep1, rep2, G1, G2)
    finally:
        gpu.debug_set()


@pytest.mark.parametrize("n_kf,band,parts", [(600, 8, 0), (1500, 8, 0), (1500, 8, 8), (2400, 11, 0)])
def test_loop_closure_ring_partition(gpu, n_kf, band, parts):
    """One loop closure between the last and the first keyframes: the plan keeps the keyframe order, the closure blocks go to ghost rows behind
    the last pose, the partition starts and ends with a copy of the first separator and the cyclic reduction merges the two at its root
    (tsba_plan.h, tsba_bandp.h, tsba_bandcre.h) -- the band of the open chain instead of twice that under reverse Cuthill-McKee.
    Against the reordering path on the same map: same LM trajectory, same poses."""
    P = synth.config_global(n_kf=n_kf, n_pt=30*n_kf, band=band, loop=True)
    o = abi.options_global(); o.its[0] = 6
    try:
        gpu.debug_set(band_parts=parts)
        gpu.upload(P, o)
        info = gpu.solver_info()
        assert info["ring"] == 1 and info["kf_reordered"] == 0 and info["sep_cr"] == 1 and info["band_rows"] <= 6*(band + 3), info
        assert info["interiors"] & (info["interiors"] - 1) == 0 and (parts == 0 or info["interiors"] == parts), info
        G1 = P.copy(); rep1 = gpu.GlobalBA(G1, options=o)
        assert rep1["accepted"][0] >= 3 and rep1["termination"][0] != 5 and rep1["cost1"][0] < rep1["cost0"][0]
        gpu.debug_set(no_ring=1)
        gpu.upload(P, o)
        info = gpu.solver_info()
        assert info["ring"] == 0 and info["kf_reordered"] == 1, info
        G2 = P.copy(); rep2 = gpu.GlobalBA(G2, options=o)
        _same_trajectory(rep1, rep2, G1, G2)
    finally:
        gpu.debug_set()


@pytest.mark.parametrize("n_kf,k0,band,parts", [(600, 200, 8, 0), (900, 450, 8, 0), (1500, 300, 7, 0), (1500, 300, 7, 8), (1500, 1000, 7, 0), (700, 30, 8, 0)])
def test_loop_closure_with_a_tail(gpu, n_kf, k0, band, parts):
    """The usual loop closure: the last keyframes meet keyframe k0 > 0 -- a tail before the loop.  The rows stay in keyframe order, the loop's first
    poses are a separator in the MIDDLE of the chain with their ghost behind the last pose; the separator labels count from that separator (the
    tail's downwards), and the cyclic reduction ends with it and the ghost (tsba_bandp.h: bandp_part_ring).  Against the reordering path.
    (k0 = 30: a tail too short for an interior -- not a ring for the partition; the closure goes through the low-rank correction of the band solve.)"""
    P = synth.config_global(n_kf=n_kf, n_pt=20*n_kf, band=band, loop=True, loop_at=k0)
    o = abi.options_global(); o.its[0] = 6
    try:
        gpu.debug_set(band_parts=parts)
        gpu.upload(P, o)
        info = gpu.solver_info()
        if k0 < 40:                                                 # not a ring for the partition: the closure as a low-rank correction of the band solve (tsba_wb.h) ...
            assert info["ring"] == 0 and info["kf_reordered"] == 0 and info["far_band_blocks"] == band, info
            G1 = P.copy(); rep1 = gpu.GlobalBA(G1, options=o)
            gpu.debug_set(band_parts=parts, far_solver=1)           # ... against the reordering path
            gpu.upload(P, o)
            info = gpu.solver_info()
            assert info["ring"] == 0 and info["kf_reordered"] == 1 and info["far_band_blocks"] == 0, info
            G2 = P.copy(); rep2 = gpu.GlobalBA(G2, options=o)
            _same_trajectory(rep1, rep2, G1, G2, atol=1e-8)
            return
        assert info["ring"] == 1 and info["kf_reordered"] == 0 and info["sep_cr"] == 1 and info["band_rows"] <= 6*(band + 3), info
        G1 = P.copy(); rep1 = gpu.GlobalBA(G1, options=o)
        assert rep1["accepted"][0] >= 3 and rep1["termination"][0] != 5 and rep1["cost1"][0] < rep1["cost0"][0]
        gpu.debug_set(no_ring=1)
        gpu.upload(P, o)
        info = gpu.solver_info()
        assert info["ring"] == 0 and info["kf_reordered"] == 1, info
        G2 = P.copy(); rep2 = gpu.GlobalBA(G2, options=o)
        _same_trajectory(rep1, rep2, G1, G2)
    finally:
        gpu.debug_set()


@pytest.mark.parametrize("n_kf,band,seed", [(120, 8, 21), (200, 8, 4)])
def test_loop_closure_ring_with_text_planes(gpu, n_kf, band, seed):
    """The ghost-row path with text planes in the problem (their slot pairs take the second loop of the Schur kernels; 120 keyframes: one
    workgroup per S block, 200: four blocks per wave) against the reordering path."""
    P = synth.make_problem(n_kf=n_kf, n_pt=30*n_kf, n_text=n_kf//4, seed=seed, feats=(12, 8, 6), max_targets=6, text_targets=4, frozen_frac=0.0,
                           band=band, n_levels=1, rot_deg=0.2, trans_m=0.01, loop=True)
    o = abi.options_global(); o.use_text = 1; o.its[0] = 6
    try:
        gpu.upload(P, o)
        info = gpu.solver_info()
        assert info["ring"] == 1 and info["kf_reordered"] == 0, info
        G1 = P.copy(); rep1 = gpu.GlobalBA(G1, options=o)
        assert rep1["accepted"][0] >= 2 and rep1["termination"][0] != 5
        gpu.debug_set(no_ring=1)
        gpu.upload(P, o); assert gpu.solver_info()["ring"] == 0
        G2 = P.copy(); rep2 = gpu.GlobalBA(G2, options=o)
        _same_trajectory(rep1, rep2, G1, G2)
        np.testing.assert_allclose(G1.theta, G2.theta, rtol=0, atol=1e-8)
    finally:
        gpu.debug_set()


# ------------------------------------------------------------------------------------------------ N > 1 on one device
def _on_ranks(world, fn):
    """Run fn(optimizer, rank) on `world` contexts of this process, one thread each, joined through the in-process communicator."""
    from textslam_amd.optimizer import Optimizer, local_group_create, local_group_destroy
    group = local_group_create(world)
    out, err = [None]*world, [None]*world

    def run(rank):
        try:
            g = Optimizer(0)
            g.comm_init_local(group, rank, world)
            out[rank] = fn(g, rank)
            g.close()
        except Exception as e:                                          # noqa: BLE001 -- reported below
            err[rank] = e
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    local_group_destroy(group)
    assert all(e is None for e in err), err
    return out


@pytest.mark.parametrize("nshard", [2, 3])
@pytest.mark.parametrize("shape", ["dense", "band"])
def test_device_landmark_shards_sum_to_unsharded_system(nshard, shape):
    """The sharded upload on the DEVICE path: rank r keeps the landmarks hosted in its keyframe range, runs the split (multi-GPU) kernel
    sequence -- participation flags and pose sums all-reduced -- and leaves its PARTIAL S, g before the exchange of the reduced system.
    The parts of all ranks must sum to the unsharded system, every rank must derive the same band layout and the same free poses, and
    the ranks' plans must be proper parts (pairs and S blocks local to the rank's keyframe range)."""
    from textslam_amd.optimizer import Optimizer
    if shape == "dense":
        P = synth.make_problem(n_kf=24, n_pt=900, n_text=12, seed=11, feats=(12, 8, 6), max_targets=6, text_targets=4, frozen_frac=0.1, band=8, n_levels=1)
        o = abi.options_global(); o.use_text = 1
    else:
        P = synth.config_global(n_kf=300, n_pt=9000, band=8, far_frac=0.0)
        o = abi.options_global()
    radius = o.initial_radius

    def system(g, rank):
        g.upload(P, o)
        info = g.solver_info()
        r = g.reduced_system(radius) if shape == "dense" else g.reduced_band(radius)
        return (r["S"].copy() if shape == "dense" else r["ab"].copy()), r["g"].copy(), r["free"].copy(), info
    g0 = Optimizer(0); g0.comm_init(None, 0, 1)                        # unsharded, split kernel sequence (S without the pose damping)
    S0, gv0, free0, info0 = system(g0, 0); g0.close()
    parts = _on_ranks(nshard, system)
    S = np.zeros_like(S0); gv = np.zeros_like(gv0)
    for Sr, gr, fr, info in parts:
        assert Sr.shape == S0.shape and np.array_equal(fr, free0)
        assert info["band_rows"] == info0["band_rows"] and info["band_storage"] == info0["band_storage"] and info["world"] == nshard
        assert np.abs(Sr).max() > 0 and not np.allclose(Sr, S0)       # a proper part
        S += Sr; gv += gr
    assert sum(i["n_scene_candidates"] for _, _, _, i in parts) == info0["n_scene_candidates"]       # every block on exactly one rank
    if shape == "band":                                                # keyframe-range ownership keeps the pairs / S blocks local
        assert max(i["n_pair"] for _, _, _, i in parts) < 0.75*info0["n_pair"] and max(i["n_sblock"] for _, _, _, i in parts) < 0.75*info0["n_sblock"]
    scale = np.abs(S0).max()
    if shape == "dense":                                              # (rows of constant poses: untouched storage on both sides)
        m = _free_rows(free0).size
        assert np.abs(S[:m, :m] - S0[:m, :m]).max() <= 1e-12*scale
    else:
        assert np.abs(S - S0).max() <= 1e-12*scale
    assert np.abs(gv - gv0).max() <= 1e-12*np.abs(gv0).max()


def _solve_on_ranks(P, o, world, call="GlobalBA"):
    """The whole N-rank solve on one device: `world` contexts, one host thread each, collectives through the in-process group."""
    def solve(g, rank):
        G = P.copy()
        rep = getattr(g, call)(G, options=o)
        return G, rep, g.solver_info()
    return _on_ranks(world, solve)


@pytest.mark.parametrize("world", [2, 3])
def test_multi_rank_global_ba_matches_single_rank(gpu, oracle_lib, world):
    """The product's N > 1 path end to end -- sharded upload, split kernel sequence, all-reduce of S, g, the pose sums and the
    landmark deltas, identical decisions on every rank -- against the 1-rank solve and the oracle."""
    P = synth.config_global(n_kf=30, n_pt=1500, band=6)
    o = abi.options_global()
    G1 = P.copy(); rep1 = gpu.GlobalBA(G1, options=o)
    ranks = _solve_on_ranks(P, o, world)
    for G, rep, info in ranks:
        assert info["world"] == world
        _same_trajectory(rep1, rep, G1, G)
        assert rep["n_sblock"] == rep1["n_sblock"]                      # block counts are global
    for G, rep, _ in ranks[1:]:                                         # every rank ends with the same map
        assert np.array_equal(G.pose, ranks[0][0].pose) and np.array_equal(G.rho, ranks[0][0].rho)
    R = P.copy(); ro = oracle_lib.solve(R, o)
    assert ranks[0][1]["iters"] == ro["iters"] and ranks[0][1]["accepted"] == ro["accepted"]
    np.testing.assert_allclose(ranks[0][0].pose, R.pose, rtol=0, atol=1e-8)


def test_multi_rank_band_solver_and_text(gpu):
    """N = 2 on a map that takes the band path (300 KF: band storage of S is what the ranks exchange) and on a window with text."""
    P = synth.config_global(n_kf=300, n_pt=9000, band=8)
    o = abi.options_global(); o.its[0] = 6
    G1 = P.copy(); rep1 = gpu.GlobalBA(G1, options=o)
    for G, rep, info in _solve_on_ranks(P, o, 2):
        assert info["band_storage"] == 1
        _same_trajectory(rep1, rep, G1, G)
    Q = synth.make_problem(n_kf=24, n_pt=900, n_text=12, seed=11, feats=(12, 8, 6), max_targets=6, text_targets=4, frozen_frac=0.1, band=8, n_levels=1)
    o = abi.options_global(); o.use_text = 1; o.its[0] = 8
    G1 = Q.copy(); rep1 = gpu.GlobalBA(G1, options=o)
    for G, rep, info in _solve_on_ranks(Q, o, 2):
        _same_trajectory(rep1, rep, G1, G)
        np.testing.assert_allclose(G.theta, G1.theta, rtol=0, atol=1e-9)


@pytest.mark.parametrize("k0", [0, 150])
def test_multi_rank_ring_map(gpu, k0):
    """N = 2 on a loop-closure map (k0 = 150: a tail before the loop): every rank recognises the ring from ALL observations, keeps its shard's
    closure blocks in the ghost rows, and the ghost rows travel with the packed band."""
    P = synth.config_global(n_kf=400, n_pt=12000, band=8, loop=True, loop_at=k0)
    o = abi.options_global(); o.its[0] = 6
    G1 = P.copy(); rep1 = gpu.GlobalBA(G1, options=o)
    assert gpu.solver_info()["ring"] == 1
    for G, rep, info in _solve_on_ranks(P, o, 2):
        assert info["ring"] == 1 and info["world"] == 2 and info["band_storage"] == 1, info
        _same_trajectory(rep1, rep, G1, G)


def test_multi_gpu_rccl_two_ranks():
    """Real RCCL over two devices (skipped on a one-GPU box): 2-rank solve against the 1-rank answer."""
    import subprocess, sys, os, json
    from textslam_amd.optimizer import Optimizer, TsbaError
    try:
        Optimizer(1).close()                                            # (tsba_create fails with TSBA_ERR_DEVICE beyond the device count)
    except TsbaError:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "global_ba", "--kf", "300", "--pts", "9000",
                        "--steps", "2", "--warmup", "1", "--check-single"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["rccl_ranks"] == 2 and line["config"]["max_pose_diff_vs_single_rank"] <= 1e-9


@pytest.mark.parametrize("kw", [dict(n_kf=600, n_pt=12000, band=8), dict(n_kf=900, n_pt=18000, band=9, loop=True), dict(n_kf=900, n_pt=18000, band=8, closures=2),
                                dict(n_kf=700, n_pt=14000, band=8, far_frac=0.02), dict(n_kf=1500, n_pt=30000, band=7, loop=True, loop_at=300)])
def test_schur_lists_built_on_the_device_equal_the_host_lists(gpu, kw):
    """Large maps build the slot pairs of the S blocks on the device (csrc/tsba_devplan.h: a wave per block walks pose a's slots and looks for pose b among
    each landmark's slots) instead of in two host passes over 2 M pairs.  Same entries in the same (landmark) order, so the Schur sums -- and with them the
    whole solve -- are bit-identical to the host-built lists: open chain, ring, ring with a tail, two closures and scattered long-range points (the
    band / long-range split restricts the pairs to one cluster of a landmark)."""
    P = synth.config_global(**kw)
    o = abi.options_global(); o.its[0] = 6
    try:
        gpu.debug_set(host_pair_lists=1, far_solver=2 if (kw.get("far_frac") or kw.get("closures")) else 0)
        G1 = P.copy(); r1 = gpu.GlobalBA(G1, options=o); i1 = gpu.solver_info()
        gpu.debug_set(far_solver=2 if (kw.get("far_frac") or kw.get("closures")) else 0)
        G2 = P.copy(); r2 = gpu.GlobalBA(G2, options=o); i2 = gpu.solver_info()
    finally:
        gpu.debug_set()
    assert i1 == i2 and i1["band_storage"] == 1, (i1, i2)
    assert r1["iters"] == r2["iters"] and r1["accepted"] == r2["accepted"] and r1["cost0"] == r2["cost0"] and r1["cost1"] == r2["cost1"]
    assert np.array_equal(G1.pose, G2.pose) and np.array_equal(G1.rho, G2.rho)
    assert r1["cost1"][0] < 0.5*r1["cost0"][0]


@pytest.mark.parametrize("shape", ["c4", "tiny", "init_pair", "window31"])
def test_schur_lists_built_on_the_device_equal_the_host_lists_on_windows(gpu, shape):
    """Round 6: windows take the device-built slot pairs as well (they are the largest part of the plan a one-shot tsba_local_ba call builds on its calling thread).  Same entries
    in the same order as the host lists: the whole three-pass solve -- poses, inverse depths, planes, flags, costs -- is the same bits, through the one-shot call (levels of the
    later passes staged over the copy stream while the first pass runs) and through upload + solve."""
    P = {"c4": synth.config_c4, "tiny": synth.tiny, "init_pair": synth.init_pair, "window31": lambda: synth.make_problem(31, 2500, 20, 77, feats=(16, 8, 6))}[shape]()
    o = abi.options_init() if shape == "init_pair" else abi.options_local()
    outs = []
    try:
        for mode in (1, 2):                                   # (2: the device lists at any size; by default only from 4096 scene observations on)
            gpu.debug_set(host_pair_lists=mode)
            G = P.copy(); r = gpu.LocalBundleAdjustment(G, options=o)
            gpu.upload(P, o); r2 = gpu.solve(); G2 = gpu.download(P.copy())
            outs.append((G, r, G2, r2))
    finally:
        gpu.debug_set()
    (Ga, ra, Ga2, ra2), (Gb, rb, Gb2, rb2) = outs
    for X, Y in ((Ga, Gb), (Ga2, Gb2), (Ga, Gb2)):
        assert np.array_equal(X.pose, Y.pose) and np.array_equal(X.rho, Y.rho) and np.array_equal(X.theta, Y.theta)
        assert np.array_equal(X.sgood, Y.sgood) and np.array_equal(X.tobs_good, Y.tobs_good) and np.array_equal(X.tfgood, Y.tfgood)
    assert ra["iters"] == rb["iters"] == rb2["iters"] and ra["cost1"] == rb["cost1"] == rb2["cost1"] and ra["accepted"] == rb["accepted"]


@pytest.mark.parametrize("name,variants", [
    ("mid_global_long_range", [dict(far_solver=2), dict(far_solver=3, pcg_block=1), dict(far_solver=1), dict(far_solver=1, no_band_stream=1)]),
    ("mid_global_ring", [dict(), dict(no_ring=1), dict(no_ring=1, no_kf_reorder=1)]),
    ("mid_global_two_closures", [dict(far_solver=2), dict(far_solver=3), dict(far_solver=1)]),
])
def test_global_ba_against_the_committed_fixtures(gpu, name, variants):
    """tsba_global_ba against COMMITTED vectors (tests/golden/mid_global_*.npz: the oracle's LM trace, final parameters and first-linearisation
    gradient on a map with 3 % long-range points, on a ring and on a map with two closures), through every solver path that can take the map:
    conjugate gradients with and without the low-rank correction, the reordered band, the wide-band / dense Cholesky, the ghost-row partition."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    P, o = make_golden.make_global_case(name)
    assert make_golden.global_digest(P) == str(g["digest"])
    paths = set()
    try:
        for kw in variants:
            gpu.debug_set(**kw)
            G = P.copy(); rep = gpu.GlobalBA(G, options=o); tr = gpu.lm_trace(0)
            paths.add(rep["solver_path"])
            assert rep["iters"] == g["iters"].tolist() and rep["accepted"] == g["accepted"].tolist() and rep["termination"] == g["termination"].tolist(), (kw, rep)
            assert np.array_equal(tr[:, 3], g["trace"][:, 3]), kw
            np.testing.assert_allclose(tr[:, 0], g["trace"][:, 0], rtol=1e-9, err_msg=str(kw))
            np.testing.assert_allclose(tr[:, 2], g["trace"][:, 2], rtol=1e-7, err_msg=str(kw))       # (the radius follows the gain ratio, a ratio of differences)
            np.testing.assert_allclose(rep["cost1"], g["cost1"], rtol=1e-9)
            np.testing.assert_allclose(G.pose, g["pose"], rtol=0, atol=1e-8)
            np.testing.assert_allclose(G.rho, g["rho"], rtol=0, atol=1e-8)
            assert rep["pcg_unconverged"] == 0
    finally:
        gpu.debug_set()
    assert len(paths) >= 2, paths                                       # (the variants really went through different solvers)


def test_a_landmark_listed_twice_at_a_keyframe_keeps_the_host_lists(gpu, oracle_lib):
    """The device build of the S blocks' slot-pair lists relies on one slot per (landmark, keyframe) -- what the reference's maps have.  An input that lists
    an observation twice is still a valid problem (two residual blocks): the plan notices and keeps the host lists; the solve matches the oracle."""
    P = synth.config_global(n_kf=200, n_pt=5000, band=8)
    kf, pt, fl, uv = P.sobs_kf[0], P.sobs_pt[0], P.sobs_flag[0], P.sobs_uv0[0]
    pick = np.arange(50, len(kf), len(kf)//40)[:40]                  # forty observations listed twice (next to the original: the lists stay keyframe-major)
    pick = pick[P.pt_host[pt[pick]] != kf[pick]]
    rep_idx = np.sort(np.concatenate([np.arange(len(kf)), pick]))
    P.sobs_kf[0], P.sobs_pt[0], P.sobs_flag[0] = kf[rep_idx].copy(), pt[rep_idx].copy(), fl[rep_idx].copy()
    P.sobs_uv0[0] = (uv[rep_idx] + 0.25*(np.arange(len(rep_idx)) % 2)[:, None]).copy()       # (the second listing with a slightly different pixel)
    o = abi.options_global(); o.its[0] = 5
    G, R = P.copy(), P.copy()
    rg = gpu.GlobalBA(G, options=o)
    ro = oracle_lib.solve(R, o)
    assert rg["n_sblock"] == ro["n_sblock"] and rg["iters"] == ro["iters"] and rg["accepted"] == ro["accepted"]
    np.testing.assert_allclose(rg["cost0"], ro["cost0"], rtol=1e-11)
    np.testing.assert_allclose(rg["cost1"], ro["cost1"], rtol=1e-9)
    np.testing.assert_allclose(G.pose, R.pose, rtol=0, atol=1e-8)
