"""GPU parity of the loop-closure optimisers (libtsloop.so through the C ABI) against the CPU oracle.  fp64 on both sides, different
derivative routes (closed tangent-space forms on the device, chained ambient Jacobians in the oracle) and summation orders:
converged Sim3 within 1e-8 relative, identical iteration counts / termination / inlier sets."""
import numpy as np
import pytest

from textslam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lo():
    from textslam_amd.loop import LoopOptimizer
    return LoopOptimizer(0)


@pytest.mark.parametrize("seed,n,outl", [(1, 300, 0.1), (2, 60, 0.0), (3, 1500, 0.25), (4, 9, 0.0)])
def test_optimize_sim3_parity(lo, oracle_lib, seed, n, outl):
    m = synth.sim3_matches(seed=seed, n=n, outlier_frac=outl)
    a = (m["P1"], m["uv1"], m["P2"], m["uv2"], m["inliers"], m["sim0"], m["K"])
    ng, sg, ig, rg = lo.OptimizeSim3(*a)
    no, so, io, ro = oracle_lib.optimize_sim3(*a)
    assert (rg["iters"], rg["accepted"], rg["termination"]) == (ro["iters"], ro["accepted"], ro["termination"])
    np.testing.assert_allclose(rg["cost0"], ro["cost0"], rtol=1e-12)
    np.testing.assert_allclose(rg["cost1"], ro["cost1"], rtol=1e-9)
    np.testing.assert_allclose(sg, so, rtol=0, atol=1e-8)
    assert ng == no and np.array_equal(ig, io)
    if outl > 0:
        assert ng < n


def test_optimize_sim3_edge_cases(lo, oracle_lib):
    m = synth.sim3_matches(seed=5, n=30)
    n, sim, inl, rep = lo.OptimizeSim3(m["P1"], m["uv1"], m["P2"], m["uv2"], np.zeros(30, np.uint8), m["sim0"], m["K"])
    assert n == 0 and rep["termination"] == 5 and rep["status"] == -3                 # nothing to optimise
    n, sim, inl, rep = lo.OptimizeSim3(np.zeros((0, 3)), np.zeros((0, 2)), np.zeros((0, 3)), np.zeros((0, 2)), np.zeros(0, np.uint8), m["sim0"], m["K"])
    assert n == 0 and rep["termination"] == 5
    # restart from the solution: converges at once (function tolerance), same answer on both sides
    a = (m["P1"], m["uv1"], m["P2"], m["uv2"], m["inliers"], m["sim0"], m["K"])
    n1, s1, i1, r1 = lo.OptimizeSim3(*a)
    n2, s2, i2, r2 = lo.OptimizeSim3(m["P1"], m["uv1"], m["P2"], m["uv2"], i1.astype(np.uint8), s1, m["K"])
    no, so, io, ro = oracle_lib.optimize_sim3(m["P1"], m["uv1"], m["P2"], m["uv2"], i1.astype(np.uint8), s1, m["K"])
    assert r2["iters"] == ro["iters"] <= 3 and np.allclose(s2, so, atol=1e-9) and n2 == no
    from textslam_amd.loop import LoopError
    bad = m["sim0"].copy(); bad[7] = 0.0
    with pytest.raises(LoopError):
        lo.OptimizeSim3(m["P1"], m["uv1"], m["P2"], m["uv2"], m["inliers"], bad, m["K"])


@pytest.mark.parametrize("n_kf", [12, 40, 150])
def test_optimize_loop_parity(lo, oracle_lib, n_kf):
    """7 (n_kf - 3) unknowns: 63 -> the one-workgroup LDS solver; 259 and 1029 -> the multi-workgroup blocked Cholesky.  Numeric-diff
    Jacobians on both sides (rounding amplified by 1 / step): poses within 1e-6, identical LM trajectories."""
    g = synth.pose_graph(seed=n_kf, n_kf=n_kf)
    a = (g["pose"], g["fixed"], g["edge_i"], g["edge_j"], g["meas"])
    xg, rg = lo.OptimizeLoop(*a)
    xo, ro = oracle_lib.optimize_loop(*a)
    assert (rg["iters"], rg["accepted"], rg["termination"]) == (ro["iters"], ro["accepted"], ro["termination"])
    np.testing.assert_allclose(rg["cost0"], ro["cost0"], rtol=1e-9)
    np.testing.assert_allclose(rg["cost1"], ro["cost1"], rtol=1e-5)
    np.testing.assert_allclose(xg, xo, rtol=0, atol=1e-6)
    fx = g["fixed"].astype(bool)
    assert np.array_equal(xg[fx], g["pose"][fx])


def test_optimize_loop_edge_cases(lo):
    from textslam_amd.loop import LoopError
    g = synth.pose_graph(seed=1, n_kf=10)
    x, rep = lo.OptimizeLoop(g["pose"], np.ones(10, np.uint8), g["edge_i"], g["edge_j"], g["meas"])      # every keyframe constant
    assert rep["status"] == -3 and rep["termination"] == 5 and np.array_equal(x, g["pose"])
    with pytest.raises(LoopError):
        lo.OptimizeLoop(g["pose"], g["fixed"], np.array([0, 99], np.int32), np.array([1, 2], np.int32), g["meas"][:2])
    # only the covisibility connections (no loop connection): still a valid problem, same answer as the oracle
    import oracle
    keep = np.arange(len(g["edge_i"]) - 9)
    a = (g["pose"], g["fixed"], g["edge_i"][keep], g["edge_j"][keep], g["meas"][keep])
    x, rep = lo.OptimizeLoop(*a); xo, ro = oracle.optimize_loop(*a)
    assert rep["iters"] == ro["iters"] and rep["cost1"] <= rep["cost0"] and np.allclose(x, xo, atol=1e-6)


@pytest.mark.parametrize("n_kf", [12, 150])
def test_optimize_loop_first_step_tight(lo, oracle_lib, n_kf):
    """ONE LM iteration from the same start: the step is a function of the first linearisation only (numeric-diff Jacobians, gradient,
    damped normal equations), before any divergence of the two LM trajectories can build up -- poses and cost within 1e-8 / 1e-9 where
    the 20-iteration end state is only comparable to 1e-6."""
    g = synth.pose_graph(seed=n_kf, n_kf=n_kf)
    o = lo.default_options_loop(); o.max_it = 1
    a = (g["pose"], g["fixed"], g["edge_i"], g["edge_j"], g["meas"])
    xg, rg = lo.OptimizeLoop(*a, options=o)
    xo, ro = oracle_lib.optimize_loop(*a, options=o)
    assert (rg["iters"], rg["accepted"]) == (ro["iters"], ro["accepted"]) == (1, 1)
    np.testing.assert_allclose(rg["cost0"], ro["cost0"], rtol=1e-12)
    np.testing.assert_allclose(rg["cost1"], ro["cost1"], rtol=1e-9)
    np.testing.assert_allclose(xg, xo, rtol=0, atol=1e-8)
    assert np.abs(xg - g["pose"]).max() > 1e-4                              # a real step was taken
