"""One context, many problems: local BA, pose-only, global BA (partitioned band solver), local BA again -- every answer identical to
the one a fresh context gives (no state leaks between uploads: solver selection, pose-only path, band buffers, slab reuse)."""
import numpy as np
import pytest

from textslam_amd import synth, abi

pytestmark = pytest.mark.gpu


def _run(opt, kind):
    if kind == "local":
        P = synth.tiny(seed=11); o = abi.options_local(); G = P.copy(); rep = opt.LocalBundleAdjustment(G, options=o)
    elif kind == "pose":
        P = synth.config_c3(); o = abi.options_pose(); G = P.copy(); rep = opt.PoseOptim(G, options=o)
    else:
        P = synth.config_global(n_kf=300, n_pt=9000, band=8); o = abi.options_global(); o.its[0] = 4; G = P.copy(); rep = opt.GlobalBA(G, options=o)
    return G.pose.copy(), rep["iters"], rep["cost1"]


def test_context_reuse_matches_fresh_contexts():
    from textslam_amd.optimizer import Optimizer
    shared = Optimizer(0)
    for kind in ("local", "pose", "global", "local", "global", "pose"):
        a = _run(shared, kind)
        b = _run(Optimizer(0), kind)
        assert a[1] == b[1], kind
        assert np.array_equal(a[0], b[0]) and a[2] == b[2], kind          # deterministic kernels: bit-identical


def test_sliding_window_with_the_plane_cache_matches_fresh_contexts():
    """LocalBundleAdjustment is called once per new keyframe on the last keyframes (tracking.cc:828-842): a context that is told the keyframes'
    identities (tsba_problem.kf_id) keeps their pyramid planes on the device and copies only those of the keyframes it has not seen.  A
    6-window slide (12 keyframes apiece over an 18-keyframe sequence, then a jump back that evicts nothing and a window of other images under
    NEW identities): every answer bit-identical to a fresh context without identities; the cache copied each keyframe once."""
    from textslam_amd.optimizer import Optimizer
    Q = synth.make_problem(n_kf=18, n_pt=900, n_text=14, seed=21, feats=(16, 8, 6), text_targets=4, band=8)
    ids = 5000 + 3*np.arange(18)
    o = abi.options_local()
    shared = Optimizer(0)
    seen = set()
    for k0 in (0, 1, 2, 3, 4, 6, 1):
        W = synth.window_of(Q, k0, 12, kf_ids=ids)
        G1 = W.copy(); r1 = shared.LocalBundleAdjustment(G1, options=o)
        F = W.copy(); F.kf_id = None
        G2 = F.copy(); r2 = Optimizer(0).LocalBundleAdjustment(G2, options=o)
        assert r1["iters"] == r2["iters"] and r1["cost1"] == r2["cost1"], k0
        assert np.array_equal(G1.pose, G2.pose) and np.array_equal(G1.rho, G2.rho) and np.array_equal(G1.theta, G2.theta), k0
        assert np.array_equal(G1.sgood, G2.sgood) and np.array_equal(G1.tfgood, G2.tfgood), k0
        seen |= set(ids[k0:k0 + 12].tolist())
        hits, misses = shared.img_cache_stats()
        assert misses == len(seen), (k0, hits, misses)               # every keyframe crossed the bus once
    # other images under other identities (same geometry): nothing of the old windows may be used
    Q2 = synth.make_problem(n_kf=12, n_pt=500, n_text=10, seed=22, feats=(16, 8, 6), text_targets=4, band=8)
    Q2.kf_id = 9000 + np.arange(12)
    G1 = Q2.copy(); r1 = shared.LocalBundleAdjustment(G1, options=o)
    F = Q2.copy(); F.kf_id = None
    G2 = F.copy(); r2 = Optimizer(0).LocalBundleAdjustment(G2, options=o)
    assert r1["iters"] == r2["iters"] and np.array_equal(G1.pose, G2.pose) and np.array_equal(G1.theta, G2.theta)
