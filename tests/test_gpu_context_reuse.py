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
