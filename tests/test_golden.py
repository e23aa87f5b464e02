"""Oracle vs the committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py)."""
import os
import sys
import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", sorted(make_golden.CASES))
def test_oracle_reproduces_golden(oracle_lib, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    P, o = make_golden.make_case(name)
    assert make_golden.input_digest(P) == str(g["digest"]), "synthetic generator drifted: regenerate fixtures deliberately"
    ev = oracle_lib.evaluate(P, o, int(g["level"]))
    np.testing.assert_allclose(ev["resid"], g["resid"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(ev["jac_scene"], g["jac_scene"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(ev["jac_text"], g["jac_text"], rtol=1e-11, atol=1e-9)
    Q = P.copy()
    rep = oracle_lib.solve(Q, o)
    assert rep["iters"] == g["iters"].tolist()
    np.testing.assert_allclose(rep["cost1"], g["cost1"], rtol=1e-10)
    np.testing.assert_allclose(Q.pose, g["pose"], rtol=0, atol=1e-10)
    assert np.array_equal(Q.sgood, g["sgood"]) and np.array_equal(Q.tfgood, g["tfgood"]) and np.array_equal(Q.tobs_good, g["tobs_good"])


@pytest.mark.parametrize("name", sorted(make_golden.GLOBAL_CASES))
def test_oracle_reproduces_global_golden(oracle_lib, name):
    """Global BA on a map with long-range points / after a loop closure / with two closures: the committed LM trace (every trial's candidate cost, model cost
    change, radius and decision), final parameters and first-linearisation gradient -- through the dense built-in path AND through the block-sparse
    storage with the plugged solver."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    P, o = make_golden.make_global_case(name)
    assert make_golden.global_digest(P) == str(g["digest"]), "synthetic generator drifted: regenerate fixtures deliberately"
    rb = oracle_lib.reduced_blocks(P, o, 0, o.initial_radius)
    assert len(rb["br"]) == int(g["n_blocks"]) and np.array_equal(rb["free_idx"], g["free_idx"])
    np.testing.assert_allclose(rb["g"], g["g"], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(rb["cost"], float(g["cost_lin"]), rtol=1e-13)
    np.testing.assert_allclose(np.abs(rb["val"]).sum(), float(g["abs_sum_S"]), rtol=1e-12)
    for plugged in (False, True):
        if plugged:
            oracle_lib.set_sparse_solver(oracle_lib.sparse_solver)
        try:
            Q = P.copy(); rep, tr = oracle_lib.solve_traced(Q, o)
        finally:
            oracle_lib.set_sparse_solver(None)
        assert rep["iters"] == g["iters"].tolist() and rep["accepted"] == g["accepted"].tolist() and rep["termination"] == g["termination"].tolist()
        assert np.array_equal(tr[0][:, 3], g["trace"][:, 3])
        np.testing.assert_allclose(tr[0][:, :3], g["trace"][:, :3], rtol=1e-9 if plugged else 1e-11)
        np.testing.assert_allclose(Q.pose, g["pose"], rtol=0, atol=1e-8 if plugged else 1e-10)


def test_orb_oracle_reproduces_golden(oracle_lib):
    from textslam_amd.orbextractor import synthetic_frame
    g = np.load(os.path.join(GOLD, "orb_frame.npz"))
    kp, desc = oracle_lib.orb_extract(synthetic_frame(int(g["seed"])))
    assert np.array_equal(kp, g["kp"]) and np.array_equal(desc, g["desc"])
