"""GPU parity tests of the global BA on maps with LONG-RANGE coupling (SURVEY 8d's C6 as written: band co-visibility + 1 % long-range
observations; several loop closures in one session, loopClosing.cc:587-591): the reduced camera system is a band plus scattered 6x6 blocks,
solved by conjugate gradients preconditioned with the band solver (csrc/tsba_pcg.h) -- no (6 n_kf)^2 matrix.

  * first LM step: S dp = -g checked against the assembled matrix (band part M + the blocks of E) -- dense numpy solve at 600 / 900
    keyframes, residual of the sparse system at 5000;
  * whole solve: same LM trajectory (iterations, accepted steps, termination, cost, poses) as the direct solvers this replaces -- the
    reordered band (reverse Cuthill-McKee) where that exists, the wide-band / dense Cholesky otherwise -- and as the CPU oracle;
  * two ranks (in-process communicator): every rank derives the same block list, the blocks travel with the band.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from textslam_amd import synth, abi
from test_gpu_global import _on_ranks, _same_trajectory

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from textslam_amd.optimizer import Optimizer
    g = Optimizer(0)
    yield g
    g.close()


def _sparse_system(gpu, radius):
    """First linearisation: the reduced system as a scipy sparse matrix over the compressed free-pose rows, g, and the pose step by row."""
    rb = gpu.reduced_band(radius)
    n, bw, ab = rb["n"], rb["bw"], rb["ab"]
    diags = [ab[0]] + [ab[d][:n - d] for d in range(1, bw + 1)]
    A = sp.diags(diags, [-d for d in range(bw + 1)], shape=(n, n), format="lil")
    a, b, v = gpu.far_blocks()
    row = rb["rowblk"]
    nblk = 0
    for q in range(len(a)):
        ia, ib = row[a[q]], row[b[q]]
        if ia < 0 or ib < 0:
            continue
        A[6*ib:6*ib + 6, 6*ia:6*ia + 6] += v[q].T                 # lower triangle: rows of the later keyframe (a block of E may lie inside the band)
        nblk += 1
    A = sp.csr_matrix(A)
    A = A + sp.tril(A, -1).T
    return A, rb, nblk


@pytest.mark.parametrize("n_kf,far,closures", [(600, 0.02, 0), (900, 0.0, 2), (900, 0.01, 3)])
def test_first_step_against_the_assembled_system(gpu, n_kf, far, closures):
    P = synth.config_global(n_kf=n_kf, n_pt=20*n_kf, band=8, far_frac=far, closures=closures)
    o = abi.options_global()
    try:
        gpu.debug_set(far_solver=2)
        gpu.upload(P, o)
        info = gpu.solver_info()
        assert info["far_band_blocks"] == 8 and info["far_blocks"] > 0 and info["band_storage"] == 1 and info["band_rows"] == 48 and info["kf_reordered"] == 0, info
        A, rb, nblk = _sparse_system(gpu, o.initial_radius)
        assert nblk > 0
        ref = -np.linalg.solve(A.toarray(), rb["g"])
        err = np.abs(rb["dp_rows"] - ref).max()/np.abs(ref).max()
        assert err <= 1e-8, err
    finally:
        gpu.debug_set()


@pytest.mark.parametrize("n_kf,far,closures", [(600, 0.02, 0), (900, 0.0, 2), (1500, 0.01, 2), (1500, 0.0, 6)])
def test_same_trajectory_as_the_direct_solvers(gpu, n_kf, far, closures):
    """The map through the conjugate gradients and through what it replaces (far_solver = 1: the reordered band where reverse Cuthill-McKee
    finds one, the wide-band Cholesky on the dense matrix otherwise).  Six closures: more than 64 keyframes touched -- the low-rank correction
    with k up to 768."""
    P = synth.config_global(n_kf=n_kf, n_pt=20*n_kf, band=8, far_frac=far, closures=closures)
    o = abi.options_global(); o.its[0] = 8
    try:
        gpu.debug_set(far_solver=2)
        G1 = P.copy(); rep1 = gpu.GlobalBA(G1, options=o)
        info, st = gpu.solver_info(), gpu.pcg_stats()
        assert info["far_band_blocks"] == 8 and info["far_blocks"] > 0, info
        assert st["systems"] >= rep1["iters"][0] and st["hit_cap"] == 0 and 0 < st["iterations"] and st["max_iterations"] <= 150, st
        if far == 0.0:                                              # closures only: the low-rank correction makes the band solve (nearly) exact
            assert st["iterations"] <= 3*st["systems"], st
        gpu.debug_set(far_solver=1)
        G2 = P.copy(); rep2 = gpu.GlobalBA(G2, options=o)
        assert gpu.solver_info()["far_band_blocks"] == 0
        _same_trajectory(rep1, rep2, G1, G2, atol=1e-8)
        assert rep1["cost1"][0] < 0.5*rep1["cost0"][0]
    finally:
        gpu.debug_set()


@pytest.mark.parametrize("far,closures", [(0.01, 2), (0.0, 3), (0.03, 0)])
def test_conjugate_gradient_variants_agree(gpu, far, closures):
    """Five ways through the same systems: the single-vector iteration with the preconditioner applied by the single-vector solve phase
    (tsba_bandsv.h: the production path), by re-running the factorisation, by the many-column solve phase with one column (tsba_bandms.h), the
    enlarged conjugate gradients (32 columns per application), and the default -- which on a map whose long-range coupling is a few loop
    closures corrects the band solve by their low-rank part exactly (tsba_wb.h) and needs next to no iterations.  Same LM trajectory every way."""
    P = synth.config_global(n_kf=900, n_pt=18000, band=8, far_frac=far, closures=closures)
    o = abi.options_global(); o.its[0] = 5
    try:
        runs = []
        for kw in (dict(far_solver=3, pcg_block=1), dict(far_solver=3, pcg_block=1, pcg_refactor=1), dict(far_solver=3, pcg_block=1, pcg_refactor=2), dict(far_solver=3, pcg_block=2), dict(far_solver=2),
                   dict(far_solver=3, pcg_block=1, sv_per_level=1), dict(far_solver=3, pcg_block=1, sv_per_level=4)):
            gpu.debug_set(band_parts=16, sep_solver=2, **kw)
            G = P.copy(); rep = gpu.GlobalBA(G, options=o)
            info = gpu.solver_info()
            assert info["far_band_blocks"] == 8 and info["sep_cr"] == 1 and info["interiors"] == 16, info
            runs.append((G, rep, gpu.pcg_stats()))
        for G, rep, st in runs[1:]:
            _same_trajectory(runs[0][1], rep, runs[0][0], G, atol=1e-8)
            assert st["hit_cap"] == 0 and st["systems"] == runs[0][2]["systems"], st
        for k in (1, 2):
            assert abs(runs[0][2]["iterations"] - runs[k][2]["iterations"]) <= runs[0][2]["systems"], [r[2] for r in runs]
        assert runs[3][2]["iterations"] < runs[0][2]["iterations"], [r[2] for r in runs]
        # the separator tree of the solve phase as a launch per level (the sixth run) instead of one launch: the same bits all the way
        assert np.array_equal(runs[5][0].pose, runs[0][0].pose) and np.array_equal(runs[5][0].rho, runs[0][0].rho) and runs[5][2]["iterations"] == runs[0][2]["iterations"]
        # ... and the iteration's update step as a launch of its own (the seventh) instead of inside the first kernel of the preconditioner application
        assert np.array_equal(runs[6][0].pose, runs[0][0].pose) and np.array_equal(runs[6][0].rho, runs[0][0].rho) and runs[6][2]["iterations"] == runs[0][2]["iterations"]
        if far == 0.0:                                              # loop closures only: the low-rank correction makes the band solve (nearly) exact
            assert runs[4][2]["iterations"] <= 3*runs[4][2]["systems"], runs[4][2]
    finally:
        gpu.debug_set()


def test_parity_with_the_oracle(gpu, oracle_lib):
    """130 keyframes, 3 % long-range points, against the CPU oracle (which knows nothing of the split)."""
    P = synth.config_global(n_kf=130, n_pt=4000, band=8, far_frac=0.03)
    o = abi.options_global(); o.its[0] = 8
    try:
        gpu.debug_set(far_solver=2)
        G, R = P.copy(), P.copy()
        rg = gpu.GlobalBA(G, options=o)
        assert gpu.solver_info()["far_band_blocks"] == 8
        ro = oracle_lib.solve(R, o)
        assert rg["iters"] == ro["iters"] and rg["accepted"] == ro["accepted"] and rg["termination"] == ro["termination"]
        np.testing.assert_allclose(rg["cost1"], ro["cost1"], rtol=1e-9)
        np.testing.assert_allclose(G.pose, R.pose, rtol=0, atol=1e-8)
        np.testing.assert_allclose(G.rho, R.rho, rtol=0, atol=1e-8)
    finally:
        gpu.debug_set()


def test_two_ranks(gpu):
    """N = 2: the ranks derive the same list of blocks outside the band from ALL observations, each assembles its landmarks' part, the
    blocks are all-reduced next to the band and every rank iterates on the same summed system."""
    P = synth.config_global(n_kf=600, n_pt=12000, band=8, far_frac=0.02)
    o = abi.options_global(); o.its[0] = 6
    G1 = P.copy(); rep1 = gpu.GlobalBA(G1, options=o)
    info1 = gpu.solver_info()
    assert info1["far_band_blocks"] == 8

    def solve(g, rank):
        G = P.copy()
        rep = g.GlobalBA(G, options=o)
        return G, rep, g.solver_info(), g.pcg_stats()
    for G, rep, info, st in _on_ranks(2, solve):
        assert info["world"] == 2 and info["far_band_blocks"] == 8 and info["far_blocks"] == info1["far_blocks"], info
        assert st["hit_cap"] == 0
        _same_trajectory(rep1, rep, G1, G, atol=1e-8)


def test_c6_with_long_range_observations_full_size(gpu):
    """SURVEY 8d's C6 as written: 5000 keyframes, ~500 k observations, band co-visibility + 1 % long-range points.  First step: residual of
    the assembled sparse system; the solve: cost decrease, unit quaternions, no dense matrix (band storage), bit-reproducible."""
    P = synth.config_global(n_kf=5000, n_pt=70000, band=10, far_frac=0.01)
    o = abi.options_global()
    gpu.upload(P, o)
    info = gpu.solver_info()
    assert info["far_band_blocks"] in (8, 10) and info["far_blocks"] > 5000 and info["band_storage"] == 1 and info["band_rows"] == 6*info["far_band_blocks"] and info["interiors"] > 32, info
    A, rb, nblk = _sparse_system(gpu, o.initial_radius)
    assert nblk > 5000
    res = A @ rb["dp_rows"] + rb["g"]
    assert np.abs(res).max() <= 1e-8*np.abs(rb["g"]).max(), (np.abs(res).max(), np.abs(rb["g"]).max())
    rep = gpu.solve(); G = gpu.download(P.copy())
    st = gpu.pcg_stats()
    assert st["hit_cap"] == 0 and st["systems"] >= rep["iters"][0], st
    assert rep["cost1"][0] < 0.05*rep["cost0"][0]
    np.testing.assert_allclose(np.linalg.norm(G.pose[:, :4], axis=1), 1.0, atol=1e-12)
    rep2 = gpu.solve(); G2 = gpu.download(P.copy())
    assert rep2["iters"] == rep["iters"] and np.array_equal(G.pose, G2.pose) and np.array_equal(G.rho, G2.rho)


@pytest.mark.parametrize("n_kf,band,parts,T", [(900, 9, 17, 1), (900, 9, 17, 70), (1100, 12, 11, 5), (1500, 10, 27, 64), (600, 6, 8, 130), (1300, 7, 40, 3),
                                                (600, 8, -30, 66), (700, 11, 9, 40), (260, 6, -20, 2)])
def test_multi_right_hand_side_solve_phase(gpu, n_kf, band, parts, T):
    """The solve phase of the partitioned band solver on other right-hand sides (csrc/tsba_bandms.h: what the conjugate gradients apply as
    preconditioner): M X = R for T random columns with the factor of the first linearisation, against scipy's banded Cholesky on the
    downloaded band.  Open chains, separators of 6 .. 12 pose blocks, 7 .. 39 of them (product-form separator kernels up to 11 blocks, the substitution
    kernels at 12), incl. maps on which the device settles for fewer interiors than the host asked for."""
    from scipy.linalg import solveh_banded
    P = synth.config_global(n_kf=n_kf, n_pt=40*n_kf, band=band)
    o = abi.options_global()
    try:
        gpu.debug_set(band_parts=abs(parts), sep_solver=2)       # parts < 0: more interiors asked for than the map holds -- the device takes fewer (bandp_part)
        gpu.upload(P, o)
        info = gpu.solver_info()
        assert info["band_stream"] == 1 and (parts < 0 or info["interiors"] == parts) and info["sep_cr"] == 1 and info["far_band_blocks"] == 0, info
        rb = gpu.reduced_band(o.initial_radius)
        rng = np.random.default_rng(5)
        R = rng.standard_normal((rb["n"], T))*np.abs(rb["g"]).max()
        R[:, 0] = -rb["g"]                                          # column 0: the system the factorisation itself solved
        X = gpu.multi_solve(R)
        ref = solveh_banded(rb["ab"], R, lower=True)
        err = np.abs(X - ref).max(axis=0)/np.abs(ref).max(axis=0)
        assert err.max() <= 1e-8, err
        assert np.abs(X[:, 0] - rb["dp_rows"]).max() <= 1e-9*np.abs(rb["dp_rows"]).max()
        # the single-vector solve phase (csrc/tsba_bandsv.h: a lane owns a row) on two of the columns
        for k in (0, T - 1):
            x1 = gpu.multi_solve(R[:, k:k + 1].copy(), single=True)[:, 0]
            assert np.abs(x1 - ref[:, k]).max() <= 1e-8*np.abs(ref[:, k]).max(), (k, np.abs(x1 - ref[:, k]).max()/np.abs(ref[:, k]).max())
            # the separator tree as ONE launch (workgroups polling each other's results, k_sv_cre_tree: production) and as a launch per level: the same bits
            gpu.debug_set(band_parts=abs(parts), sep_solver=2, sv_per_level=1)
            x2 = gpu.multi_solve(R[:, k:k + 1].copy(), single=True)[:, 0]
            gpu.debug_set(band_parts=abs(parts), sep_solver=2)
            x3 = gpu.multi_solve(R[:, k:k + 1].copy(), single=True)[:, 0]
            assert np.array_equal(x1, x2) and np.array_equal(x1, x3)
    finally:
        gpu.debug_set()


def test_solver_status_in_the_report(gpu):
    """tsba_report says which solver ran and whether the iterative one converged (the reference's sparse Cholesky either solves the system or the
    step fails, optimizer.cc:1833-1845): an iteration cap that is hit makes the LM trial an INVALID step (trust region halved), never an accepted one."""
    P = synth.config_global(n_kf=600, n_pt=12000, band=8, far_frac=0.02)
    o = abi.options_global(); o.its[0] = 6
    try:
        gpu.debug_set(far_solver=2)
        G = P.copy(); rep = gpu.GlobalBA(G, options=o)
        assert rep["solver_path"] == 7 and rep["pcg_systems"] >= rep["iters"][0] and rep["pcg_iterations"] > 0, rep
        assert rep["pcg_unconverged"] == 0 and rep["pcg_max_iterations"] <= 150 and rep["status"] == 0, rep
        st = gpu.pcg_stats()
        assert st["iterations"] == rep["pcg_iterations"] and st["systems"] == rep["pcg_systems"] and st["hit_cap"] == 0
        gpu.debug_set(far_solver=2, pcg_max_it=3)                     # three iterations cannot reach 1e-10
        G2 = P.copy(); rep2 = gpu.GlobalBA(G2, options=o)
        tr = gpu.lm_trace(0)
        assert rep2["pcg_unconverged"] >= 1, rep2
        bad = tr[:, 3] == -1.0
        assert bad.sum() >= rep2["pcg_unconverged"] - 1 and np.all(np.isnan(tr[bad, 0])), tr       # (a solve cut off by the end of the pass has no trial)
        assert rep2["accepted"][0] == int((tr[:, 3] == 1.0).sum())
        if rep2["termination"][0] == 5:
            assert rep2["status"] == -3 and np.array_equal(G2.pose, P.pose)      # five invalid steps in a row: TSBA_ERR_NUMERIC, the parameters stay at the last accepted state
    finally:
        gpu.debug_set()
    # the other paths name themselves too
    Q = synth.config_global(n_kf=900, n_pt=18000, band=8, closures=2)
    try:
        gpu.debug_set(far_solver=2)
        rq = gpu.GlobalBA(Q.copy(), options=o)
        assert rq["solver_path"] == 6 and rq["pcg_unconverged"] == 0, rq
    finally:
        gpu.debug_set()
    R = synth.config_global(n_kf=600, n_pt=12000, band=8)
    rr = gpu.GlobalBA(R.copy(), options=o)
    assert rr["solver_path"] in (3, 4) and rr["pcg_systems"] == 0, rr
    rl = gpu.LocalBundleAdjustment(synth.tiny().copy())
    assert rl["solver_path"] == 0, rl
    rp = gpu.PoseOptim(synth.config_c3().copy())
    assert rp["solver_path"] == 8, rp


@pytest.mark.parametrize("n_kf,band,parts", [(900, 9, 17), (1100, 12, 11), (600, 8, -30), (1500, 10, 27)])
def test_separator_back_substitution_in_one_launch(gpu, n_kf, band, parts):
    """The direct solve of a chain map: the separators' back substitution as ONE launch through the inverse factors and products of the solve phase
    (k_sv_linv + k_cre_back_tree, workgroups polling their neighbours' solutions: production) against a substitution launch per level (k_cre_back,
    tsba_debug_options.sv_per_level = 2), on maps with long-range blocks (where the iterative path builds those operands anyway) -- the first step against
    scipy's sparse solve of the assembled system and against each other, the same LM run."""
    from scipy.sparse.linalg import spsolve
    P = synth.config_global(n_kf=n_kf, n_pt=40*n_kf, band=band, far_frac=0.01)      # (the one-launch form runs where the iterative path needs its operands anyway: maps with long-range blocks)
    o = abi.options_global(); o.its[0] = 6
    runs = []
    try:
        for per_level in (0, 2):
            gpu.debug_set(band_parts=abs(parts), sep_solver=2, sv_per_level=per_level, far_solver=3, pcg_block=1)
            gpu.upload(P, o)
            info = gpu.solver_info()
            assert info["band_stream"] == 1 and info["sep_cr"] == 1 and info["far_band_blocks"] > 0, info
            A, rb, nblk = _sparse_system(gpu, o.initial_radius)
            ref = spsolve(sp.csc_matrix(A), -rb["g"])
            assert nblk > 0 and np.abs(rb["dp_rows"] - ref).max() <= 1e-7*np.abs(ref).max(), (per_level, np.abs(rb["dp_rows"] - ref).max()/np.abs(ref).max())     # (conjugate gradients to 1e-10 in the M^-1 norm)
            rep = gpu.solve(); G = gpu.download(P.copy())
            runs.append((rb["dp_rows"].copy(), rep, G))
    finally:
        gpu.debug_set()
    assert np.abs(runs[0][0] - runs[1][0]).max() <= 1e-7*np.abs(runs[1][0]).max()
    _same_trajectory(runs[1][1], runs[0][1], runs[1][2], runs[0][2], atol=1e-7)
