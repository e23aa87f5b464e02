"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Tolerances (fp64 on both sides; the GPU contracts FMAs and sums in a different order):
  residuals            |d| <= 1e-10                      (pixels x weight / normalised intensity x weight)
  analytic Jacobians   rel 1e-10 of the block scale
  reduced system S, g  rel 1e-9
  LM trajectory        same iteration / acceptance counts, final cost rel 1e-9
  parameters           |d| <= 1e-8 (LM path divergence is amplified by the conditioning of the window)
  outlier flags        identical
"""
import os
import sys
import numpy as np
import pytest

from textslam_amd import synth, abi

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gpu():
    from textslam_amd.optimizer import Optimizer
    return Optimizer(0)


def _rel(a, b):
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))


def _check_eval(gpu, oracle, P, o, level):
    eo, eg = oracle.evaluate(P, o, level), gpu.evaluate(P, o, level)
    assert (eo["ns"], eo["nt"]) == (eg["ns"], eg["nt"])
    np.testing.assert_allclose(eg["resid"], eo["resid"], rtol=0, atol=1e-10)
    if P.n_tobs:
        np.testing.assert_allclose(eg["musigma"], eo["musigma"], rtol=1e-11, atol=1e-10)
    if eo["ns"]:
        assert _rel(eg["jac_scene"], eo["jac_scene"]) < 1e-10
    if eo["nt"]:
        assert _rel(eg["jac_text"], eo["jac_text"]) < 1e-10
    return eo


def _check_solve(gpu, oracle, P, o, call, atol=1e-8, rtol_cost=1e-9):
    G, R = P.copy(), P.copy()
    rep_g = call(G, o)
    rep_o = oracle.solve(R, o)
    assert rep_g["iters"] == rep_o["iters"] and rep_g["accepted"] == rep_o["accepted"]
    assert rep_g["termination"] == rep_o["termination"]
    np.testing.assert_allclose(rep_g["cost0"][0], rep_o["cost0"][0], rtol=1e-11)        # same linearisation point
    np.testing.assert_allclose(rep_g["cost0"], rep_o["cost0"], rtol=max(rtol_cost, 1e-9))
    np.testing.assert_allclose(rep_g["cost1"], rep_o["cost1"], rtol=rtol_cost)
    assert rep_g["n_sblock"] == rep_o["n_sblock"] and rep_g["n_tblock"] == rep_o["n_tblock"]
    np.testing.assert_allclose(G.pose, R.pose, rtol=0, atol=atol)
    np.testing.assert_allclose(G.rho, R.rho, rtol=0, atol=atol)
    np.testing.assert_allclose(G.theta, R.theta, rtol=0, atol=atol)
    assert np.array_equal(G.sgood, R.sgood) and np.array_equal(G.tobs_good, R.tobs_good) and np.array_equal(G.tfgood, R.tfgood)
    assert rep_g["n_bad_scene"] == rep_o["n_bad_scene"] and rep_g["n_bad_tfeat"] == rep_o["n_bad_tfeat"]
    return G, rep_g


def test_eval_parity_all_levels(gpu, oracle_lib):
    P = synth.tiny()
    o = abi.options_local()
    for l in range(3):
        _check_eval(gpu, oracle_lib, P, o, l)
    o.filter_good = 0
    _check_eval(gpu, oracle_lib, P, o, 0)


def test_eval_matches_reference_style_numeric_jacobian(gpu, oracle_lib):
    """HIP analytic text Jacobian vs the oracle's Ceres CENTRAL numeric differentiation (what the reference computes)."""
    P = synth.tiny(seed=3)
    o = abi.options_local()
    eg = gpu.evaluate(P, o, 0)
    o.text_jacobian = 1
    en = oracle_lib.evaluate(P, o, 0)
    rel = np.abs(eg["jac_text"] - en["jac_text"]) / np.abs(en["jac_text"]).max()
    assert np.median(rel) < 1e-9 and np.mean(rel < 1e-6) > 0.995


def test_reduced_system_parity(gpu, oracle_lib):
    P = synth.tiny(seed=9, n_kf=6, n_pt=120, n_text=5)
    o = abi.options_local()
    ro = oracle_lib.reduced_system(P, o, o.levels[0], o.initial_radius)
    gpu.upload(P, o)
    rg = gpu.reduced_system(o.initial_radius)
    assert np.array_equal(np.nonzero(rg["free"])[0], np.nonzero(ro["free_idx"] >= 0)[0])
    m = 6 * ro["nf"]
    assert _rel(rg["S"][:m, :m], ro["S"]) < 1e-9 and _rel(rg["g"][:m], ro["g"]) < 1e-9
    assert rg["cost"] == pytest.approx(ro["cost"], rel=1e-12)
    dp = -np.linalg.solve(ro["S"], ro["g"])
    free = np.nonzero(rg["free"])[0]
    idx = np.concatenate([np.arange(6*k, 6*k + 6) for k in free])
    assert _rel(rg["dp"][idx], dp) < 1e-8


@pytest.mark.parametrize("seed", [7, 21, 33])
def test_local_ba_parity(gpu, oracle_lib, seed):
    P = synth.tiny(seed=seed, n_kf=6, n_pt=150, n_text=5)
    _check_solve(gpu, oracle_lib, P, abi.options_local(), lambda G, o: gpu.LocalBundleAdjustment(G, options=o))


def test_local_ba_notreachwin_gauge(gpu, oracle_lib):
    P = synth.tiny(seed=5)
    o = abi.options_local(abi.STATE_NOTREACHWIN)      # only the two initial keyframes are constant
    G, _ = _check_solve(gpu, oracle_lib, P, o, lambda G, o: gpu.LocalBundleAdjustment(G, options=o))
    assert np.array_equal(G.pose.reshape(-1, 7)[:2], P.pose.reshape(-1, 7)[:2])
    assert not np.array_equal(G.pose.reshape(-1, 7)[2], P.pose.reshape(-1, 7)[2])


def test_pose_optim_parity_c3(gpu, oracle_lib):
    P = synth.config_c3()
    G, rep = _check_solve(gpu, oracle_lib, P, abi.options_pose(), lambda G, o: gpu.PoseOptim(G, options=o))
    assert np.array_equal(G.rho, P.rho) and np.array_equal(G.theta, P.theta)       # landmarks are frozen
    assert np.abs(G.pose - P.truth["pose"]).max() < np.abs(P.pose - P.truth["pose"]).max()


@pytest.mark.parametrize("shape", ["c3", "no_text", "no_text_planes", "small"])
def test_pose_optim_call_with_the_single_frame_plan_agrees(gpu, oracle_lib, shape):
    """The per-frame tsba_pose_optim call writes its plan down on the calling thread (build_plan_single_frame), stages every level with the upload and leaves as one
    host-to-device copy (round 6); tsba_debug_options.host_pair_lists = 1 takes the generic builder on plan threads with the later levels staged during the solve (until
    round 6).  Same lists (tests/test_band_partition.py compares their checksums on the CPU), so the same bits -- one-shot, repeated on the same context, and against the oracle."""
    if shape == "c3": P, o = synth.config_c3(), abi.options_pose()
    elif shape == "no_text": P, o = synth.config_c3(seed=5), abi.options_pose(); o.use_text = 0
    elif shape == "no_text_planes":
        P, o = synth.make_problem(1, 400, 0, 12, frozen_frac=1.0, max_targets=1), abi.options_pose()
        if P.n_levels < 3: o.n_passes = 1; o.levels[0] = 0                     # (a map without planes is built with one level)
    else: P, o = synth.make_problem(1, 40, 3, 11, feats=(8, 6, 4), frozen_frac=1.0, n_out=2, max_targets=1, text_targets=1), abi.options_pose()
    outs = []
    try:
        for mode in (1, 0, 0):
            gpu.debug_set(host_pair_lists=mode)
            G = P.copy(); r = gpu.PoseOptim(G, options=o); outs.append((G, r))
    finally:
        gpu.debug_set()
    for G, r in outs[1:]:
        assert np.array_equal(G.pose, outs[0][0].pose) and np.array_equal(G.sgood, outs[0][0].sgood) and np.array_equal(G.tfgood, outs[0][0].tfgood)
        assert r["iters"] == outs[0][1]["iters"] and r["cost1"] == outs[0][1]["cost1"] and r["accepted"] == outs[0][1]["accepted"]
    R = P.copy(); ro = oracle_lib.solve(R, o)
    assert outs[1][1]["iters"] == ro["iters"]
    np.testing.assert_allclose(outs[1][0].pose, R.pose, rtol=0, atol=1e-7)


def test_scene_only_global_style(gpu, oracle_lib):
    P = synth.config_global(n_kf=12, n_pt=600, band=6)
    _check_solve(gpu, oracle_lib, P, abi.options_global(), lambda G, o: gpu.GlobalBA(G, options=o))


def test_c1_plumbing_parity(gpu, oracle_lib):
    P = synth.config_c1()
    _check_solve(gpu, oracle_lib, P, abi.options_local(), lambda G, o: gpu.LocalBundleAdjustment(G, options=o), atol=1e-7)


def test_c4_full_size_parity_and_properties(gpu, oracle_lib):
    """The headline window (20 KF x 5000 pts x 100 planes): parity with the oracle plus size-independent properties."""
    P = synth.config_c4()
    o = abi.options_local()
    G, rep = _check_solve(gpu, oracle_lib, P, o, lambda G, o: gpu.LocalBundleAdjustment(G, options=o), atol=1e-7)
    assert all(c1 < c0 for c0, c1 in zip(rep["cost0"], rep["cost1"]))             # every pass reduces its cost
    assert np.all(G.sgood <= P.sgood) and np.all(G.tfgood <= P.tfgood)            # flags only go good -> bad
    assert np.array_equal(G.pose.reshape(-1, 7)[:3], P.pose.reshape(-1, 7)[:3])   # gauge: first 3 participating KFs
    assert np.array_equal(G.rho[P.pt_host < 0], P.rho[P.pt_host < 0])             # frozen landmarks untouched
    assert np.array_equal(G.theta[P.text_host < 0], P.theta[P.text_host < 0])
    assert np.allclose(np.linalg.norm(G.pose.reshape(-1, 7)[:, :4], axis=1), 1.0, atol=1e-12)
    # determinism: the resident solve restarts from the uploaded state and is bit-reproducible
    gpu.upload(P, o)
    gpu.solve(); A = gpu.download(P.copy())
    gpu.solve(); B = gpu.download(P.copy())
    assert np.array_equal(A.pose, B.pose) and np.array_equal(A.rho, B.rho) and np.array_equal(A.theta, B.theta)
    assert np.array_equal(A.pose, G.pose)


@pytest.mark.parametrize("name", ["tiny_local", "tiny_scene", "tiny_pose"])
def test_against_golden_fixtures(gpu, name):
    import make_golden
    g = np.load(os.path.join(GOLD, name + ".npz"))
    P, o = make_golden.make_case(name)
    assert make_golden.input_digest(P) == str(g["digest"])
    ev = gpu.evaluate(P, o, int(g["level"]))
    np.testing.assert_allclose(ev["resid"], g["resid"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(ev["jac_scene"], g["jac_scene"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(ev["jac_text"], g["jac_text"], rtol=1e-9, atol=1e-7)
    G = P.copy()
    rep = gpu.PoseOptim(G, options=o) if P.n_kf == 1 else gpu.LocalBundleAdjustment(G, options=o)
    assert rep["iters"] == g["iters"].tolist()
    np.testing.assert_allclose(rep["cost1"], g["cost1"], rtol=1e-9)
    np.testing.assert_allclose(G.pose, g["pose"], rtol=0, atol=1e-8)
    assert np.array_equal(G.sgood, g["sgood"]) and np.array_equal(G.tfgood, g["tfgood"]) and np.array_equal(G.tobs_good, g["tobs_good"])


def test_edge_cases(gpu, oracle_lib):
    # every observation flagged bad: nothing to optimise, zero iterations, parameters untouched
    P = synth.tiny(seed=4)
    P.sgood[:] = 0; P.tobs_good[:] = 0
    G = P.copy(); rep = gpu.LocalBundleAdjustment(G)
    assert rep["iters"] == [0, 0, 0] and np.array_equal(G.pose, P.pose) and np.array_equal(G.rho, P.rho)
    # no text at all / text disabled (bFlag_noText)
    P = synth.tiny(seed=6)
    o = abi.options_local(); o.use_text = 0
    _check_solve(gpu, oracle_lib, P, o, lambda G, o: gpu.LocalBundleAdjustment(G, options=o))
    # outlier pass disabled (bFlag_rapid)
    o = abi.options_local(); o.outlier_scene = o.outlier_text = 0
    G, rep = _check_solve(gpu, oracle_lib, P, o, lambda G, o: gpu.LocalBundleAdjustment(G, options=o))
    assert np.array_equal(G.sgood, P.sgood)
    # a text box that projects outside the image: sigma = 0 -> its residuals vanish (nume_BAText.h:85-90)
    P = synth.tiny(seed=8)
    P.text_box_ray[0] += 5.0
    _check_solve(gpu, oracle_lib, P, abi.options_local(), lambda G, o: gpu.LocalBundleAdjustment(G, options=o))


def test_error_codes(gpu):
    from textslam_amd.optimizer import TsbaError
    P = synth.tiny()
    o = abi.options_local(); o.text_jacobian = 1
    with pytest.raises(TsbaError):
        gpu.upload(P, o)
    o = abi.options_local(); o.levels[0] = 3
    with pytest.raises(TsbaError):
        gpu.upload(P, o)
    with pytest.raises(TsbaError):
        gpu.PoseOptim(P)                     # n_kf != 1


def test_global_ba_large_reduced_system(gpu, oracle_lib):
    """40 keyframes -> 228 x 228 reduced system: the multi-workgroup MFMA Cholesky path (does not fit the LDS solver)."""
    P = synth.config_global(n_kf=40, n_pt=2000, band=8)
    _check_solve(gpu, oracle_lib, P, abi.options_global(), lambda G, o: gpu.GlobalBA(G, options=o))


def test_multi_gpu_kernel_sequence_single_process(oracle_lib):
    """The sharded (multi-GPU) kernel sequence with world size 1: split sums -> exchange buffers -> decision, damping
    added after the (here trivial) all-reduce.  Once without a communicator, once through a 1-rank RCCL communicator."""
    from textslam_amd.optimizer import Optimizer
    P = synth.config_global(n_kf=30, n_pt=1500, band=6)
    o = abi.options_global()
    for use_rccl in (False, True):
        g = Optimizer(0)
        if use_rccl:
            g.comm_init(g.comm_unique_id(), 0, 1)
            g.lib.tsba_comm_init(g.ctx, None, 0, 1)          # world stays 1: force the split sequence on top of the communicator
        else:
            g.comm_init(None, 0, 1)
        _check_solve(g, oracle_lib, P, o, lambda G, oo: g.GlobalBA(G, options=oo))
        Q = synth.tiny(seed=7)                               # and a small local window through the same sequence
        _check_solve(g, oracle_lib, Q, abi.options_local(), lambda G, oo: g.LocalBundleAdjustment(G, options=oo))
        g.close()


def test_init_ba_parity(gpu, oracle_lib):
    """optimizer::InitBA (rows R4 / R8): unweighted, Huber 3, levels 3,2,1,0, host keyframe constant."""
    P = synth.init_pair(seed=5)
    # two views with every depth free leave the global scale unobservable: the reduced system is singular along that gauge
    # direction up to the LM damping, so round-off differences are amplified -- looser tolerance than the windowed problems
    G, rep = _check_solve(gpu, oracle_lib, P, abi.options_init(), lambda G, o: gpu.InitBA(G, options=o), atol=1e-4, rtol_cost=1e-5)
    assert rep["n_passes"] == 4 and np.array_equal(G.pose.reshape(-1, 7)[0], P.pose.reshape(-1, 7)[0])


def test_init_ba_first_linearisation_and_gauge_invariants(gpu, oracle_lib):
    """InitBA where it CAN be tight.  (1) Every pyramid level's residuals / Jacobians / mu, sigma, and the reduced system of the first
    linearisation (S, g: 1e-9 -- the step itself is not compared, S is singular along the scale gauge up to the damping).
    (2) The end state in the quantities the scale gauge cannot touch: rotation, translation DIRECTION, rho |t| and theta |t|."""
    P = synth.init_pair(seed=5)
    o = abi.options_init()
    for l in range(4):
        _check_eval(gpu, oracle_lib, P, o, l)
    ro = oracle_lib.reduced_system(P, o, o.levels[0], o.initial_radius)
    gpu.upload(P, o)
    rg = gpu.reduced_system(o.initial_radius)
    m = 6*ro["nf"]
    assert m == 6 and _rel(rg["S"][:m, :m], ro["S"]) < 1e-9 and _rel(rg["g"][:m], ro["g"]) < 1e-9
    assert rg["cost"] == pytest.approx(ro["cost"], rel=1e-12)
    G, R = P.copy(), P.copy()
    gpu.InitBA(G, options=o); oracle_lib.solve(R, o)

    def invariants(Q):
        pose = Q.pose.reshape(-1, 7)[1]; t = pose[4:]; s = np.linalg.norm(t)
        return pose[:4], t/s, Q.rho*s, Q.theta*s
    for a, b in zip(invariants(G), invariants(R)):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-6)


def test_landmarker_parity(gpu, oracle_lib):
    """optimizer::OptimizeLandmarker (rows R5 / R9): every pose constant, rho and theta refined, scene outlier pass."""
    P = synth.landmark_refine(seed=9)
    G, rep = _check_solve(gpu, oracle_lib, P, abi.options_landmarker(), lambda G, o: gpu.OptimizeLandmarker(G, options=o))
    assert np.array_equal(G.pose, P.pose) and not np.array_equal(G.rho, P.rho)


def test_theta_optim_parity_and_covariance(gpu, oracle_lib):
    """optimizer::ThetaOptimMultiFs (row R9): theta only, no loss, then the 3x3 covariance of the plane."""
    P = synth.landmark_refine(seed=3, n_pt=0, n_text=2)
    o = abi.options_theta()
    G, R = P.copy(), P.copy()
    rep_g, cov_g = gpu.ThetaOptimMultiFs(G, text=1, options=o)
    rc, rep_o, cov_o = oracle_lib.theta_optim(R, o, 1)
    assert rc == 0
    assert rep_g["iters"] == rep_o["iters"] and rep_g["termination"] == rep_o["termination"]
    np.testing.assert_allclose(G.theta, R.theta, rtol=0, atol=1e-8)
    np.testing.assert_allclose(cov_g, cov_o, rtol=1e-7)
    assert np.all(np.linalg.eigvalsh(cov_g) > 0)


@pytest.mark.gpu
def test_text_label_image_parity(gpu, oracle_lib):
    """tsba_text_label_image after a local BA vs the oracle's fillPoly restatement on the optimised parameters: identical images."""
    P = synth.tiny(seed=31, n_kf=5, n_pt=80, n_text=6, text_targets=4)
    o = abi.options_local()
    G = P.copy()
    gpu.LocalBundleAdjustment(G, options=o)                   # G now holds the optimised poses / planes, the context the same state
    s = G.struct()
    for lvl in (0, 2):
        shape = (int(s.img_h[lvl]), int(s.img_w[lvl]))
        for kf in (0, P.n_kf - 1):
            lab = gpu.TextLabelImage(kf, lvl, shape)
            ref = oracle_lib.label_image(G, kf, lvl)
            assert lab.shape == ref.shape
            nbad = int(np.count_nonzero(lab != ref))
            assert nbad == 0, "level %d kf %d: %d of %d pixels differ" % (lvl, kf, nbad, lab.size)
            assert (lab >= 0).any()


@pytest.mark.gpu
@pytest.mark.parametrize("n_kf,band,far", [(40, 6, 0.0), (70, 9, 0.0), (50, 5, 0.08)])
def test_global_ba_band_cholesky_profiles(gpu, oracle_lib, n_kf, band, far):
    """The profile-aware Cholesky on several envelopes: a narrow band over 3 / 5 block columns with a short last block, and a
    band plus long-range (loop-closure-like) observations, which widen the host's band bound towards the dense case."""
    P = synth.config_global(n_kf=n_kf, n_pt=50*n_kf, band=band, far_frac=far)
    o = abi.options_global(); o.its[0] = 6
    _check_solve(gpu, oracle_lib, P, o, lambda G, oo: gpu.GlobalBA(G, options=oo), atol=1e-7, rtol_cost=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("n_kf,n_pt,band", [(300, 15000, 8), (600, 20000, 12)])
def test_global_ba_partitioned_band_solver(gpu, n_kf, n_pt, band):
    """Substructured band solver (tsba_bandp.h): at 300 keyframes the host picks several interiors + separators.  The LM step of
    the first linearisation against a dense numpy solve of the same reduced system, and the full GlobalBA against the
    single-workgroup streaming solver (tsba_debug_set band_parts = 1): same LM trajectory, poses within 1e-9."""
    P = synth.config_global(n_kf=n_kf, n_pt=n_pt, band=band)          # band 12: border of 78 rows (two panel rounds, 3159 border-block tasks)
    o = abi.options_global(); o.its[0] = 6
    gpu.upload(P, o)
    rg = gpu.reduced_system(o.initial_radius)
    free = np.nonzero(rg["free"])[0]; idx = np.concatenate([np.arange(6*k, 6*k + 6) for k in free]); m = len(idx)
    S = rg["S"][:m, :m]; S = np.tril(S) + np.tril(S, -1).T
    assert np.abs(S).sum() > 0                                     # (the band paths leave S intact)
    ref = -np.linalg.solve(S, rg["g"][:m])
    assert np.abs(rg["dp"][idx] - ref).max() <= 1e-8*np.abs(ref).max()
    assert gpu.solver_info()["interiors"] > 1
    G1 = P.copy(); rep1 = gpu.GlobalBA(G1, options=o)
    gpu.debug_set(band_parts=1)
    try:
        G2 = P.copy(); rep2 = gpu.GlobalBA(G2, options=o)
        assert gpu.solver_info()["interiors"] == 1
    finally:
        gpu.debug_set()
    assert rep1["iters"] == rep2["iters"] and rep1["accepted"] == rep2["accepted"] and rep1["termination"] == rep2["termination"]
    assert rep1["accepted"][0] >= 3 and rep1["termination"][0] != 5
    np.testing.assert_allclose(G1.pose, G2.pose, rtol=0, atol=1e-9)
    np.testing.assert_allclose(rep1["cost1"], rep2["cost1"], rtol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("n_kf,state", [(4, abi.STATE_NOTREACHWIN), (9, abi.STATE_LOCAL), (10, abi.STATE_NOTREACHWIN), (11, abi.STATE_LOCAL),
                                        (20, abi.STATE_LOCAL), (21, abi.STATE_NOTREACHWIN), (22, abi.STATE_NOTREACHWIN), (27, abi.STATE_LOCAL),
                                        (31, abi.STATE_NOTREACHWIN)])
def test_small_system_solver_sizes(gpu, oracle_lib, n_kf, state):
    """The LDS solver of small reduced systems (k_solve_t: blocked 6x6 LDL^T, rhs as an extra row, back-substitution in one wave with
    60 rows per register) over 2 .. 29 free poses -- both sides of every 10-block boundary of the back-substitution and of the 16-row
    MFMA tile edges -- against a dense numpy solve of the downloaded system, and the whole LocalBundleAdjustment against the oracle."""
    P = synth.make_problem(n_kf=n_kf, n_pt=40*n_kf, n_text=3, seed=100 + n_kf, feats=(12, 8, 6), text_targets=3, max_targets=6, band=12)
    o = abi.options_local(state)
    gpu.upload(P, o)
    assert gpu.solver_info()["lds_solver"] == 1
    rg = gpu.reduced_system(o.initial_radius)
    idx = np.concatenate([np.arange(6*k, 6*k + 6) for k in np.nonzero(rg["free"])[0]]); m = len(idx)
    S = rg["S"][:m, :m]; S = np.tril(S) + np.tril(S, -1).T
    ref = -np.linalg.solve(S, rg["g"][:m])
    assert np.abs(rg["dp"][idx] - ref).max() <= 1e-9*np.abs(ref).max(), m
    assert np.all(rg["dp"][np.setdiff1d(np.arange(6*n_kf), idx)] == 0.0)
    _check_solve(gpu, oracle_lib, P, o, lambda G, oo: gpu.LocalBundleAdjustment(G, options=oo), atol=1e-7)


@pytest.mark.gpu
def test_theta_optim_singular_information_is_not_an_error(gpu):
    """A plane nobody observes usefully (sigma = 0 everywhere: its box projects outside the images) has a singular information matrix.
    The reference then keeps thetaVariance as it was and still returns true (optimizer.cc:2224-2241): TSBA_OK, cov untouched, cov_valid 0."""
    P = synth.landmark_refine(seed=3, n_pt=0, n_text=2)
    P.text_box_ray[1] += 50.0
    prev = np.arange(9, dtype=np.float64).reshape(3, 3)
    rep, cov = gpu.ThetaOptimMultiFs(P.copy(), text=1, options=abi.options_theta(), cov0=prev)
    assert rep["cov_valid"] == 0 and rep["status"] == 0 and np.array_equal(cov, prev)
    rep, cov = gpu.ThetaOptimMultiFs(P.copy(), text=0, options=abi.options_theta(), cov0=prev)
    assert rep["cov_valid"] == 1 and np.all(np.linalg.eigvalsh(cov) > 0)


@pytest.mark.parametrize("n_kf", [4, 5, 7, 20, 31])
def test_small_window_solver_schedules_agree(gpu, n_kf):
    """The reduced system of a small window through its schedules (tsba_debug_options.solve_variant): the production kernel (two panel waves:
    look-ahead, 6x6 LDL^T and panel solve on the same wave) and the round-4 experiment with a separate wave that factors the next diagonal block
    while the panel is solved (tsba_solve_la.h; measured slower, kept for A/B runs) -- same LM trajectory, first step to 1e-11.  1, 2, 4, 17 and
    28 free poses (the LOCAL gauge fixes three keyframes)."""
    P = synth.config_c4() if n_kf == 20 else synth.make_problem(n_kf=n_kf, n_pt=60*n_kf, n_text=max(2, n_kf//2), seed=40 + n_kf, feats=(16, 8, 6))
    o = abi.options_local()
    runs = []
    try:
        for var in (0, 1, 2, 3, 4):
            gpu.debug_set(solve_variant=var)
            gpu.upload(P, o)
            rs = gpu.reduced_system(o.initial_radius)
            rep = gpu.solve(); G = gpu.download(P.copy())
            runs.append((rs["dp"].copy(), rep, G))
    finally:
        gpu.debug_set()
    ref = runs[0]
    assert np.abs(ref[0]).max() > 0
    for dp, rep, G in runs[1:]:
        assert np.abs(dp - ref[0]).max() <= 1e-11*np.abs(ref[0]).max()
        assert rep["iters"] == ref[1]["iters"] and rep["accepted"] == ref[1]["accepted"] and rep["termination"] == ref[1]["termination"]
        np.testing.assert_allclose(rep["cost1"], ref[1]["cost1"], rtol=1e-10)
        np.testing.assert_allclose(G.pose, ref[2].pose, rtol=0, atol=1e-9)
    # variant 3 = the production solver with the back-substitution as a launch of its own; production (0) runs both in one launch (k_solve_back: the
    # back-substitution blocks poll the pose step): the same arithmetic in the same order, the same bits
    assert np.array_equal(runs[3][2].pose, ref[2].pose) and np.array_equal(runs[3][2].rho, ref[2].rho) and np.array_equal(runs[3][2].theta, ref[2].theta)
    assert runs[3][1]["cost1"] == ref[1]["cost1"]
    # variant 4 = variant 3 with the 6x6 diagonal blocks factored from an LDS scratch copy in every lane (production until round 5) instead of in
    # place across lanes 0..5 with v_readlane broadcasts: the same operations on the same operands
    assert np.array_equal(runs[4][0], runs[3][0])
    assert np.array_equal(runs[4][2].pose, ref[2].pose) and np.array_equal(runs[4][2].rho, ref[2].rho) and np.array_equal(runs[4][2].theta, ref[2].theta)
    assert runs[4][1]["cost1"] == ref[1]["cost1"]


def _full_state(gpu, rep, G):
    return (rep["iters"], rep["accepted"], rep["termination"], rep["cost0"], rep["cost1"], rep["n_sblock"], rep["n_tblock"],
            rep["n_bad_scene"], rep["n_bad_tfeat"], rep["n_bad_text"])


@pytest.mark.parametrize("knob", ["pass_launches", "trial_launches"])
@pytest.mark.parametrize("case", ["tiny", "c4", "c4_oneshot", "init", "landmarker", "no_text", "no_outlier", "pose_c3", "pose_oneshot", "pose_no_text"])
def test_pass_boundaries_in_one_launch_agree(gpu, case, knob):
    """A window's pass begins with k_pass_begin (participation + gauge + LM state reset + mu / sigma) and ends with k_pass_end (outlier pass + the next
    level's mu / sigma + clearing), its final state reaches the report through k_solve_end (tsba_kernels_pass.h); tsba_debug_options.pass_launches = 1
    keeps the launches of rounds 1-4 (k_pass_reset, k_participation, k_gauge_wave, k_musigma | k_outlier, state copies).  Same arithmetic: the
    reports (every per-pass field), the parameters, the flags and the LM traces are bit-identical -- resident solves, solves repeated on one upload
    and one-shot calls (levels staged while the first pass runs) alike.
    knob = trial_launches: the same statement for the round-5 experiment k_lin_mid (k_mid inside the speculative linearisation's launch: the last workgroups of the
    linearisation to finish take the k_mid blocks; measured slower and off by default) against k_mid as a launch of its own with the same block size.
    pose_*: PoseOptim -- all LM steps of a pass in one launch (k_pose_pass: a barrier across its workgroups where the launches were) against a launch per
    step (k_pose_iter, pass_launches = 1)."""
    oneshot = None
    if case.startswith("pose") and knob != "pass_launches":
        pytest.skip("the pose-only path has no k_mid")
    if case == "tiny":
        P, o = synth.tiny(), abi.options_local()
    elif case in ("c4", "c4_oneshot"):
        P, o = synth.config_c4(), abi.options_local()
        if case == "c4_oneshot":
            oneshot = lambda Q: gpu.LocalBundleAdjustment(Q, options=o)
    elif case == "init":
        P, o = synth.init_pair(), abi.options_init()
    elif case in ("pose_c3", "pose_oneshot"):
        P, o = synth.config_c3(), abi.options_pose()
        if case == "pose_oneshot":
            oneshot = lambda Q: gpu.PoseOptim(Q, options=o)
    elif case == "pose_no_text":
        P, o = synth.config_c3(), abi.options_pose()
        o.use_text = 0
    elif case == "landmarker":
        P, o = synth.landmark_refine(), abi.options_landmarker()
    elif case == "no_text":
        P, o = synth.make_problem(n_kf=9, n_pt=700, n_text=0, seed=91, feats=(16, 8, 6)), abi.options_local()
        o.n_passes = 2; o.levels[0] = 0; o.levels[1] = 0          # (a scene-only synthetic problem has one pyramid level: two passes on it)
    else:
        P, o = synth.make_problem(n_kf=12, n_pt=900, n_text=6, seed=92, feats=(16, 8, 6)), abi.options_local()
        o.outlier_scene = o.outlier_text = 0
    runs = []
    try:
        for old in (0, 1, 0):
            gpu.debug_set(**{knob: (old if knob == "pass_launches" else 2 - old)})       # (trial_launches: 2 = the k_lin_mid experiment, 1 = its two-launch partner)
            for rep_no in range(2):                              # twice on one upload: tsba_solve restarts from the uploaded parameters
                G = P.copy()
                if oneshot:
                    rep = oneshot(G)
                else:
                    if rep_no == 0:
                        gpu.upload(P, o)
                    rep = gpu.solve(); gpu.download(G)
                traces = [gpu.lm_trace(ps) for ps in range(o.n_passes)]
                runs.append((_full_state(gpu, rep, G), G, traces, rep["poll_timeouts"]))
    finally:
        gpu.debug_set()
    ref = runs[2]                                                # (the launches of rounds 1-4)
    assert sum(ref[0][0]) > 0 and all(r[3] == 0 for r in runs)   # (no poll ran into its bound)
    for st, G, tr, _ in runs:
        if case.startswith("pose"):
            assert runs[0][0] == runs[1][0] == runs[4][0] == runs[5][0] and np.array_equal(runs[0][1].pose, runs[5][1].pose)      # (the one-launch path: the same bits every time)
            # two kernels, one arithmetic -- but not one compilation of it: the compiler contracts the sweep's multiply-adds differently inside k_pose_pass's loop
            # and inside k_pose_iter, sums differ in the last bit (observed 2e-16 relative).  Decisions, counts and flags identical; costs and the pose to 1e-12
            assert st[:3] == ref[0][:3] and st[5:] == ref[0][5:], (st, ref[0])
            np.testing.assert_allclose(st[3], ref[0][3], rtol=1e-12); np.testing.assert_allclose(st[4], ref[0][4], rtol=1e-12)
            np.testing.assert_allclose(G.pose, ref[1].pose, rtol=0, atol=1e-12)
            assert np.array_equal(G.rho, ref[1].rho) and np.array_equal(G.theta, ref[1].theta)
            assert np.array_equal(G.sgood, ref[1].sgood) and np.array_equal(G.tobs_good, ref[1].tobs_good) and np.array_equal(G.tfgood, ref[1].tfgood)
            continue
        assert st == ref[0], (st, ref[0])
        assert np.array_equal(G.pose, ref[1].pose) and np.array_equal(G.rho, ref[1].rho) and np.array_equal(G.theta, ref[1].theta)
        assert np.array_equal(G.sgood, ref[1].sgood) and np.array_equal(G.tobs_good, ref[1].tobs_good) and np.array_equal(G.tfgood, ref[1].tfgood)
        for a, b in zip(tr, ref[2]):
            assert np.array_equal(a, b, equal_nan=True)
