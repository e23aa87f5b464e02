"""Full-size global-BA configurations against the CPU ORACLE (not against the GPU's own downloaded system): BASELINE config 4 (C5: 500 KF x
50 k points x 1000 text planes) and every 5000-keyframe variant of config 5 (C6: open chain, SURVEY 8d's 1 % long-range points, the ring
right after a loop closure, two separate closures).  Reference: PyrGlobalBA, /root/reference/src/optimizer.cc:1701-1851 (Ceres LM over
SPARSE_NORMAL_CHOLESKY, :1833-1840).

Per configuration:
  * first linearisation: cost0 (rtol 1e-11), the reduced gradient g and every 6x6 block of the reduced camera system S -- band, ghost rows and
    the blocks outside the band alike, compared by keyframe pair -- against the oracle's block-sparse restatement (rel 1e-9);
  * first LM step: the pose step of the GPU solves the ORACLE's system (direct sparse solve where the factor stays sparse, residual of the
    oracle's system on the map with long-range points);
  * the LM prefix: the oracle's trust-region loop on block-sparse normal equations with an exact linear solve (oracle.sparse_solver) against
    the GPU's trace, trial by trial: K = number of leading trials with the same decision and the same candidate cost.  On maps of this size
    the trajectory is NOT a function of the algorithm alone: two exact solvers inside the oracle itself (band Cholesky / SuperLU) part at
    K(1e-9) = 4, K(1e-6) = 6 on the open chain (docs/ledger_r04.md 13.1) -- rounding differences of 1e-16 in the step grow by ~30 x per LM iteration.  The test
    asserts a floor for K and records the measured values in gpurun_out/lm_prefix.json.
  * N = 2 / 8 in-process ranks at 5000 keyframes: the same prefix statement for the sharded solve against the single-rank GPU trace.
"""
import json
import os
import numpy as np
import pytest

from textslam_amd import abi
from test_gpu_global import _on_ranks

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONFIGS = {
    "c5_text":       dict(n_kf=500, n_pt=50000, n_text=1000, seed=7, feats=(64, 24, 12), band=12, n_levels=1),
    "c6_open_chain": dict(n_kf=5000, n_pt=70000, band=10),
    "c6_long_range": dict(n_kf=5000, n_pt=70000, band=10, far_frac=0.01),
    "c6_ring":       dict(n_kf=5000, n_pt=70000, band=10, loop=True),
    "c6_closures2":  dict(n_kf=5000, n_pt=70000, band=10, closures=2),
}
PREFIX_ITS = 12          # LM trials compared (the trajectories of two exact implementations part well before)
# floors for (K at 1e-9, K at 1e-6, trials with the same decision), set below the measured values (docs/ledger_r04.md 13.1): C5 12 / 12 / 12, ring 11 / 11 / 11 and two
# closures 10 / 10 / 10 (the whole run, costs to 1e-13: closed loops pin the drift modes), open chain 1 / 5 / 12 (its free end makes S ill-conditioned:
# two backward-stable solvers differ by cond x eps in the step), long-range points 0 / 3 / 12 with the product's 1e-10 conjugate-gradient tolerance
# Round 5: every floor is (measured in rounds 4 and 5) - 1, so that a regression of two trials fails; the 1e-9 floor of the open chain stays at its
# measured 1 (the first trial has agreed to 6e-11 on every box so far) and that of the long-range map at 0 -- there the first step is checked directly
# instead (its residual in the oracle's system <= 1e-9, no unconverged solve, first trial <= 1e-8; with the iteration run to 1e-13: <= 1e-9).
K_FLOOR = {"c5_text": (11, 11, 11), "c6_open_chain": (1, 4, 11), "c6_long_range": (0, 2, 11), "c6_ring": (10, 10, 10), "c6_closures2": (9, 9, 9)}


@pytest.fixture(scope="module")
def gpu():
    from textslam_amd.optimizer import Optimizer
    g = Optimizer(0)
    yield g
    g.close()


def _options(name):
    o = abi.options_global()
    if name == "c5_text":
        o.use_text = 1
    return o


def prefix_length(tr_a, tr_b, rtol):
    """Leading LM trials with the same decision and candidate costs within rtol."""
    K = 0
    for a, b in zip(tr_a, tr_b):
        if a[3] != b[3]:
            break
        if np.isnan(a[0]) != np.isnan(b[0]) or (not np.isnan(a[0]) and not abs(a[0] - b[0]) <= rtol*abs(b[0])):
            break
        K += 1
    return K


def _record(name, **kw):
    path = os.path.join(ROOT, "gpurun_out", "lm_prefix.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = json.load(open(path)) if os.path.exists(path) else {}
    data.setdefault(name, {}).update(kw)
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)


def compare_first_linearisation(gpu, oracle_lib, P, o, direct=True, measure_only=False):
    """cost, reduced gradient, every 6x6 block of S by keyframe pair and the first pose step of the uploaded problem against the oracle's block-sparse
    restatement.  -> (oracle blocks, worst block difference relative to the block's scale, relative residual of the oracle's system at the GPU's step)"""
    nk = P.n_kf
    ob = oracle_lib.reduced_blocks(P, o, 0, o.initial_radius)
    kf_of = np.nonzero(ob["free_idx"] >= 0)[0]                       # free-pose block -> keyframe (keyframe order)
    n = 6*ob["nf"]
    gb = gpu.reduced_blocks(o.initial_radius)
    assert abs(gb["cost"] - ob["cost"]) <= 1e-11*ob["cost"], (gb["cost"], ob["cost"])
    g_or = np.zeros(6*nk)
    for q, k in enumerate(kf_of):
        g_or[6*k:6*k + 6] = ob["g"][6*q:6*q + 6]
    gscale = np.abs(g_or).max()
    g_gap = float(np.abs(gb["g"] - g_or).max()/gscale)
    if measure_only:                                                  # (a state where the reduction is ill-conditioned: the caller asserts what the measurements support)
        sscale = np.abs(ob["val"]).max(); worst = 0.0
        for q in range(len(ob["br"])):
            blk = gb["blocks"].get((int(kf_of[ob["br"][q]]), int(kf_of[ob["bc"][q]])))
            if blk is not None:
                worst = max(worst, float(np.abs(blk - ob["val"][q]).max()/max(np.abs(ob["val"][q]).max(), 1e-6*sscale)))
        A = oracle_lib.blocks_to_sparse(n, ob["br"], ob["bc"], ob["val"])
        dp = np.concatenate([gb["dp"][6*k:6*k + 6] for k in kf_of])
        return ob, worst, float(np.abs(A @ dp + ob["g"]).max()/np.abs(ob["g"]).max()), g_gap
    assert g_gap <= 1e-9, g_gap
    # every block of S, by keyframe pair (the oracle's blocks are in keyframe order: rows = the later keyframe)
    sscale = np.abs(ob["val"]).max()
    seen, worst = set(), 0.0
    for q in range(len(ob["br"])):
        key = (int(kf_of[ob["br"][q]]), int(kf_of[ob["bc"][q]]))
        blk = gb["blocks"].get(key)
        ref = ob["val"][q]
        if blk is None:
            assert np.abs(ref).max() <= 1e-9*sscale, (key, np.abs(ref).max())     # (structurally present in the oracle, numerically nothing)
            continue
        seen.add(key)
        bscale = max(np.abs(ref).max(), 1e-6*sscale)
        worst = max(worst, float(np.abs(blk - ref).max()/bscale))
    assert worst <= 1e-9, worst
    for key, blk in gb["blocks"].items():                             # nothing on the GPU that the oracle does not have
        assert key in seen or np.abs(blk).max() <= 1e-9*sscale, key
    # first LM step: the GPU's pose step solves the oracle's system
    A = oracle_lib.blocks_to_sparse(n, ob["br"], ob["bc"], ob["val"])
    dp = np.concatenate([gb["dp"][6*k:6*k + 6] for k in kf_of])
    res = A @ dp + ob["g"]
    rres = float(np.abs(res).max()/np.abs(ob["g"]).max())
    assert rres <= 1e-8, rres
    if direct:                                                        # (where the sparse factor stays sparse; else the residual above is the statement)
        ref = oracle_lib.sparse_solver(A.tocsc(), -ob["g"])
        assert np.abs(dp - ref).max() <= 1e-8*np.abs(ref).max(), np.abs(dp - ref).max()/np.abs(ref).max()
    return ob, worst, rres


@pytest.mark.parametrize("kw,dbg,expect", [
    (dict(n_kf=600, n_pt=12000, band=8), dict(), dict(interiors_min=2)),
    (dict(n_kf=900, n_pt=18000, band=9, loop=True), dict(), dict(ring=1)),
    (dict(n_kf=1500, n_pt=30000, band=7, loop=True, loop_at=300), dict(), dict(ring=1)),
    (dict(n_kf=900, n_pt=18000, band=9, loop=True), dict(no_ring=1), dict(kf_reordered=1)),
    (dict(n_kf=700, n_pt=14000, band=8, far_frac=0.02), dict(far_solver=2), dict(far=1)),
    (dict(n_kf=900, n_pt=18000, band=8, closures=2), dict(far_solver=2), dict(far=1)),
])
def test_every_storage_of_the_reduced_system_against_the_oracle_blocks(gpu, oracle_lib, kw, dbg, expect):
    """tsba_debug_reduced_blocks over every storage the library keeps S in -- band rows of the partitioned solver, ghost rows of a ring (with and
    without a tail), rows in reverse Cuthill-McKee order, band + blocks outside it (scattered long-range points; two closures) -- block by block
    against the oracle at sizes where that takes a second.  (The wide-band Cholesky factors S in place: nothing to export after its solve.)"""
    P = synth_global(**kw); o = abi.options_global()
    try:
        gpu.debug_set(**dbg)
        gpu.upload(P, o)
        info = gpu.solver_info()
        assert info["band_storage"] == 1, info
        if expect.get("ring"):
            assert info["ring"] == 1, info
        if expect.get("kf_reordered"):
            assert info["kf_reordered"] == 1 and info["ring"] == 0, info
        if expect.get("far"):
            assert info["far_blocks"] > 0 and info["far_band_blocks"] > 0, info
        if expect.get("interiors_min"):
            assert info["interiors"] >= expect["interiors_min"], info
        compare_first_linearisation(gpu, oracle_lib, P, o)
    finally:
        gpu.debug_set()


def synth_global(**kw):
    from textslam_amd import synth
    return synth.config_global(**kw)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_first_linearisation_and_lm_prefix_against_the_oracle(gpu, oracle_lib, map_cache, name):
    P = map_cache(**CONFIGS[name]); o = _options(name)
    gpu.upload(P, o)
    info = gpu.solver_info()
    assert info["band_storage"] == 1, info
    if name == "c6_long_range":
        assert info["far_blocks"] > 5000 and info["far_band_blocks"] > 0, info
    if name == "c6_closures2":
        assert info["far_blocks"] > 0, info
    if name == "c6_ring":
        assert info["ring"] == 1, info
    ob, worst, rres = compare_first_linearisation(gpu, oracle_lib, P, o, direct=name != "c6_long_range")
    _record(name, n_blocks=len(ob["br"]), worst_block_rel=worst, cost0=ob["cost"], first_step_residual=rres)
    # ---- the LM prefix
    o.its[0] = PREFIX_ITS
    gpu.upload(P, o); rep_g = gpu.solve(); tr_g = gpu.lm_trace(0)
    st = gpu.pcg_stats()
    assert st["hit_cap"] == 0, st
    R = P.copy()
    if name in ("c5_text", "c6_open_chain"):                          # a band in keyframe order: the oracle's own band Cholesky
        rep_o, tr_o = oracle_lib.solve_traced(R, o)
    else:
        oracle_lib.SOLVER_LOG.clear()
        oracle_lib.set_sparse_solver(oracle_lib.sparse_solver)
        try:
            rep_o, tr_o = oracle_lib.solve_traced(R, o)
        finally:
            oracle_lib.set_sparse_solver(None)
        assert oracle_lib.SOLVER_LOG and oracle_lib.SOLVER_LOG[0][0] == ("gmres" if name == "c6_long_range" else "banded"), oracle_lib.SOLVER_LOG[:2]
    tr_o = tr_o[0]
    assert abs(rep_g["cost0"][0] - rep_o["cost0"][0]) <= 1e-11*rep_o["cost0"][0]
    K9, K6, K3 = (prefix_length(tr_g, tr_o, t) for t in (1e-9, 1e-6, 1e-3))
    Kdec = 0
    for a, b in zip(tr_g, tr_o):                                      # decisions only
        if a[3] != b[3]:
            break
        Kdec += 1
    rel = [float(abs(a[0] - b[0])/abs(b[0])) if not (np.isnan(a[0]) or np.isnan(b[0])) else None for a, b in zip(tr_g, tr_o)]
    print(f"\n{name}: LM prefix K(1e-9) = {K9}, K(1e-6) = {K6}, K(1e-3) = {K3}, decisions agree for {Kdec} of {min(len(tr_g), len(tr_o))} trials; "
          f"final cost GPU {rep_g['cost1'][0]:.6g} / oracle {rep_o['cost1'][0]:.6g}")
    _record(name, K_1e9=K9, K_1e6=K6, K_1e3=K3, K_decisions=Kdec, trials=int(min(len(tr_g), len(tr_o))), rel_cost_by_trial=rel,
            cost1_gpu=rep_g["cost1"][0], cost1_oracle=rep_o["cost1"][0], iters_gpu=rep_g["iters"][0], iters_oracle=rep_o["iters"][0],
            accepted_gpu=rep_g["accepted"][0], accepted_oracle=rep_o["accepted"][0])
    f9, f6, fd = K_FLOOR[name]
    assert K9 >= f9 and K6 >= f6 and Kdec >= fd, (K9, K6, Kdec, rel)
    assert rep_g["poll_timeouts"] == 0 and rep_g["pcg_unconverged"] == 0, rep_g
    if name == "c6_long_range":
        assert rres <= 1e-9, rres                                 # the first step solves the ORACLE's system (measured 2e-12)
    assert tr_g[0][3] == tr_o[0][3] and abs(tr_g[0][0] - tr_o[0][0]) <= (1e-8 if name == "c6_long_range" else 1e-9)*tr_o[0][0]     # the first trial: cost at x0 + dx (measured: 6e-11 on the open chain, <= 1e-12 elsewhere, 5e-9 with the 1e-10 iterative solve)
    if name == "c6_long_range":
        # the prefix is limited by the inexact linear solve, not by the assembly: with the conjugate gradients run to 1e-13 the first trial agrees
        # with the oracle as on the open chain
        try:
            gpu.debug_set(pcg_tol_exp=13, pcg_max_it=400)
            gpu.upload(P, o); rep_t = gpu.solve(); tr_t = gpu.lm_trace(0)
            assert gpu.pcg_stats()["hit_cap"] == 0
        finally:
            gpu.debug_set()
        K9t, K6t = prefix_length(tr_t, tr_o, 1e-9), prefix_length(tr_t, tr_o, 1e-6)
        rel_t = [float(abs(a[0] - b[0])/abs(b[0])) if not (np.isnan(a[0]) or np.isnan(b[0])) else None for a, b in zip(tr_t, tr_o)]
        print(f"{name}, conjugate gradients to 1e-13: K(1e-9) = {K9t}, K(1e-6) = {K6t}; first trial {rel_t[0]:.1e}")
        _record(name, K_1e9_tight=K9t, K_1e6_tight=K6t, rel_cost_by_trial_tight=rel_t)
        assert rel_t[0] <= 1e-9 and K6t >= 2, (K9t, K6t, rel_t)
    # both end far below the start (the two trajectories are both valid LM runs)
    assert rep_g["cost1"][0] < 0.2*rep_g["cost0"][0] and abs(rep_g["cost1"][0] - rep_o["cost1"][0]) <= 0.1*rep_o["cost1"][0]


@pytest.mark.parametrize("name", ["c6_open_chain", "c6_long_range"])
@pytest.mark.parametrize("world", [2, 8])
def test_sharded_lm_prefix_at_5000_keyframes(gpu, map_cache, name, world):
    """N in-process ranks (the product's own N > 1 path: sharded upload, split kernels, every collective) against the single-rank GPU trace
    on the 5000-keyframe maps: first trial to 1e-10, then the prefix length."""
    P = map_cache(**CONFIGS[name]); o = _options(name); o.its[0] = PREFIX_ITS
    gpu.upload(P, o); rep1 = gpu.solve(); tr1 = gpu.lm_trace(0)

    def solve(g, rank):
        g.upload(P, o); rep = g.solve()
        return rep, g.lm_trace(0), g.solver_info()
    outs = _on_ranks(world, solve)
    for rep, tr, info in outs:
        assert info["world"] == world
        assert np.array_equal(tr, outs[0][1], equal_nan=True)           # every rank takes the same decisions on the same sums
    tr = outs[0][1]
    K9, K6 = prefix_length(tr, tr1, 1e-9), prefix_length(tr, tr1, 1e-6)
    print(f"\n{name}, {world} ranks against 1: LM prefix K(1e-9) = {K9}, K(1e-6) = {K6} of {len(tr1)} trials; final cost {outs[0][0]['cost1'][0]:.6g} / {rep1['cost1'][0]:.6g}")
    _record(f"{name}_ranks{world}", K_1e9=K9, K_1e6=K6, trials=int(len(tr1)), cost1=outs[0][0]["cost1"][0], cost1_single=rep1["cost1"][0])
    assert abs(outs[0][0]["cost0"][0] - rep1["cost0"][0]) <= 1e-12*rep1["cost0"][0]
    assert tr[0][3] == tr1[0][3] and abs(tr[0][0] - tr1[0][0]) <= 1e-9*tr1[0][0]
    assert K9 >= 1 and K6 >= 2, (K9, K6)             # measured: 1-2 / 3-7 (the sharded sums differ from the unsharded ones in the last bits)


CONVERGED_ITS = 600


def _gauge_aligned_pose_gap(Pa, Pb):
    """Largest camera-centre distance between two solutions of one map after the best rigid + scale alignment of the centres (a global BA without
    fixed scale leaves a similarity free along nearly flat directions; the first keyframe is held, the rest can drift together)."""
    def centres(pose):
        q, t = pose[:, :4]/np.linalg.norm(pose[:, :4], axis=1, keepdims=True), pose[:, 4:]
        w, x, y, z = q.T
        R = np.stack([1 - 2*(y*y + z*z), 2*(x*y - w*z), 2*(x*z + w*y), 2*(x*y + w*z), 1 - 2*(x*x + z*z), 2*(y*z - w*x),
                      2*(x*z - w*y), 2*(y*z + w*x), 1 - 2*(x*x + y*y)], axis=1).reshape(-1, 3, 3)
        return -np.einsum("nji,nj->ni", R, t)                       # c = -R^T t  (T_cw)
    A, B = centres(Pa), centres(Pb)
    ma, mb = A.mean(0), B.mean(0)
    H = (A - ma).T @ (B - mb)
    U, S, Vt = np.linalg.svd(H)
    D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
    Rm = Vt.T @ D @ U.T
    sc = np.trace(np.diag(S) @ D)/((A - ma)**2).sum()
    Aal = sc*(A - ma) @ Rm.T + mb
    extent = np.linalg.norm(B - mb, axis=1).max()
    return float(np.linalg.norm(Aal - B, axis=1).max()/extent), float(np.linalg.norm(A - B, axis=1).max()/extent)


@pytest.mark.parametrize("name", ["c6_open_chain", "c6_long_range"])
def test_converged_answers_against_the_oracle(gpu, oracle_lib, map_cache, name):
    """SURVEY 8d's tolerance is stated on CONVERGED answers, and the reference runs one deterministic solve to Ceres' exit (optimizer.cc:1833-1846).  The
    12-trial prefix above cannot say whether GPU and oracle arrive at the same answer on the two maps where their trajectories part (cond x eps, docs/ledger_r04.md
    13.1): here both run until the function tolerance ends them (oracle: 231 / 129 iterations, tests/golden/make_converged.py; end states committed).

      (a) AT the oracle's converged answer the two implementations agree on everything the next iteration is made of: cost (1e-11), reduced gradient and
          every 6x6 block of S (1e-9), and the GPU's pose step solves the oracle's system -- the full first-linearisation comparison, at the END state;
      (b) both runs from the common start end by the same criterion; the final costs are compared, recorded, and asserted against the sharpness of that
          criterion on this map.  A function-tolerance exit is taken where ONE step improves the cost by less than 1e-6: that is not a distance to a minimum.
          The oracle itself, started again at its own answer, runs on for dozens of iterations and lowers the cost by `again` (4.5e-4 on the open chain,
          fixture) before the tolerance ends it a second time: the reference's own exit defines the converged cost of these maps to no better than that.
          The two converged costs are recorded with the camera centres' gap after similarity alignment and asserted at twice what was measured;
      (c) the GPU started at the oracle's answer against the oracle started there (same state, same initial trust region): the LM prefix again."""
    fx = np.load(os.path.join(ROOT, "tests", "golden", f"converged_{name}.npz"))
    assert int(fx["term"]) == 1 and int(fx["again_term"]) == 1
    again = (float(fx["cost1"]) - float(fx["again_cost1"]))/float(fx["cost1"])
    P = map_cache(**CONFIGS[name]); o = _options(name); o.its[0] = CONVERGED_ITS
    # (b) from the common start
    gpu.upload(P, o); rep_g = gpu.solve(); G = gpu.download(P.copy())
    assert rep_g["poll_timeouts"] == 0 and rep_g["pcg_unconverged"] == 0, rep_g
    assert abs(rep_g["cost0"][0] - float(fx["cost0"])) <= 1e-11*float(fx["cost0"])
    assert rep_g["termination"][0] == 1, rep_g                       # the function tolerance, as the oracle
    rel_cost = abs(rep_g["cost1"][0] - float(fx["cost1"]))/float(fx["cost1"])
    gap_al, gap_raw = _gauge_aligned_pose_gap(G.pose, fx["pose"])
    # (a) at the oracle's end state: every residual and every Jacobian entry (tsba_eval, the reference's block order), the cost; the REDUCED system is
    # measured and recorded -- after 231 iterations some landmarks are barely constrained (V_j near the LM floor), and W V^-1 b amplifies the last bits
    # of V_j by 1/V_j: the reduced gradient agrees to ~1e-4 of its scale where it agreed to 1e-9 at the start (the amplification that parts the trajectories)
    Q = P.copy(); Q.pose[:] = fx["pose"]; Q.rho[:] = fx["rho"]
    o1 = _options(name)
    eo, eg = oracle_lib.evaluate(Q, o1, 0), gpu.evaluate(Q, o1, 0)
    assert (eo["ns"], eo["nt"]) == (eg["ns"], eg["nt"]) and eo["ns"] > 400000
    # Block by block, relative to the block's own magnitude.  After 231 iterations the 5 % outlier observations have pushed a handful of points to where one of
    # their cameras sees them at almost zero depth AND almost on the optical axis: Jacobian entries of 1e9 .. 1e12 on a residual of 1e-5 .. 1e3 pixels (open
    # chain: one block at 1.2e12, its residual moves by 8e-6 when the translations are scaled by 1 + 2e-16).  Those blocks -- max |J| > 1e6, under 1 % of all, counted and
    # recorded -- are compared to 1e-3; every other block to 1e-9.
    ro, rg = eo["resid"].reshape(-1, 2), eg["resid"].reshape(-1, 2)
    Jo, Jg = eo["jac_scene"].reshape(eo["ns"], -1), eg["jac_scene"].reshape(eo["ns"], -1)
    jmax = np.abs(Jo).max(1); ill = jmax > 1e6
    rgap = np.abs(rg - ro).max(1)/np.maximum(1.0, np.abs(ro).max(1)); jgap = np.abs(Jg - Jo).max(1)/np.maximum(1.0, jmax)
    resid_gap, jac_gap = float(rgap[~ill].max()), float(jgap[~ill].max())
    ill_resid_gap, ill_jac_gap = (float(rgap[ill].max()), float(jgap[ill].max())) if ill.any() else (0.0, 0.0)
    print(f"\n{name}: at the oracle's answer {int(ill.sum())} of {len(ill)} blocks have Jacobian entries above 1e6 (largest {jmax.max():.3g}): residuals {ill_resid_gap:.1e}, Jacobians {ill_jac_gap:.1e}; "
          f"all other blocks: residuals {resid_gap:.1e}, Jacobians {jac_gap:.1e}")
    assert resid_gap <= 1e-9 and jac_gap <= 1e-9, (resid_gap, jac_gap)
    assert int(ill.sum()) <= len(ill)//50 and ill_resid_gap <= 1e-3 and ill_jac_gap <= 1e-3, (int(ill.sum()), ill_resid_gap, ill_jac_gap)      # (measured: 3967 / 2397 of ~496 k blocks, 3.8e-4 / 5.9e-7)
    gpu.upload(Q, o1)
    ob, worst, rres, g_gap = compare_first_linearisation(gpu, oracle_lib, Q, o1, direct=False, measure_only=True)
    assert abs(ob["cost"] - float(fx["cost1"])) <= 1e-12*float(fx["cost1"])
    assert g_gap <= 1e-2 and worst <= 1e-2 and rres <= 1e-2, (g_gap, worst, rres)      # (measured: 1.1e-4, 4.5e-7, 1.4e-4 on the open chain)
    # (c) both started again there
    o2 = _options(name); o2.its[0] = 200
    gpu.upload(Q, o2); rep_a = gpu.solve(); tr_a = gpu.lm_trace(0)
    tr_o = fx["again_trace"]
    K9, K6 = prefix_length(tr_a, tr_o, 1e-9), prefix_length(tr_a, tr_o, 1e-6)
    Kdec = 0
    for a, b in zip(tr_a, tr_o):
        if a[3] != b[3]:
            break
        Kdec += 1
    moved = (rep_a["cost0"][0] - rep_a["cost1"][0])/rep_a["cost0"][0]
    print(f"\n{name}: converged GPU {rep_g['cost1'][0]:.9g} in {rep_g['iters'][0]} iterations / oracle {float(fx['cost1']):.9g} in {int(fx['iters'])}: rel {rel_cost:.2e} "
          f"(the oracle started again at its answer: {int(fx['again_iters'])} iterations, cost lower by {again:.2e}); camera centres {gap_al:.2e} of the map's extent after "
          f"similarity alignment ({gap_raw:.2e} before); at the oracle's answer: residuals {resid_gap:.1e}, Jacobians {jac_gap:.1e}, reduced gradient {g_gap:.1e}, S blocks {worst:.1e}, step residual {rres:.1e}; started again there: GPU {rep_a['iters'][0]} "
          f"iterations, cost lower by {moved:.2e}, prefix K(1e-9) = {K9}, K(1e-6) = {K6}, decisions {Kdec} of {min(len(tr_a), len(tr_o))}")
    _record(name, converged=dict(cost1_gpu=rep_g["cost1"][0], cost1_oracle=float(fx["cost1"]), rel_cost=rel_cost, iters_gpu=rep_g["iters"][0], iters_oracle=int(fx["iters"]),
                                 accepted_gpu=rep_g["accepted"][0], accepted_oracle=int(fx["accepted"]), oracle_again_rel=again, oracle_again_iters=int(fx["again_iters"]),
                                 centre_gap_aligned=gap_al, centre_gap_raw=gap_raw,
                                 at_oracle_answer=dict(residual_gap=resid_gap, jacobian_gap_rel=jac_gap, ill_conditioned_blocks=int(ill.sum()), ill_residual_gap=ill_resid_gap, ill_jacobian_gap_rel=ill_jac_gap, reduced_gradient_gap_rel=g_gap, worst_block_rel=worst, first_step_residual=rres, gpu_again_iters=rep_a["iters"][0], gpu_again_rel=moved,
                                                       K_1e9=K9, K_1e6=K6, K_decisions=Kdec, trials=int(min(len(tr_a), len(tr_o))))))
    fl = CONVERGED_FLOOR[name]
    checks = dict(again_terminates=rep_a["termination"][0] == 1 and rep_a["poll_timeouts"] == 0, again_same_length=rep_a["iters"][0] == int(fx["again_iters"]),
                  again_same_decrease=abs(moved - again) <= 1e-4*again, prefix_1e9=K9 >= fl["K9"], prefix_1e6=K6 >= fl["K6"], decisions=Kdec >= fl["Kdec"],
                  converged_cost=rel_cost <= fl["rel_cost"])
    assert all(checks.values()), (checks, rel_cost, again, moved, K9, K6, Kdec)


# Measured in round 5 (gpurun_out/lm_prefix.json -> profiles/r05_lm_prefix_vs_oracle.json), floors = measured - a few trials / measured x 2:
#   started AGAIN at the oracle's converged answer the two implementations run the same LM: open chain 30 of 30 trials to 1e-9 (both stop after 30 iterations
#   having lowered the cost by 4.5408e-4), long-range map 64 of 64 recorded decisions, 64 trials to 1e-6 (K(1e-9) = 0: the 1e-10 conjugate-gradient tolerance),
#   both stop after 96 iterations at -4.1002e-3;
#   from the COMMON START both end on the function tolerance, at costs 1.5e-3 (open chain: 154 against 231 iterations) and 2.9e-2 (long-range: 107 against 129)
#   apart -- NOT SURVEY 8d's 1e-6: the trajectories part after 1 - 5 trials (cond x eps, above) and the exit is taken wherever one step gains < 1e-6, which on
#   these valleys is path-dependent (the oracle itself moves on by 4.5e-4 / 4.1e-3 when started again at its answer).
# Round 6: k_mid sums a pair's text groups and a plane's slots on several lanes and its block partials in one reduction -- other summation orders, 1e-16 in the cost.
# On the long-range map that moves K(1e-6) from 64 to 45 of 64 trials (decisions still 64 of 64, the same 96 iterations, the same decrease to 1e-4): the floor is
# 40.  What these two maps can and cannot say is in docs/ledger_r06.md 15.1: two exact CPU solvers inside the oracle end 6.7e-3 apart on the open chain.
CONVERGED_FLOOR = {"c6_open_chain": dict(K9=26, K6=27, Kdec=28, rel_cost=4e-3), "c6_long_range": dict(K9=0, K6=40, Kdec=60, rel_cost=6e-2)}



# ---------------------------------------------------------------------------------------------------------------------------------------------------------------
# Round 6: SURVEY 8d's "converged parameters rel 1e-6" on the 5000-keyframe open chain and the map with long-range points AS THE REFERENCE HANDS THEM TO GlobalBA
# (tests/golden/make_converged.py): no point that a local BA has flagged (map::GetAllMapPoints(false), src/optimizer.cc:337-341, src/map.cc:36-47,
# src/tracking.cc:2215-2230) and every camera near its place (GlobalBA runs after the loop correction, loopClosing.cc:587-591).  The gauge is fixed by the two
# constant keyframes (optimizer.cc:405-406), so the parameters are compared directly -- no similarity alignment.
HANDED_OVER = {
    "c6_open_chain_handed_over": dict(n_kf=5000, n_pt=105000, band=10, drop_outlier_points=True, perturb_in_camera=True),
    "c6_long_range_handed_over": dict(n_kf=2000, n_pt=42000, band=10, far_frac=0.01, drop_outlier_points=True, perturb_in_camera=True),      # (2000 keyframes: where the oracle's system has an exact -- dense -- solve; make_converged.py)
}
SURVEY_8D_CONVERGED = 1e-6


def _parameter_gaps(G, pose_ref, rho_ref):
    """-> (quaternion gap, translation gap relative to the map's extent, camera-centre gap relative to the extent, inverse-depth gap relative to the value)"""
    qa = G.pose[:, :4]/np.linalg.norm(G.pose[:, :4], axis=1, keepdims=True); qb = pose_ref[:, :4]/np.linalg.norm(pose_ref[:, :4], axis=1, keepdims=True)
    qa = qa*np.sign((qa*qb).sum(1, keepdims=True))
    extent = float(np.abs(pose_ref[:, 4:]).max())
    _, raw = _gauge_aligned_pose_gap(G.pose, pose_ref)             # (the gauge is fixed: the RAW gap of the camera centres, no alignment)
    return (float(np.abs(qa - qb).max()), float(np.abs(G.pose[:, 4:] - pose_ref[:, 4:]).max()/extent), raw,
            float((np.abs(G.rho - rho_ref)/np.abs(rho_ref)).max()))


@pytest.mark.parametrize("name", list(HANDED_OVER))
def test_converged_parameters_on_the_map_as_the_reference_hands_it_over(gpu, name):
    """(A) both sides run the reference's solve to Ceres' own exit (function tolerance 1e-6, optimizer.cc:1833-1846) from the common start: same number of
    iterations and accepted steps, final cost and final parameters within SURVEY 8d's 1e-6.
    (B) both sides with function_tolerance = parameter_tolerance = gradient_tolerance = 0 until no step changes the cost any more -- a true stationary point,
    where the answer does not depend on the iteration an exit test fires in: cost to 1e-9, parameters to 1e-6."""
    from textslam_amd import synth
    fx = np.load(os.path.join(ROOT, "tests", "golden", f"converged_{name}.npz"))
    P = synth.config_global(**HANDED_OVER[name])
    # (A) Ceres' exit
    o = abi.options_global(); o.its[0] = CONVERGED_ITS
    gpu.upload(P, o); rep = gpu.solve(); G = gpu.download(P.copy())
    assert rep["poll_timeouts"] == 0 and rep["pcg_unconverged"] == 0, rep
    assert abs(rep["cost0"][0] - float(fx["cost0"])) <= 1e-11*float(fx["cost0"])
    relA = abs(rep["cost1"][0] - float(fx["cost1"]))/float(fx["cost1"])
    gq, gt, gc, gr = _parameter_gaps(G, fx["pose"], fx["rho"])
    print(f"\n{name} (A) Ceres' exit: GPU {rep['iters'][0]} iterations / {rep['accepted'][0]} accepted, cost {rep['cost1'][0]!r}; oracle {int(fx['iters'])} / {int(fx['accepted'])}, "
          f"cost {float(fx['cost1'])!r}: rel {relA:.2e}; quaternions {gq:.2e}, translations {gt:.2e} and camera centres {gc:.2e} of the map's extent, inverse depths rel {gr:.2e}")
    # (B) a stationary point
    o3 = abi.options_global(); o3.its[0] = 600; o3.function_tolerance = 0.0; o3.parameter_tolerance = 0.0; o3.gradient_tolerance = 0.0
    gpu.upload(P, o3); rep3 = gpu.solve(); G3 = gpu.download(P.copy())
    assert rep3["poll_timeouts"] == 0 and rep3["pcg_unconverged"] == 0, rep3
    relB = abs(rep3["cost1"][0] - float(fx["stationary_cost1"]))/float(fx["stationary_cost1"])
    sq, st, sc, sr = _parameter_gaps(G3, fx["stationary_pose"], fx["stationary_rho"])
    print(f"{name} (B) zero tolerances: GPU {rep3['iters'][0]} iterations / {rep3['accepted'][0]} accepted, termination {rep3['termination'][0]}, cost {rep3['cost1'][0]!r}; oracle "
          f"{int(fx['stationary_iters'])} / {int(fx['stationary_accepted'])}, cost {float(fx['stationary_cost1'])!r}: rel {relB:.2e}; quaternions {sq:.2e}, translations {st:.2e} and "
          f"camera centres {sc:.2e} of the map's extent, inverse depths rel {sr:.2e}")
    _record(name, ceres_exit=dict(iters_gpu=rep["iters"][0], iters_oracle=int(fx["iters"]), accepted_gpu=rep["accepted"][0], accepted_oracle=int(fx["accepted"]),
                                  cost1_gpu=rep["cost1"][0], cost1_oracle=float(fx["cost1"]), rel_cost=relA, quaternion_gap=gq, translation_gap=gt, centre_gap=gc, rho_gap_rel=gr),
            stationary=dict(iters_gpu=rep3["iters"][0], iters_oracle=int(fx["stationary_iters"]), termination_gpu=rep3["termination"][0], cost1_gpu=rep3["cost1"][0],
                            cost1_oracle=float(fx["stationary_cost1"]), rel_cost=relB, quaternion_gap=sq, translation_gap=st, centre_gap=sc, rho_gap_rel=sr))
    assert rep["termination"][0] == 1 and rep["iters"][0] == int(fx["iters"]) and rep["accepted"][0] == int(fx["accepted"]), rep
    assert relA <= SURVEY_8D_CONVERGED and max(gq, gt, gc, gr) <= SURVEY_8D_CONVERGED, (relA, gq, gt, gc, gr)
    assert rep3["iters"][0] < 600, rep3                              # (it ended because no step changes the cost any more, not on the iteration cap)
    assert relB <= 1e-9 and max(sq, st, sc, sr) <= SURVEY_8D_CONVERGED, (relB, sq, st, sc, sr)
