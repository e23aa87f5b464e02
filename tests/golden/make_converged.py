"""TEST INFRASTRUCTURE.  The CPU oracle run TO CONVERGENCE (Ceres' function tolerance 1e-6, the exit optimizer.cc:1833-1846 takes on a map that is
given enough iterations) on the two 5000-keyframe maps whose 12-trial LM prefix parts from the GPU's (open chain, 1 % long-range points): the end
states are committed as fixtures (the runs take 3 and 5 minutes of one host core) and tests/test_gpu_fullsize_oracle.py compares the GPU's converged
answer with them.

    python tests/golden/make_converged.py          # -> tests/golden/converged_<name>.npz
"""
import os
import sys
import time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

CONFIGS = {
    "c6_open_chain": dict(n_kf=5000, n_pt=70000, band=10),
    "c6_long_range": dict(n_kf=5000, n_pt=70000, band=10, far_frac=0.01),
    # round 6: the same two maps AS THE REFERENCE HANDS THEM TO GlobalBA.  (i) map::GetAllMapPoints(false) (src/optimizer.cc:337-341, src/map.cc:36-47) holds no
    # point that any local BA has flagged (tracking::mpPtsCondUpdate, src/tracking.cc:2215-2230): synth's drop_outlier_points -- 105 000 candidates leave ~70 000
    # points.  (ii) GlobalBA runs after the loop correction (loopClosing.cc:587-591): every camera is near its place -- synth's perturb_in_camera moves the CAMERA by
    # 0.2 degrees / 1 cm instead of turning it around the world origin (1.7 m at keyframe 5000, the start the two configurations above inherited from the windows).
    # For these two the oracle is also run with function_tolerance = parameter_tolerance = gradient_tolerance = 0 until no step changes the cost any more: a true
    # stationary point (the `stationary_*` arrays), where "converged parameters" does not depend on which iteration an exit test happened to fire in.
    "c6_open_chain_handed_over": dict(n_kf=5000, n_pt=105000, band=10, drop_outlier_points=True, perturb_in_camera=True),
    # The map with 1 % long-range points at 2000 keyframes: the reduced system of such a map has no sparse factor (at 5000 keyframes it fills to a dense 30 000 x
    # 30 000 matrix, and the plug that tests/test_gpu_fullsize_oracle.py uses there -- GMRES to 1e-13 -- does not get below 1e-11 once the trust region has grown
    # to 1e9: the oracle ended with five invalid steps after 53 minutes).  At 2000 keyframes the 12 000 x 12 000 system is solved EXACTLY by LAPACK's dense
    # Cholesky (dense_solver below), ~6 s per LM iteration, which the 200-iteration run to a stationary point can afford.
    "c6_long_range_handed_over": dict(n_kf=2000, n_pt=42000, band=10, far_frac=0.01, drop_outlier_points=True, perturb_in_camera=True),
}
MAX_ITS = 3000
STATIONARY_ITS = 600


def main():
    import oracle
    from textslam_amd import synth, abi

    def dense_solver(A, rhs):
        import scipy.linalg
        return scipy.linalg.solve(A.toarray(), rhs, assume_a="pos", check_finite=False)

    def solve(P, o, plug, cap=64):
        oracle.set_sparse_solver((dense_solver if P.n_kf <= 2500 else oracle.sparse_solver) if plug else None)      # (an exact solve of the block-sparse system where the band Cholesky does not apply)
        try:
            return oracle.solve_traced(P, o, cap=cap)
        finally:
            oracle.set_sparse_solver(None)

    for name, kw in CONFIGS.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        plug = "long_range" in name
        P = synth.config_global(**kw); o = abi.options_global(); o.its[0] = MAX_ITS
        R = P.copy()
        t0 = time.time()
        rep, tr = solve(R, o, plug)
        print(name, "iterations", rep["iters"][0], "accepted", rep["accepted"][0], "termination", rep["termination"][0],
              "cost", rep["cost0"][0], "->", rep["cost1"][0], f"{time.time() - t0:.0f} s", flush=True)
        assert rep["termination"][0] == 1                        # function tolerance
        # how sharply the reference's own exit defines "converged" on this map: the oracle started again AT its answer (initial trust region, as any new
        # solve) goes on until the function tolerance ends it a second time
        Q = R.copy(); o2 = abi.options_global(); o2.its[0] = 200
        rep2, tr2 = solve(Q, o2, plug)
        print("   started again at its answer:", rep2["iters"][0], "iterations, termination", rep2["termination"][0], "cost", rep2["cost0"][0], "->", rep2["cost1"][0],
              f"(moved {(rep2['cost0'][0] - rep2['cost1'][0])/rep2['cost0'][0]:.2e})", flush=True)
        out = dict(pose=R.pose, rho=R.rho, cost0=rep["cost0"][0], cost1=rep["cost1"][0],
                   iters=rep["iters"][0], accepted=rep["accepted"][0], term=rep["termination"][0], trace=np.array(tr[0]),
                   again_cost1=rep2["cost1"][0], again_iters=rep2["iters"][0], again_term=rep2["termination"][0], again_trace=np.array(tr2[0]))
        if name.endswith("_handed_over"):
            S = P.copy(); o3 = abi.options_global(); o3.its[0] = STATIONARY_ITS
            o3.function_tolerance = 0.0; o3.parameter_tolerance = 0.0; o3.gradient_tolerance = 0.0
            t0 = time.time()
            rep3, tr3 = solve(S, o3, plug, cap=STATIONARY_ITS)
            tr3 = np.array(tr3[0])
            print("   zero tolerances:", rep3["iters"][0], "iterations, accepted", rep3["accepted"][0], "termination", rep3["termination"][0], "cost", repr(rep3["cost1"][0]),
                  "last model cost changes", tr3[-3:, 1], f"{time.time() - t0:.0f} s", flush=True)
            assert rep3["iters"][0] < STATIONARY_ITS              # (it ended because no step changed the cost, not on the cap)
            out.update(stationary_pose=S.pose, stationary_rho=S.rho, stationary_cost1=rep3["cost1"][0], stationary_iters=rep3["iters"][0],
                       stationary_accepted=rep3["accepted"][0], stationary_term=rep3["termination"][0], stationary_trace=tr3)
        np.savez_compressed(os.path.join(HERE, f"converged_{name}.npz"), **out)


if __name__ == "__main__":
    main()
