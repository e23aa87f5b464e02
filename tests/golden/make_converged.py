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
}
MAX_ITS = 3000


def main():
    import oracle
    from textslam_amd import synth, abi
    for name, kw in CONFIGS.items():
        P = synth.config_global(**kw); o = abi.options_global(); o.its[0] = MAX_ITS
        R = P.copy()
        t0 = time.time()
        oracle.set_sparse_solver(oracle.sparse_solver if name == "c6_long_range" else None)      # (an exact solve of the block-sparse system where the band Cholesky does not apply)
        try:
            rep, tr = oracle.solve_traced(R, o)
        finally:
            oracle.set_sparse_solver(None)
        print(name, "iterations", rep["iters"][0], "accepted", rep["accepted"][0], "termination", rep["termination"][0],
              "cost", rep["cost0"][0], "->", rep["cost1"][0], f"{time.time() - t0:.0f} s")
        assert rep["termination"][0] == 1                        # function tolerance
        # how sharply the reference's own exit defines "converged" on this map: the oracle started again AT its answer (initial trust region, as any new
        # solve) goes on until the function tolerance ends it a second time
        Q = R.copy(); o2 = abi.options_global(); o2.its[0] = 200
        oracle.set_sparse_solver(oracle.sparse_solver if name == "c6_long_range" else None)
        try:
            rep2, tr2 = oracle.solve_traced(Q, o2)
        finally:
            oracle.set_sparse_solver(None)
        print("   started again at its answer:", rep2["iters"][0], "iterations, termination", rep2["termination"][0], "cost", rep2["cost0"][0], "->", rep2["cost1"][0],
              f"(moved {(rep2['cost0'][0] - rep2['cost1'][0])/rep2['cost0'][0]:.2e})")
        np.savez_compressed(os.path.join(HERE, f"converged_{name}.npz"), pose=R.pose, rho=R.rho, cost0=rep["cost0"][0], cost1=rep["cost1"][0],
                            iters=rep["iters"][0], accepted=rep["accepted"][0], term=rep["termination"][0], trace=np.array(tr[0]),
                            again_cost1=rep2["cost1"][0], again_iters=rep2["iters"][0], again_term=rep2["termination"][0], again_trace=np.array(tr2[0]))


if __name__ == "__main__":
    main()
