"""TEST INFRASTRUCTURE.  How sharply the reference's own solve defines its answer on the two 5000-keyframe maps that keep their 5 % gross outliers: the CPU oracle
run to Ceres' function-tolerance exit (optimizer.cc:1833-1846) TWICE, with two exact solvers of the same reduced system -- its built-in band Cholesky (the committed
fixture tests/golden/converged_c6_open_chain.npz) and the block-sparse storage with the plugged scipy solve (LAPACK banded Cholesky under reverse Cuthill-McKee).
Same arithmetic for every residual, Jacobian and normal-equation entry; the two differ in the elimination order of ONE linear solve per trial, i.e. by cond x eps
in the step.  If these two part by more than 1e-6, "converged parameters rel 1e-6"  is not a property of the reference's own answer on that input.

    python tests/golden/make_oracle_vs_oracle.py      # -> profiles/r06_oracle_vs_oracle.json   (about 10 minutes of one host core)
"""
import json
import os
import sys
import time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def centres(pose):
    q, t = pose[:, :4]/np.linalg.norm(pose[:, :4], axis=1, keepdims=True), pose[:, 4:]
    w, x, y, z = q.T
    R = np.stack([1 - 2*(y*y + z*z), 2*(x*y - w*z), 2*(x*z + w*y), 2*(x*y + w*z), 1 - 2*(x*x + z*z), 2*(y*z - w*x),
                  2*(x*z - w*y), 2*(y*z + w*x), 1 - 2*(x*x + y*y)], axis=1).reshape(-1, 3, 3)
    return -np.einsum("nji,nj->ni", R, t)


def main():
    import oracle
    from textslam_amd import synth, abi
    out = {}
    for name, kw in {"c6_open_chain": dict(n_kf=5000, n_pt=70000, band=10)}.items():
        fx = np.load(os.path.join(HERE, f"converged_{name}.npz"))
        P = synth.config_global(**kw); o = abi.options_global(); o.its[0] = 3000
        R = P.copy(); t0 = time.time()
        oracle.set_sparse_solver(oracle.sparse_solver)
        try:
            rep, tr = oracle.solve_traced(R, o)
        finally:
            oracle.set_sparse_solver(None)
        A, B = centres(R.pose), centres(fx["pose"])
        extent = float(np.linalg.norm(B - B.mean(0), axis=1).max())
        e = dict(band_cholesky=dict(iterations=int(fx["iters"]), accepted=int(fx["accepted"]), cost=float(fx["cost1"])),
                 block_sparse_plugged=dict(iterations=int(rep["iters"][0]), accepted=int(rep["accepted"][0]), cost=float(rep["cost1"][0]), termination=int(rep["termination"][0])),
                 rel_cost=abs(float(rep["cost1"][0]) - float(fx["cost1"]))/float(fx["cost1"]),
                 camera_centre_gap_of_extent=float(np.linalg.norm(A - B, axis=1).max()/extent),
                 rho_gap_rel_max=float((np.abs(R.rho - fx["rho"])/np.abs(fx["rho"])).max()), seconds=time.time() - t0)
        print(name, e, flush=True)
        out[name] = e
    json.dump(out, open(os.path.join(ROOT, "profiles", "r06_oracle_vs_oracle.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
