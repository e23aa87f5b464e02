"""GPU diagnostic (not a pytest): resident solve of C6 with 1 % long-range points -- production (variant 0: the separator tree of the solve phase, the back
substitution of the factorisation's own solve and the iteration's update step each inside one launch) the iteration's update step and the interiors' back substitution each inside a neighbouring launch) against tsba_debug_options.sv_per_level = 8 (the
interiors' back substitution as a launch of its own) and 15 (everything as in round 3: a launch per level, k_pcg_update, k_sv_back_int)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
opt = Optimizer(0)
far = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
P = synth.config_global(n_kf=5000, n_pt=70000, band=10, far_frac=far); o = abi.options_global()
res = {}
for mode in (0, 8, 0, 15):
    opt.debug_set(sv_per_level=mode)
    opt.upload(P, o)
    ts = []
    for _ in range(4):
        rep = opt.solve(); ts.append(rep["t_solve_ms"])
    G = P.copy(); opt.download(G)
    print("tree %s: min %.3f ms  iters %s accepted %s cost1 %.12g  pcg %d its / %d systems, unconverged %d" % ("variant %d" % mode, min(ts), rep["iters"], rep["accepted"],
          rep["cost1"][0], rep["pcg_iterations"], rep["pcg_systems"], rep["pcg_unconverged"]), flush=True)
    if mode in res: assert np.array_equal(res[mode], G.pose)
    res[mode] = G.pose.copy()
print("same bits (not expected: the two back substitutions round differently):", np.array_equal(res[0], res[15]))
