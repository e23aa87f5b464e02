"""GPU diagnostic (not a pytest): the single-vector solve phase (csrc/tsba_bandsv.h) on the 5000-keyframe chain -- error against the factorisation's own
solve; with TSBA_LIB pointing at a -DSV_STAMPS build the cycle stamps of four interiors are printed by the library."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
gpu = Optimizer(0)
P = synth.config_global(n_kf=n_kf, n_pt=14*n_kf, band=10); o = abi.options_global()
gpu.debug_set(sep_solver=2); gpu.upload(P, o)
print(gpu.solver_info())
rb = gpu.reduced_band(o.initial_radius)
R = (-rb["g"]).reshape(-1, 1).copy()
for _ in range(3):
    x = gpu.multi_solve(R, single=True)[:, 0]
print("err vs the factorisation's solve %.2e" % (np.abs(x - rb["dp_rows"]).max()/np.abs(rb["dp_rows"]).max()))
