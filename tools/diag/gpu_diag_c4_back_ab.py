"""GPU diagnostic (not a pytest): resident C4 solve with the back-substitution in the solver's launch (production, k_solve_back) against a launch of its
own (tsba_debug_options.solve_variant = 3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
opt = Optimizer(0)
P = synth.config_c4(); o = abi.options_local()
res = {}
for mode in (0, 3, 0, 3):
    opt.debug_set(solve_variant=mode)
    opt.upload(P, o)
    ts = []
    for _ in range(30):
        rep = opt.solve(); ts.append(rep["t_solve_ms"])
    G = opt.download(P.copy())
    ts.sort()
    print("%s: min %.4f median %.4f ms  iters %s cost1 %.12g" % ("one launch   " if mode == 0 else "two launches ", ts[0], ts[len(ts)//2], rep["iters"], rep["cost1"][2]), flush=True)
    if mode in res: assert np.array_equal(res[mode], G.pose)
    res[mode] = G.pose.copy()
print("bit-identical:", np.array_equal(res[0], res[3]))
