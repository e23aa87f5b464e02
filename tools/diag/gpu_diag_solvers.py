"""GPU diagnostic (not a pytest): the two small-system solvers on the C4 window -- time per launch and agreement."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
g = Optimizer(0)
for name, P in (("C4", synth.config_c4()), ("31 KF", synth.make_problem(n_kf=31, n_pt=2000, n_text=10, seed=5, band=12)), ("10 KF", synth.config_c1())):
    o = abi.options_local()
    for solver in (1, 2):
        g.debug_set(small_solver=solver)
        g.upload(P, o)
        rep = g.solve(); rep = g.solve()
        import time
        t = time.perf_counter(); n = 10
        for _ in range(n): rep = g.solve()
        dt = (time.perf_counter() - t)/n*1e3
        print("%-6s solver %d: solve %.3f ms  iters %s  time_solve %.2f us  cost1 %s" % (name, solver, dt, rep["iters"], g.time_solve(200)*1e3, rep["cost1"]), flush=True)
g.debug_set()
