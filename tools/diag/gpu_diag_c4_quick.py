"""GPU diagnostic (not a pytest): resident C4 solve time, a few repetitions, and the kernel split from HIP-side timing hooks."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
opt = Optimizer(0)
P = synth.config_c4(); o = abi.options_local()
opt.upload(P, o)
ts = []
for _ in range(30):
    rep = opt.solve(); ts.append(rep["t_solve_ms"])
ts.sort()
print("C4 resident solve: min %.4f median %.4f ms; iters %s accepted %s cost1 %s" % (ts[0], ts[len(ts)//2], rep["iters"], rep["accepted"], ["%.12g" % c for c in rep["cost1"]]))
