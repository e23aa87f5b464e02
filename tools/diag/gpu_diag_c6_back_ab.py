"""GPU diagnostic (not a pytest): resident solve of the 5000-keyframe open chain (C6) and of C5 with the separators' back substitution in one launch
(production) against a launch per level (tsba_debug_options.sv_per_level = 2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
opt = Optimizer(0)
for name, kw in (("C6", dict(n_kf=5000, n_pt=70000, band=10)), ("C5", dict(n_kf=500, n_pt=50000, band=12))):
    P = synth.config_global(**kw); o = abi.options_global()
    for mode in (0, 2, 0, 2):
        opt.debug_set(sv_per_level=mode)
        opt.upload(P, o)
        ts = []
        for _ in range(5):
            rep = opt.solve(); ts.append(rep["t_solve_ms"])
        print("%s %s: min %.3f ms; iters %s accepted %s cost1 %.9g" % (name, "per level " if mode else "one launch", min(ts), rep["iters"], rep["accepted"], rep["cost1"][0]), flush=True)
