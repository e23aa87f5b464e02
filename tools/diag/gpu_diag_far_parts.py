"""GPU diagnostic (not a pytest): resident solve of C6 with 1 % long-range points against the number of interiors of the band solver."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
opt = Optimizer(0)
far = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
P = synth.config_global(n_kf=5000, n_pt=70000, band=10, far_frac=far); o = abi.options_global()
for parts in [int(a) for a in sys.argv[2:]] or [0, 160, 192, 224, 256]:
    opt.debug_set(band_parts=parts)
    opt.upload(P, o)
    ts = []
    for _ in range(3):
        rep = opt.solve(); ts.append(rep["t_solve_ms"])
    print("interiors %3d (asked %3d): min %.3f ms  iters %s accepted %s cost1 %.12g  pcg %d its / %d systems, unconverged %d" % (opt.solver_info()["interiors"], parts, min(ts), rep["iters"], rep["accepted"],
          rep["cost1"][0], rep["pcg_iterations"], rep["pcg_systems"], rep["pcg_unconverged"]), flush=True)
