"""GPU diagnostic (not a pytest): resident global-BA solve time against the number of interiors of the partitioned band solver (tsba_debug_options.band_parts; 0 = the cost model's choice).
usage: python tools/diag/gpu_sweep_parts.py n_kf n_pt band P1 P2 ..."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
n_kf, n_pt, band = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
P = synth.config_global(n_kf=n_kf, n_pt=n_pt, band=band); o = abi.options_global()
opt = Optimizer(0)
for parts in [0] + [int(a) for a in sys.argv[4:]]:
    try:
        opt.debug_set(band_parts=parts)
        opt.upload(P, o); rep = opt.solve()
        ts = []
        for k in range(6):
            t = time.perf_counter(); rep = opt.solve(); ts.append((time.perf_counter() - t)*1e3)
        info = opt.solver_info()
        print("band_parts %3d: interiors %3d band_rows %3d  %.3f ms per solve (min %.3f)  iters %s cost %.6e  solve %.1f us per trial" % (parts, info["interiors"], info["band_rows"], float(np.median(ts)), min(ts), rep["iters"], rep["cost1"][-1], opt.time_solve(10)*1e3), flush=True)
    except Exception as e:
        print("band_parts", parts, "failed:", e, flush=True)
opt.debug_set()
