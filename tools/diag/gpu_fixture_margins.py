import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests", "golden"))
import numpy as np
import make_golden
from textslam_amd.optimizer import Optimizer
gpu = Optimizer(0)
V = {"mid_global_long_range": [dict(far_solver=2), dict(far_solver=3, pcg_block=1), dict(far_solver=1), dict(far_solver=1, no_band_stream=1)],
     "mid_global_ring": [dict(), dict(no_ring=1), dict(no_ring=1, no_kf_reorder=1)],
     "mid_global_two_closures": [dict(far_solver=2), dict(far_solver=3), dict(far_solver=1)]}
for name, variants in V.items():
    g = np.load(os.path.join(R, "tests", "golden", name + ".npz")); P, o = make_golden.make_global_case(name)
    for kw in variants:
        gpu.debug_set(**kw); G = P.copy(); rep = gpu.GlobalBA(G, options=o); tr = gpu.lm_trace(0)
        print(name, kw, rep["solver_path"], "cost %.1e radius %.1e pose %.1e rho %.1e" % (np.abs(tr[:,0]/g["trace"][:,0]-1).max(), np.abs(tr[:,2]/g["trace"][:,2]-1).max(), np.abs(G.pose-g["pose"]).max(), np.abs(G.rho-g["rho"]).max()))
gpu.debug_set()
