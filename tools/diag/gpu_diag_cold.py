import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time, numpy as np, os
from textslam_amd import synth
from textslam_amd.abi import options_local
from textslam_amd.optimizer import Optimizer
P = synth.config_c4()
opt = Optimizer(0)
o = options_local()
for it in range(4):
    g = P.copy(); g.struct()
    t = time.perf_counter(); r = opt.LocalBundleAdjustment(g, options=o); dt = (time.perf_counter()-t)*1e3
    print("cold %.2f ms  upload %.2f solve %.2f download %.2f" % (dt, r["t_upload_ms"], r["t_solve_ms"], r["t_download_ms"]), flush=True)
