"""GPU diagnostic (not a pytest): the sliding-window tsba_local_ba call (C4 with keyframe identities) in a loop, for tools/diag/timeline_cold_call.sh."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
opt = Optimizer(0)
P = synth.config_c4(); o = abi.options_local()
for k in range(8):
    G = P.copy(); G.kf_id = 1000 + np.arange(20); G.kf_id[19] = 7000 + k; G.struct()
    t = time.perf_counter(); rep = opt.LocalBundleAdjustment(G, options=o); dt = (time.perf_counter() - t)*1e3
    print("sliding call %d: wall %.3f ms upload %.3f solve %.3f download %.3f" % (k, dt, rep["t_upload_ms"], rep["t_solve_ms"], rep["t_download_ms"]), flush=True)
    time.sleep(0.002)
