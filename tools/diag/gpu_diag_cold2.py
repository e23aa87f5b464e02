"""GPU diagnostic (not a pytest): where a cold tsba_local_ba call on C4 goes, with and without the plane cache (tsba_problem.kf_id)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
opt = Optimizer(0)
opt.debug_set(verbose=1)
P = synth.config_c4(); o = abi.options_local()
for k in range(4):
    G = P.copy(); G.struct(); t = time.perf_counter(); rep = opt.LocalBundleAdjustment(G, options=o); dt = (time.perf_counter() - t)*1e3
    print("no ids, call %d: wall %.2f ms  t_upload_ms %.2f  t_solve_ms %.2f t_download_ms %.2f" % (k, dt, rep['t_upload_ms'], rep['t_solve_ms'], rep['t_download_ms']), flush=True)
for k in range(4):
    G = P.copy(); G.kf_id = 1000 + np.arange(20); G.kf_id[19] = 7000 + k; G.struct()
    t = time.perf_counter(); rep = opt.LocalBundleAdjustment(G, options=o); dt = (time.perf_counter() - t)*1e3
    print("ids, call %d: wall %.2f ms  t_upload_ms %.2f  t_solve_ms %.2f t_download_ms %.2f" % (k, dt, rep['t_upload_ms'], rep['t_solve_ms'], rep['t_download_ms']), opt.img_cache_stats(), flush=True)
