"""GPU diagnostic (not a pytest): the one-shot tsba_pose_optim call (C3) as the tracking thread makes it per frame -- wall time per call and its parts."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
opt = Optimizer(0)
if len(sys.argv) > 1 and sys.argv[1] == "verbose": opt.debug_set(verbose=1)
P3 = synth.config_c3(); o3 = abi.options_pose()
ws = []
for k in range(40):
    G = P3.copy(); t = time.perf_counter(); rep = opt.PoseOptim(G, options=o3); dt = (time.perf_counter() - t)*1e3
    ws.append(dt)
    if k < 12 or dt > 1.0: print("pose-only C3 call %2d: wall %.3f ms  t_upload_ms %.3f  t_solve_ms %.3f  t_download_ms %.3f iters %s" % (k, dt, rep['t_upload_ms'], rep['t_solve_ms'], rep['t_download_ms'], rep['iters']))
ws = np.array(ws[2:])
print("median %.3f  min %.3f  max %.3f  mean %.3f" % (np.median(ws), ws.min(), ws.max(), ws.mean()))
