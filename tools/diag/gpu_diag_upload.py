"""GPU diagnostic (not a pytest): where the time of a cold tsba_local_ba call goes (upload / plan construction vs solve)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
opt = Optimizer(0)
if len(sys.argv) > 1 and sys.argv[1] == "verbose": opt.debug_set(verbose=1)
P = synth.config_c4(); o = abi.options_local()
for k in range(4):
    G = P.copy(); t = time.time(); rep = opt.LocalBundleAdjustment(G, options=o); dt = (time.time() - t)*1e3
    print("call %d: wall %.2f ms  t_upload_ms %.2f  t_solve_ms %.2f" % (k, dt, rep['t_upload_ms'], rep['t_solve_ms']))
for k in range(3):
    t = time.time(); opt.upload(P, o); dt = (time.time() - t)*1e3
    print("upload only %d: %.2f ms" % (k, dt))
P3 = synth.config_c3(); o3 = abi.options_pose()
for k in range(4):
    G = P3.copy(); t = time.time(); rep = opt.PoseOptim(G, options=o3); dt = (time.time() - t)*1e3
    print("pose-only C3 call %d: wall %.2f ms  t_upload_ms %.2f  t_solve_ms %.2f  iters %s" % (k, dt, rep['t_upload_ms'], rep['t_solve_ms'], rep['iters']))
