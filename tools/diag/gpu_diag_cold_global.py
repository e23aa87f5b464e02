"""GPU diagnostic (not a pytest): where the time of a cold tsba_global_ba call on the 5000-keyframe map goes (plan, upload, solve).
usage: python tools/diag/gpu_diag_cold_global.py [loop]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
loop = len(sys.argv) > 1 and sys.argv[1] == "loop"
g = Optimizer(0)
P = synth.config_global(n_kf=5000, n_pt=70000, band=10, loop=loop); o = abi.options_global()
g.debug_set(verbose=1)
for k in range(3):
    G = P.copy(); t = time.time(); rep = g.GlobalBA(G, options=o); dt = (time.time() - t)*1e3
    print("call %d: wall %.1f ms  t_upload_ms %.1f  t_solve_ms %.1f  iters %s" % (k, dt, rep['t_upload_ms'], rep['t_solve_ms'], rep['iters']), flush=True)
