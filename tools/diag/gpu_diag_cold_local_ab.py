"""GPU diagnostic (not a pytest): the cold tsba_local_ba call on C4 with the slot-pair lists of the S blocks built on the host (production for windows)
against the device build (tsba_debug_options.host_pair_lists = 2 forces it for small windows too)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
opt = Optimizer(0)
P = synth.config_c4(); o = abi.options_local()
res = {}
for mode in (0, 2, 0, 2):
    opt.debug_set(host_pair_lists=mode)
    ts = []
    for k in range(8):
        G = P.copy(); G.kf_id = 1000 + np.arange(20); G.kf_id[19] = 7000 + k; G.struct()
        t = time.perf_counter(); rep = opt.LocalBundleAdjustment(G, options=o); ts.append(((time.perf_counter() - t)*1e3, rep['t_upload_ms'], rep['t_solve_ms']))
    ts.sort()
    print("lists on %s: wall min %.3f median %.3f ms (upload %.3f, solve %.3f)" % ("device" if mode == 2 else "host", ts[0][0], ts[4][0], ts[0][1], ts[0][2]), rep["iters"], "%.10g" % rep["cost1"][2], flush=True)
    if mode in res: assert np.array_equal(res[mode], G.pose)
    res[mode] = G.pose.copy()
print("bit-identical results host / device lists:", np.array_equal(res[0], res[2]))
