"""GPU diagnostic (not a pytest): the sliding-window tsba_local_ba call (C4) with the point slot pairs of the S blocks built on the host (production for windows) and on the
device (tsba_debug_options.host_pair_lists = 2: tsba_devplan.h, production for maps of more than 126 keyframes): call time, parts, and whether the results are the same bits."""
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
    ws, ups, sol = [], [], []
    for k in range(12):
        G = P.copy(); G.kf_id = 1000 + np.arange(20); G.kf_id[19] = 7000 + k + 100*mode; G.struct()
        t = time.perf_counter(); rep = opt.LocalBundleAdjustment(G, options=o); ws.append((time.perf_counter() - t)*1e3); ups.append(rep["t_upload_ms"]); sol.append(rep["t_solve_ms"])
    res[mode] = G
    print("host_pair_lists %d: sliding call median %.3f min %.3f ms; upload median %.3f min %.3f; solve median %.3f; iters %s cost %.9e" % (mode, np.median(ws[2:]), min(ws[2:]), np.median(ups[2:]), min(ups[2:]), np.median(sol[2:]), rep["iters"], rep["cost1"][-1]), flush=True)
opt.debug_set()
print("same bits:", np.array_equal(res[0].pose, res[2].pose), np.array_equal(res[0].rho, res[2].rho), np.array_equal(res[0].theta, res[2].theta))
