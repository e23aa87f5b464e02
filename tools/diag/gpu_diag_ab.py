"""A/B timing of the resident C4 LocalBundleAdjustment solve (run twice in one gpurun call with different TSBA_* switches)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from textslam_amd import synth
from textslam_amd.abi import options_local
from textslam_amd.optimizer import Optimizer
P = synth.config_c4()
opt = Optimizer(0)
opt.upload(P, options_local())
ts = []
for it in range(25):
    t = time.perf_counter(); r = opt.solve(); ts.append((time.perf_counter() - t)*1e3)
ts = np.array(ts[5:])
print("solve ms  min %.3f  median %.3f   iters %s cost1 %s" % (ts.min(), np.median(ts), r["iters"], ["%.9e" % c for c in r["cost1"]]), flush=True)
