import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
g = Optimizer(0)
def timed(name, fn, n=14):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); r = fn(); ts.append((time.perf_counter() - t)*1e3)
    print("%-52s median %.3f ms  min %.3f ms" % (name, float(np.median(ts[2:])), min(ts[2:])), flush=True)
    return r
P = synth.init_pair(seed=5); o = abi.options_init()
P2 = synth.landmark_refine(); o2 = abi.options_landmarker()
o3 = abi.options_theta()
Pt = synth.tiny(seed=7); ol = abi.options_local()
for tl in (0, 2):
    g.debug_set(trial_launches=tl)
    r = timed("trial_launches %d InitBA" % tl, lambda: g.InitBA(P.copy(), options=o)); print("   iters", r["iters"] if isinstance(r, dict) else None)
    timed("trial_launches %d OptimizeLandmarker" % tl, lambda: g.OptimizeLandmarker(P2.copy(), options=o2))
    timed("trial_launches %d ThetaOptimMultiFs" % tl, lambda: g.ThetaOptimMultiFs(P2.copy(), text=1, options=o3))
    timed("trial_launches %d LocalBA tiny (5 KF, 60 pts, 4 planes)" % tl, lambda: g.LocalBundleAdjustment(Pt.copy(), options=ol))
g.debug_set()
