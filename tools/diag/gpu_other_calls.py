"""GPU diagnostic (not a pytest): one-shot call times of the optimizer:: entry points the bench line does not carry (InitBA, OptimizeLandmarker, ThetaOptimMultiFs, PoseOptim on a small frame)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
g = Optimizer(0)
def timed(name, fn, n=12):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append((time.perf_counter() - t)*1e3)
    print("%-44s median %.3f ms  min %.3f ms" % (name, float(np.median(ts[2:])), min(ts[2:])), flush=True)
P = synth.init_pair(seed=5); o = abi.options_init()
timed("InitBA (2 KF, 300 points, 3 planes, 4 passes)", lambda: g.InitBA(P.copy(), options=o))
P2 = synth.landmark_refine(); o2 = abi.options_landmarker()
timed("OptimizeLandmarker (5 KF, 150 points, 4 planes)", lambda: g.OptimizeLandmarker(P2.copy(), options=o2))
o3 = abi.options_theta()
timed("ThetaOptimMultiFs (one plane)", lambda: g.ThetaOptimMultiFs(P2.copy(), text=1, options=o3))
P4 = synth.config_c3(); o4 = abi.options_pose()
timed("PoseOptim (C3: 3000 points + 200 text pixels)", lambda: g.PoseOptim(P4.copy(), options=o4))
