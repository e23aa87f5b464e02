"""GPU diagnostic (not a pytest): the pose-only call TextSLAM makes per frame (C3: 3000 points + 200 text pixels) -- one-shot tsba_pose_optim on a warm
context: wall time per call and the library's own upload / download split; with and without a keyframe id (plane cache)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
opt = Optimizer(0)
P3 = synth.config_c3(); o3 = abi.options_pose()
for ids in (False, True):
    P = P3.copy()
    if ids: P.kf_id = np.array([7], np.int64)
    for k in range(3): rep = opt.PoseOptim(P.copy(), options=o3)
    ts = []
    for k in range(10):
        G = P.copy(); t = time.time(); rep = opt.PoseOptim(G, options=o3); ts.append((time.time() - t)*1e3)
    print("ids" if ids else "no ids", "call ms min %.3f median %.3f" % (min(ts), sorted(ts)[5]), "upload %.3f download %.3f solve %.3f iters %s" % (rep.get("t_upload_ms", -1), rep.get("t_download_ms", -1), rep.get("t_solve_ms", -1), rep["iters"]))
