import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
opt = Optimizer(0)
P3 = synth.config_c3(); o3 = abi.options_pose()
P = P3.copy(); P.kf_id = np.array([7], np.int64)
for k in range(3): opt.PoseOptim(P.copy(), options=o3)
opt.debug_set(verbose=1)
for k in range(2):
    G = P.copy(); t = time.time(); rep = opt.PoseOptim(G, options=o3); print("call ms %.3f upload %.3f" % ((time.time()-t)*1e3, rep["t_upload_ms"]))
print(P.n_pt, P.n_text, [len(x) for x in P.sobs_kf], P.n_tobs)
