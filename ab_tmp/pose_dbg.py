import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
g = Optimizer(0)
P, o = synth.config_c3(), abi.options_pose()
for old in (1, 0, 0):
    g.debug_set(pass_launches=old)
    g.upload(P, o); rep = g.solve(); G = g.download(P.copy())
    print(old, {k: rep[k] for k in ("iters", "accepted", "termination", "cost0", "cost1", "n_sblock", "n_tblock", "n_bad_scene", "n_bad_tfeat", "n_bad_text", "poll_timeouts")}, G.pose.reshape(-1, 7)[-1][:3])
import time
for old in (1, 0):
    g.debug_set(pass_launches=old); g.upload(P, o); g.solve()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); g.solve(); ts.append((time.perf_counter() - t0)*1e3)
    print("pass_launches", old, "median ms", np.median(ts), "min", min(ts))
