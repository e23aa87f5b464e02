"""GPU diagnostic (not a pytest): where the host time of the sliding-window tsba_local_ba call goes (C4 with keyframe identities; verbose timers of
tsba_upload / build_plan / stage_level on stderr)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
opt = Optimizer(0)
P = synth.config_c4(); o = abi.options_local()
def call(k):
    G = P.copy(); G.kf_id = 1000 + np.arange(20); G.kf_id[19] = 7000 + k; G.struct()
    t = time.perf_counter(); rep = opt.LocalBundleAdjustment(G, options=o); return (time.perf_counter() - t)*1e3, rep
ts = sorted(call(k)[0:2] for k in range(8)) if False else [call(k) for k in range(8)]
w = sorted(x[0] for x in ts[1:])
print("sliding call: min %.3f median %.3f ms; upload %.3f solve %.3f download %.3f" % (w[0], w[len(w)//2], min(x[1]["t_upload_ms"] for x in ts[1:]), min(x[1]["t_solve_ms"] for x in ts[1:]), min(x[1]["t_download_ms"] for x in ts[1:])), flush=True)
opt.debug_set(verbose=1)
for k in range(2):
    sys.stderr.write("---- verbose call %d\n" % k); sys.stderr.flush()
    ms, rep = call(20 + k)
    print("verbose call %.3f ms upload %.3f" % (ms, rep["t_upload_ms"]), flush=True)
