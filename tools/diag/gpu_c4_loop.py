"""GPU diagnostic (not a pytest): the resident C4 window solved in a loop, for rocprofv3 --kernel-trace --stats."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
opt = Optimizer(0)
P = synth.config_c4(); o = abi.options_local()
opt.upload(P, o)
for k in range(3): opt.solve()
t = time.perf_counter()
for k in range(n): rep = opt.solve()
print("C4 resident solve %.3f ms, iters %s, poll_timeouts %d" % ((time.perf_counter() - t)*1e3/n, rep['iters'], rep['poll_timeouts']))
