"""GPU diagnostic (not a pytest): resident pose-only optimisation (C3) in a loop, for rocprofv3."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
opt = Optimizer(0)
P3 = synth.config_c3(); o3 = abi.options_pose()
opt.upload(P3, o3)
for k in range(3): opt.solve()
t = time.time()
for k in range(20): rep = opt.solve()
print("pose-only resident solve %.3f ms, iters %s" % ((time.time() - t)*1e3/20, rep['iters']))
