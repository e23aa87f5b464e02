"""GPU diagnostic (not a pytest): C6-sized global BA (5000 KF, ~460k scene blocks, band covisibility) resident solve, for rocprofv3."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
nkf = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
npt = int(sys.argv[2]) if len(sys.argv) > 2 else 70000
opt = Optimizer(0)
t = time.time(); P = synth.config_global(n_kf=nkf, n_pt=npt, band=10); print("synth s", time.time() - t)
o = abi.options_global()
t = time.time(); opt.upload(P, o); print("upload s", time.time() - t)
for _ in range(2):
    t = time.time(); rep = opt.solve()
    print("solve ms %.1f" % ((time.time() - t)*1e3), rep['iters'], rep['accepted'], rep['termination'], rep['cost0'], rep['cost1'], rep['n_sblock'], flush=True)
