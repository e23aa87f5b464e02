"""GPU diagnostic (not a pytest): C6-sized global BA (5000 KF, ~460k scene blocks, band covisibility) resident solve, for rocprofv3."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
import ctypes as C
st = (C.c_longlong*64)(); opt.lib.tsba_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]; opt.lib.tsba_debug_stamps(opt.ctx, st)
tot = sum(st[32:36]) or 1
print("band solve phases (clock64 ticks, last launch): factor %d  write-out %d  slide %d  load %d  -> %.1f%% / %.1f%% / %.1f%% / %.1f%%; chunks %d" % (st[32], st[33], st[34], st[35], 100*st[32]/tot, 100*st[33]/tot, 100*st[34]/tot, 100*st[35]/tot, st[19]))
