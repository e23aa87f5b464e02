"""GPU diagnostic (not a pytest): cycle stamps of k_cre_elim (level 1, pivot 3) on the C6 map. Run with TSBA_LIB=textslam_amd/libtsba_stamps.so."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
g = Optimizer(0)
P = synth.config_global(n_kf=5000, n_pt=70000, band=10); o = abi.options_global()
g.upload(P, o); g.solve(); g.solve()
st = (C.c_longlong*64)(); g.lib.tsba_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]; g.lib.tsba_debug_stamps(g.ctx, st)
print("k_cre_elim stamps (cycles): load %d  factor loop %d  stores %d  product %d | per-step work before the barrier, summed: panel wave %d, update wave %d" % tuple(st[48:54]))
print("k_bandp_factor stamps of interior 1 (cycles): factor loops %d  write-out %d  slide %d  loads %d  chunks %d" % (st[32], st[33], st[34], st[35], st[19]))
print("solver info", g.solver_info())
