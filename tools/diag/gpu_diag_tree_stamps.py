"""GPU diagnostic (not a pytest): wall-clock stamps of k_sv_cre_tree (pivot 1, the pivot below the top) on the 5000-keyframe chain.
Run with TSBA_LIB=textslam_amd/libtsba_stamps.so."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
gpu = Optimizer(0)
P = synth.config_global(n_kf=5000, n_pt=70000, band=10); o = abi.options_global()
gpu.debug_set(sep_solver=2); gpu.upload(P, o)
print(gpu.solver_info())
rb = gpu.reduced_band(o.initial_radius)
R = (-rb["g"]).reshape(-1, 1).copy()
gpu.lib.tsba_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
for rep in range(4):
    x = gpu.multi_solve(R, single=True)[:, 0]
    st = (C.c_longlong*64)(); gpu.lib.tsba_debug_stamps(gpu.ctx, st)
    t0 = min(st[0], st[8])
    f = lambda a: " ".join("%7.2f" % ((st[a + k] - t0)*0.01) for k in range(6))
    print("us since the first of the two started | pivot 1: start, state known, pending polled, updates out, neighbours polled, end: %s" % f(0))
    print("                                        | pivot below the top                                                         : %s" % f(8))
print("err vs the factorisation's solve %.2e" % (np.abs(x - rb["dp_rows"]).max()/np.abs(rb["dp_rows"]).max()))
