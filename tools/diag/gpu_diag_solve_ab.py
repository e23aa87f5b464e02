"""GPU diagnostic (not a pytest): the two schedules of the small-window solve (tsba_debug_options.solve_variant) on the C4 window."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer

opt = Optimizer(0)
P = synth.config_c4(); o = abi.options_local()
res = {}
for var in (0, 1, 2, 0, 1):
    opt.debug_set(solve_variant=var)
    opt.upload(P, o)
    ts = []
    for _ in range(6):
        rep = opt.solve(); ts.append(rep["t_solve_ms"])
    G = opt.download(P.copy())
    rs = opt.reduced_system(o.initial_radius)
    ms = opt.time_solve(200)
    print("variant %d: solve %.3f ms (min of 6), k_solve back-to-back %.2f us, iters %s accepted %s cost1 %s" % (var, min(ts), ms*1e3, rep["iters"], rep["accepted"], ["%.10g" % c for c in rep["cost1"]]))
    if var in res:
        assert np.array_equal(res[var][0], G.pose), "not reproducible"
    res[var] = (G.pose.copy(), rs["dp"].copy(), rep)
d01 = np.abs(res[0][1] - res[1][1]).max()/np.abs(res[1][1]).max()
print("first step, two-panel vs look-ahead schedule: max rel diff %.2e; final poses max abs diff %.2e" % (d01, np.abs(res[0][0] - res[1][0]).max()))
S = rs["S"]; free = np.nonzero(rs["free"])[0]; idx = np.concatenate([np.arange(6*k, 6*k + 6) for k in free])
m = len(idx)
ref = -np.linalg.solve(S[:m, :m], rs["g"][:m])
for var in (0, 1):
    got = np.concatenate([res[var][1][6*k:6*k + 6] for k in free])
    print("variant %d first step vs numpy: %.2e" % (var, np.abs(got - ref).max()/np.abs(ref).max()))
