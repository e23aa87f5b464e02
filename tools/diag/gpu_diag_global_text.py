"""GPU diagnostic (not a pytest): global BA WITH text planes (BASELINE config 3: 500 KF x 50k points x 1k planes, full Schur LM).
The reference's GlobalBA switches text off (optimizer.cc:1707); the kernels do not care -- parity at a size the oracle handles,
then the C5-sized timing."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
import oracle
g = Optimizer(0)
P = synth.make_problem(n_kf=60, n_pt=3000, n_text=40, seed=5, feats=(32, 16, 8), max_targets=8, text_targets=5, frozen_frac=0.0, band=10, n_levels=3, rot_deg=0.2, trans_m=0.01)
o = abi.options_global(); o.use_text = 1; o.its[0] = 8
R = P.copy(); rep = oracle.solve(R, o)
G = P.copy(); rg = g.GlobalBA(G, options=o)
print("60 KF / 40 planes: oracle", rep["iters"], rep["accepted"], rep["cost1"], " gpu", rg["iters"], rg["accepted"], rg["cost1"],
      " pose diff %.1e theta diff %.1e" % (np.abs(G.pose - R.pose).max(), np.abs(G.theta - R.theta).max()))
t = time.time()
P = synth.make_problem(n_kf=500, n_pt=50000, n_text=1000, seed=7, feats=(64, 24, 12), max_targets=8, text_targets=5, frozen_frac=0.0, band=12, n_levels=1, rot_deg=0.2, trans_m=0.01)
print("C5 synth %.1f s, n_tobs %d" % (time.time() - t, P.n_tobs))
o = abi.options_global(); o.use_text = 1
t = time.time(); g.upload(P, o); print("upload %.1f ms" % ((time.time() - t)*1e3))
for _ in range(2):
    t = time.time(); rep = g.solve()
    print("C5 with text: solve %.1f ms" % ((time.time() - t)*1e3), rep["iters"], rep["accepted"], rep["termination"], rep["cost0"], rep["cost1"], rep["n_sblock"], rep["n_tblock"], flush=True)
