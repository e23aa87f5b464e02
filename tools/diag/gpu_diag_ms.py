"""GPU diagnostic (not a pytest): the multi-right-hand-side solve phase against scipy, error by row block."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from scipy.linalg import solveh_banded
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
n_kf, band, parts, T = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (900, 9, 17, 3)
gpu = Optimizer(0)
P = synth.config_global(n_kf=n_kf, n_pt=40*n_kf, band=band); o = abi.options_global()
gpu.debug_set(band_parts=parts, sep_solver=2); gpu.upload(P, o)
rb = gpu.reduced_band(o.initial_radius)
rng = np.random.default_rng(5)
R = rng.standard_normal((rb["n"], T))*np.abs(rb["g"]).max(); R[:, 0] = -rb["g"]
X = gpu.multi_solve(R); ref = solveh_banded(rb["ab"], R, lower=True)
err = np.abs(X - ref).max(axis=1).reshape(-1, 6).max(axis=1)/np.abs(ref).max()
nb = err.size; B = band
tot = nb - (parts - 1)*B; q = tot//parts; rem = tot - q*parts
print("blocks", nb, "interior len", q, "rem", rem)
bad = np.nonzero(err > 1e-8)[0]
print("bad blocks", len(bad), bad[:60])
for p in range(min(parts, 6)):
    a = p*(q + B) + min(p, rem); b = a + q + (1 if p < rem else 0)
    print("interior", p, "[", a, b, ") err max interior %.2e" % err[a:b].max(), "right sep %.2e" % (err[b:b+B].max() if p < parts-1 else 0), "first bad in interior", (np.nonzero(err[a:b] > 1e-8)[0][:5] + a))
