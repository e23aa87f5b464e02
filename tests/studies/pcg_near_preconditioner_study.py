import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, time
import scipy.linalg as sl
from textslam_amd import synth, abi
import oracle
n_kf = int(sys.argv[1]); far = float(sys.argv[2]); closures = int(sys.argv[3]); band = 8
radius = float(sys.argv[4]) if len(sys.argv) > 4 else 1e4
P = synth.config_global(n_kf=n_kf, n_pt=14*n_kf, band=band, far_frac=far, closures=closures)
o = abi.options_global()
rs = oracle.reduced_system(P, o, 0, radius)
S, g = rs["S"].copy(), rs["g"].copy(); n = S.shape[0]; S = 0.5*(S + S.T)
# near-only problem: drop the observations of landmarks whose poses span more than `band`
kf, pt = P.sobs_kf[0], P.sobs_pt[0]; host = P.pt_host
lo = np.full(P.n_pt, 10**9); hi = np.full(P.n_pt, -1)
np.minimum.at(lo, pt, kf); np.maximum.at(hi, pt, kf); lo = np.minimum(lo, host); hi = np.maximum(hi, host)
wide = (hi - lo) > band
Q = P.copy(); keep = ~wide[pt]
Q.sobs_kf[0], Q.sobs_pt[0], Q.sobs_flag[0], Q.sobs_uv0[0] = kf[keep], pt[keep], P.sobs_flag[0][keep], P.sobs_uv0[0].reshape(-1, 2)[keep]
rn = oracle.reduced_system(Q, o, 0, radius)
Sn = 0.5*(rn["S"] + rn["S"].T)
assert Sn.shape == S.shape, (Sn.shape, S.shape)
print("n", n, "wide landmarks", wide.sum(), "obs dropped", (~keep).sum())
i, j = np.indices((n, n)); mask = np.abs(i//6 - j//6) <= band
print("near system outside band:", np.abs(np.where(mask, 0, Sn)).max())
b = -g; xref = np.linalg.solve(S, b)
def pcg(Minv, tol=1e-10, maxit=400):
    x = np.zeros(n); r = b.copy(); z = Minv(r); p = z.copy(); rz = r@z; rz0 = rz; its = 0
    while its < maxit and rz > tol*tol*rz0:
        q = S@p; al = rz/(p@q); x += al*p; r -= al*q; z = Minv(r); rzn = r@z; p = z + (rzn/rz)*p; rz = rzn; its += 1
    return x, its
for name, M in (("band(S)", np.where(mask, S, 0.0)), ("S_near", Sn)):
    c = sl.cho_factor(M); Minv = lambda r, c=c: sl.cho_solve(c, r)
    x, its = pcg(Minv)
    w = sl.eigh(S, M, eigvals_only=True)
    print(name, "PCG its", its, [pcg(Minv, t)[1] for t in (1e-6, 1e-8)], "err %.2g" % (np.abs(x - xref).max()/np.abs(xref).max()),
          "eig min %.3g max %.3g  >1.5: %d  >1.1: %d  <0.9: %d <0.5: %d" % (w.min(), w.max(), (w > 1.5).sum(), (w > 1.1).sum(), (w < 0.9).sum(), (w < 0.5).sum()))
    print("   largest", np.round(w[-12:], 2))
