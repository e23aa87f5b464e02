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
i, j = np.indices((n, n)); mask = np.abs(i//6 - j//6) <= band
M = np.where(mask, S, 0.0)
# banded Cholesky of M via scipy
ab = np.zeros((6*band + 6, n))
for d in range(6*band + 6): ab[d, :n - d] = np.diagonal(M, -d)
cb = sl.cholesky_banded(ab, lower=True)
Minv = lambda r: sl.cho_solve_banded((cb, True), r)
b = -g; xref = np.linalg.solve(S, b)
def pcg(tol=1e-10, maxit=600):
    x = np.zeros(n); r = b.copy(); z = Minv(r); p = z.copy(); rz = r@z; rz0 = rz; its = 0
    while its < maxit and rz > tol*tol*rz0:
        q = S@p; al = rz/(p@q); x += al*p; r -= al*q; z = Minv(r); rzn = r@z; p = z + (rzn/rz)*p; rz = rzn; its += 1
    return x, its
x, its = pcg(); print("n", n, "plain PCG its", its)
def ecg(t, tol=1e-10, maxit=200, split="contig"):
    dom = (np.arange(n)*t)//n if split == "contig" else (np.arange(n)//6) % t
    R = np.zeros((n, t)); R[np.arange(n), dom] = b
    X = np.zeros((n, t)); Z = Minv(R); Pm = Z.copy(); its = 0
    r0 = np.sqrt(b@Minv(b))
    while its < maxit:
        Q = S@Pm
        Gm = Pm.T@Q
        w, V = np.linalg.eigh(Gm); keep = w > 1e-13*w.max(); Li = V[:, keep]/np.sqrt(w[keep])
        Pm = Pm@Li; Q = Q@Li
        al = Pm.T@R
        X += Pm@al; R -= Q@al
        rs_ = R.sum(1); Z = Minv(R); zs = Z.sum(1)
        its += 1
        if np.sqrt(abs(rs_@zs)) <= tol*r0: break
        be = -(Q.T@Z)
        Pm = Z + Pm@be
    x = X.sum(1)
    return x, its, keep.sum()
for t in (8, 16, 32, 64):
    for split in ("contig", "stride"):
        x, its, rank = ecg(t, split=split); print("ECG t", t, split, "its", its, "final rank", rank, "err %.2g" % (np.abs(x - xref).max()/np.abs(xref).max()))
