import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, time, sys
import scipy.linalg as sl
from textslam_amd import synth, abi
import oracle
n_kf = int(sys.argv[1]); far = float(sys.argv[2]); closures = int(sys.argv[3]); band = 8
P = synth.config_global(n_kf=n_kf, n_pt=14*n_kf, band=band, far_frac=far, closures=closures)
o = abi.options_global()
t = time.time(); rs = oracle.reduced_system(P, o, 0, o.initial_radius); print("oracle reduced system s", time.time() - t)
S, g = rs["S"].copy(), rs["g"].copy(); n = S.shape[0]
print("sym err", np.abs(S - S.T).max()/np.abs(S).max())
S = 0.5*(S + S.T)
print("n", n)
# band part
B = band
i, j = np.indices((n, n))
mask = np.abs(i//6 - j//6) <= B
M = np.where(mask, S, 0.0); E = S - M
print("nnz far entries", np.count_nonzero(E))
cM = sl.cho_factor(M)
Minv = lambda r: sl.cho_solve(cM, r)
b = -g
xref = np.linalg.solve(S, b)
def pcg(apply_pre, tol=1e-10, maxit=400):
    x = np.zeros(n); r = b.copy(); z = apply_pre(r); p = z.copy(); rz = r@z; rz0 = rz; its = 0
    while its < maxit and rz > tol*tol*rz0:
        q = S@p; al = rz/(p@q); x += al*p; r -= al*q; z = apply_pre(r); rzn = r@z; p = z + (rzn/rz)*p; rz = rzn; its += 1
    return x, its
x, its = pcg(Minv); print("plain PCG its", its, "err", np.abs(x - xref).max()/np.abs(xref).max())
for tol in (1e-6, 1e-8): print(" tol", tol, pcg(Minv, tol)[1])
# deflation / coarse space: piecewise constant per segment of m keyframes
nb = n//6
for m in (100, 50, 25):
    G = (nb + m - 1)//m
    Z = np.zeros((n, 6*G))
    for k in range(nb): Z[6*k:6*k+6, 6*(k//m):6*(k//m)+6] = np.eye(6)
    Ac = Z.T@S@Z; cA = sl.cho_factor(Ac)
    def pre(r, Z=Z, cA=cA): return Minv(r) + Z@sl.cho_solve(cA, Z.T@r)      # additive two-level
    print("additive coarse m", m, "dim", 6*G, "its", pcg(pre)[1])
    # piecewise linear hats
    Zl = np.zeros((n, 6*(G+1)))
    for k in range(nb):
        s = k/m; i0 = int(np.floor(s)); w = s - i0
        Zl[6*k:6*k+6, 6*i0:6*i0+6] = (1-w)*np.eye(6); Zl[6*k:6*k+6, 6*(i0+1):6*(i0+1)+6] = w*np.eye(6)
    Al = Zl.T@S@Zl; cAl = sl.cho_factor(Al + 1e-12*np.eye(Al.shape[0]))
    def prel(r, Z=Zl, cA=cAl): return Minv(r) + Z@sl.cho_solve(cA, Z.T@r)
    print("additive linear coarse m", m, "its", pcg(prel)[1])
# enlarged CG with t domains
def ecg(t, tol=1e-10, maxit=200):
    dom = (np.arange(n)*t)//n
    R = np.zeros((n, t)); R[np.arange(n), dom] = b
    X = np.zeros((n, t)); Z = Minv(R); Pm = Z.copy(); its = 0
    r0 = np.sqrt(b@Minv(b))
    while its < maxit:
        Q = S@Pm
        Gm = Pm.T@Q
        w, V = np.linalg.eigh(Gm); keep = w > 1e-14*w.max(); Li = V[:, keep]/np.sqrt(w[keep])
        Pm = Pm@Li; Q = Q@Li
        al = Pm.T@R
        X += Pm@al; R -= Q@al
        rs_ = R.sum(1); Z = Minv(R); zs = Z.sum(1)
        its += 1
        if np.sqrt(abs(rs_@zs)) <= tol*r0: break
        be = -(Q.T@Z)
        Pm = Z + Pm@be
    x = X.sum(1)
    return x, its
for t in (4, 8, 16, 32):
    x, its = ecg(t); print("enlarged CG t", t, "its", its, "err", np.abs(x - xref).max()/np.abs(xref).max())
# spectrum
w = sl.eigh(S, M, eigvals_only=True)
print("gen eig (S,M): min %.3g max %.3g; count >2: %d, >1.2: %d, <0.8: %d, <0.5: %d" % (w.min(), w.max(), (w>2).sum(), (w>1.2).sum(), (w<0.8).sum(), (w<0.5).sum()))
