import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
import oracle
opt = Optimizer(0)
P = synth.config_global(n_kf=40, n_pt=2000, band=8)
o = abi.options_global()
ro = oracle.reduced_system(P, o, 0, o.initial_radius)
opt.upload(P, o)
rg = opt.reduced_system(o.initial_radius)
m = 6*ro['nf']
Lref = np.linalg.cholesky(ro['S'])
Lg = np.tril(rg['S'][:m,:m])
print("m", m)
for (a,b) in ((0,96),(96,192),(192,m)):
    for (c,d) in ((0,96),(96,192),(192,m)):
        if c > a: continue
        blk = np.abs(Lg[a:b,c:d]-Lref[a:b,c:d]).max()/np.abs(Lref).max()
        print("block rows %d-%d cols %d-%d  rel err %.3e" % (a,b,c,d,blk))
yref = np.linalg.solve(Lref, ro['g'])
import ctypes
# row n of S = y: fetch via S array (n+1 rows are not returned by the debug call) -> skip

N = 6*P.n_kf
buf = np.zeros((N+1)*N)
opt.lib.tsba_debug_copy_S(opt.ctx, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
S2 = buf.reshape(N+1, N)
xg = S2[m, :m]          # after back substitution row n holds x
if os.environ.get('TSBA_DEBUG_NO_BACKSUB'):
    yref = np.linalg.solve(Lref, ro['g'])
    print('y rel err', np.abs(xg-yref).max()/np.abs(yref).max())
    for (a,b) in ((0,96),(96,192),(192,m)):
        print('  y block %d-%d rel err %.3e' % (a,b,np.abs(xg[a:b]-yref[a:b]).max()/np.abs(yref).max()))
    print(xg[:8], yref[:8], ro['g'][:8])
    sys.exit(0)
xref = np.linalg.solve(ro['S'], ro['g'])
print("x rel err", np.abs(xg - xref).max()/np.abs(xref).max())
for (a,b) in ((0,96),(96,192),(192,m)):
    print("  x block %d-%d rel err %.3e" % (a,b,np.abs(xg[a:b]-xref[a:b]).max()/np.abs(xref).max()))
free = np.nonzero(rg['free'])[0]; idx = np.concatenate([np.arange(6*k, 6*k+6) for k in free])
print("dp vs -x", np.abs(rg['dp'][idx] + xg).max(), "dp[:12]", rg['dp'][:24])
print("free", free[:10], rg['free'][:10])
