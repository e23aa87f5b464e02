"""GPU diagnostic for the global-BA path (large reduced system -> multi-workgroup Cholesky)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
import oracle

def rel(a, b): return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))

opt = Optimizer(0)
for (nkf, npt) in ((40, 2000), (120, 8000)):
    P = synth.config_global(n_kf=nkf, n_pt=npt, band=8)
    o = abi.options_global()
    ro = oracle.reduced_system(P, o, 0, o.initial_radius)
    opt.upload(P, o)
    rg = opt.reduced_system(o.initial_radius)
    m = 6*ro['nf']
    print(nkf, "nf", ro['nf'], "S rel", rel(np.tril(rg['S'][:m,:m]), np.tril(ro['S'])) if False else "", "g rel", rel(rg['g'][:m], ro['g']), "cost", ro['cost'], rg['cost'])
    dp_ref = -np.linalg.solve(ro['S'], ro['g'])
    free = np.nonzero(rg['free'])[0]; idx = np.concatenate([np.arange(6*k, 6*k+6) for k in free])
    print("   dp rel diff vs numpy", rel(rg['dp'][idx], dp_ref))
    G = P.copy(); t = time.time(); rep = opt.GlobalBA(G, options=o); tg = time.time()-t
    R = P.copy(); t = time.time(); rep_o = oracle.solve(R, o); to = time.time()-t
    print("   gpu", rep['iters'], rep['accepted'], rep['termination'], rep['cost1'], "t_solve_ms", rep['t_solve_ms'], "upload", rep['t_upload_ms'])
    print("   ora", rep_o['iters'], rep_o['accepted'], rep_o['termination'], rep_o['cost1'], "t %.2fs" % to)
    print("   pose diff", np.abs(G.pose-R.pose).max(), "rho diff", np.abs(G.rho-R.rho).max(), flush=True)
# C5 size: timing only (oracle dense solve would take minutes)
P = synth.config_global(n_kf=500, n_pt=50000, band=12)
o = abi.options_global()
t = time.time(); opt.upload(P, o); print("C5 upload s", time.time()-t)
for _ in range(2):
    t = time.time(); rep = opt.solve(); print("C5 solve ms", (time.time()-t)*1e3, rep['iters'], rep['accepted'], rep['termination'], rep['cost0'], rep['cost1'], rep['n_sblock'], flush=True)
G = opt.download(P.copy())
print("C5 pose err before", np.abs(P.pose-P.truth['pose']).max(), "after", np.abs(G.pose-P.truth['pose']).max())
