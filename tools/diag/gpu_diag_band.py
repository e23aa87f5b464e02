"""GPU diagnostic (not a pytest): the streaming band solver's dp against numpy on the reduced system of the first linearisation."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
opt = Optimizer(0)
cfgs = ((40, 6), (70, 9), (120, 10), (300, 12), (600, 10), (1500, 10)) if len(sys.argv) < 2 else tuple((int(a.split(',')[0]), int(a.split(',')[1])) for a in sys.argv[1:])
for (nkf, band) in cfgs:
    P = synth.config_global(n_kf=nkf, n_pt=50*nkf, band=band)
    o = abi.options_global()
    opt.upload(P, o)
    rg = opt.reduced_system(o.initial_radius)
    free = np.nonzero(rg['free'])[0]; idx = np.concatenate([np.arange(6*k, 6*k+6) for k in free])
    m = len(idx)
    if nkf <= 120:
        ro = oracle.reduced_system(P, o, 0, o.initial_radius); ref = -np.linalg.solve(ro['S'], ro['g'])
        print('   g rel', np.abs(rg['g'][:m] - ro['g']).max()/np.abs(ro['g']).max())
    else:
        S = rg['S'][:m, :m]; S = np.tril(S) + np.tril(S, -1).T       # S, g live in the compressed (free-pose) index space
        ref = -np.linalg.solve(S, rg['g'][:m])
    got = rg['dp'][idx]
    err = np.abs(got - ref)/np.abs(ref).max()
    bad = np.nonzero(err > 1e-8)[0]
    import ctypes as C
    st = (C.c_longlong*64)(); opt.lib.tsba_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]; print('   rc', opt.lib.tsba_debug_stamps(opt.ctx, st)); print('   dbg bw,CB,ntot,chunks,failbase,jb,n:', list(st[16:23]), np.array([st[23], st[24]], np.int64).view(np.float64))
    print(nkf, band, "nfree", len(free), "max rel err %.3e" % err.max(), "first bad row", (bad[0] if len(bad) else -1), "n bad", len(bad), flush=True)
    if len(bad):
        blocks = sorted(set((bad//6).tolist()))
        print("   bad blocks:", blocks[:20], "...", blocks[-5:])
