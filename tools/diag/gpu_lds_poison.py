"""LDS hygiene: every solve repeated with the LDS of all compute units filled with NaNs / 1e300 / 0x5a bytes before each launch (tsba_debug_options.lds_poison):
a kernel that reads LDS it has not written changes the result.    python tools/diag/gpu_lds_poison.py [case ...]"""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer

gpu = Optimizer(0)
CASES = {
    "c4_window": lambda: (synth.config_c4(), abi.options_local(), {}),
    "window_31kf": lambda: (synth.make_problem(n_kf=31, n_pt=1800, n_text=15, seed=71, feats=(16, 8, 6)), abi.options_local(), {}),
    "c3_pose": lambda: (synth.config_c3(), abi.options_pose(), {}),
    "open_chain": lambda: (synth.config_global(n_kf=5000, n_pt=70000, band=10), abi.options_global(), dict(its=6)),
    "open_chain_per_level": lambda: (synth.config_global(n_kf=5000, n_pt=70000, band=10), abi.options_global(), dict(its=6, dbg=dict(sv_per_level=15))),
    "long_range": lambda: (synth.config_global(n_kf=5000, n_pt=70000, band=10, far_frac=0.01), abi.options_global(), dict(its=4)),
    "ring": lambda: (synth.config_global(n_kf=5000, n_pt=70000, band=10, loop=True), abi.options_global(), dict(its=6)),
    "closures2": lambda: (synth.config_global(n_kf=5000, n_pt=70000, band=10, closures=2), abi.options_global(), dict(its=4)),
    "c5": lambda: (synth.config_global(n_kf=500, n_pt=50000, band=12), abi.options_global(), dict(its=6)),
    "mid_700kf": lambda: (synth.config_global(n_kf=700, n_pt=14000, band=8), abi.options_global(), dict(its=6)),
    "dense_100kf": lambda: (synth.config_global(n_kf=100, n_pt=3000, band=100), abi.options_global(), dict(its=6)),
}
names = sys.argv[1:] or list(CASES)
for name in names:
    P, o, kw = CASES[name]()
    if "its" in kw:
        o.its[0] = kw["its"]
    dbg = kw.get("dbg", {})
    out = {}
    for pz in (0, 1, 2, 3, 0):
        gpu.debug_set(lds_poison=pz, **dbg)
        gpu.upload(P, o); rep = gpu.solve(); G = gpu.download(P.copy())
        tr = [gpu.lm_trace(ps) for ps in range(o.n_passes)]
        key = (pz, len(out))
        out[key] = (rep, G, tr)
    ref = out[(0, 0)]
    line = []
    for (pz, _), (rep, G, tr) in out.items():
        same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(tr, ref[2])) and np.array_equal(G.pose, ref[1].pose) and np.array_equal(G.rho, ref[1].rho)
        d = ""
        if not same:
            for ps, (a, b) in enumerate(zip(tr, ref[2])):
                for k in range(min(len(a), len(b))):
                    if not np.array_equal(a[k], b[k], equal_nan=True):
                        d = f" [pass {ps} trial {k}: cost {a[k][0]:.10g} vs {b[k][0]:.10g}, verdict {a[k][3]} vs {b[k][3]}]"; break
                if d: break
        line.append(f"poison {pz}: {'same' if same else 'DIFFERENT' + d} (its {rep['iters']}, time-outs {rep['poll_timeouts']})")
    print(name, "|", " | ".join(line), flush=True)
gpu.debug_set()
