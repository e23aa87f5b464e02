"""GPU diagnostic (not a pytest): C6 resident solve A/B over tsba_debug_set switches (name=value pairs on the command line)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
g = Optimizer(0)
P = synth.config_global(n_kf=5000, n_pt=70000, band=10); o = abi.options_global()
variants = [("default", {})] + [(a, {a.split("=")[0]: int(a.split("=")[1])}) for a in sys.argv[1:]] + [("default", {})]
for name, kw in variants:
    g.debug_set(**kw); g.upload(P, o)
    g.solve(); t = time.perf_counter(); n = 5
    for _ in range(n): rep = g.solve()
    print("%-20s solve %.2f ms  its %s acc %s cost1 %s" % (name, (time.perf_counter() - t)/n*1e3, rep["iters"], rep["accepted"], rep["cost1"]), flush=True)
