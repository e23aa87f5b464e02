"""GPU diagnostic (not a pytest): the 5000-keyframe map with a growing share of long-range points (SURVEY 8d's C6 has 1 %): which solver the plan picks,
time per solve, conjugate-gradient iterations.  usage: python tools/diag/gpu_far_sweep.py [fractions...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
fr = [float(a) for a in sys.argv[1:]] or [0.0, 0.002, 0.005, 0.01, 0.02, 0.05]
g = Optimizer(0); o = abi.options_global()
print("far_frac | far blocks | band | ms per solve | LM its (accepted) | cg iterations (max per system, unconverged) | final / initial cost | solver")
for f in fr:
    P = synth.config_global(n_kf=5000, n_pt=70000, band=10, far_frac=f)
    g.upload(P, o); info = g.solver_info()
    ts = []
    for _ in range(3):
        rep = g.solve(); ts.append(rep["t_solve_ms"])
    print("%.3f | %6d | %2d | %7.2f | %d (%d) | %4d (%d, %d) | %.4f | %d" % (f, info["far_blocks"], info["far_band_blocks"], min(ts), rep["iters"][0], rep["accepted"][0],
          rep["pcg_iterations"], rep["pcg_max_iterations"], rep["pcg_unconverged"], rep["cost1"][0]/rep["cost0"][0], rep["solver_path"]), flush=True)
