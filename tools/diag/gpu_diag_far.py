"""GPU diagnostic (not a pytest): global BA on a map with scattered long-range observations (SURVEY 8d's C6: band co-visibility + 1 % far
points) and / or several loop closures -- which solver path the upload picks, device memory, ms per resident solve.
usage: gpu_diag_far.py [n_kf] [n_pt] [far_frac] [solves] [closures]"""
import sys, os, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
nkf = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
npt = int(sys.argv[2]) if len(sys.argv) > 2 else 70000
far = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
nsolve = int(sys.argv[4]) if len(sys.argv) > 4 else 2
closures = int(sys.argv[5]) if len(sys.argv) > 5 else 0
tol_exp = int(sys.argv[6]) if len(sys.argv) > 6 else 0


def mem_used_mb():
    try:
        out = subprocess.run(["rocm-smi", "--showmeminfo", "vram", "--csv"], capture_output=True, text=True, timeout=20).stdout
        for line in out.splitlines():
            f = line.split(",")
            if len(f) >= 3 and f[0].startswith("card"):
                return float(f[2])/2**20
    except Exception:
        pass
    return float("nan")


opt = Optimizer(0)
if os.environ.get("PCG_REFACTOR"):          # 1: the factorisation re-run as preconditioner, 2: the many-column solve phase (default: the single-vector solve phase)
    opt.debug_set(pcg_refactor=int(os.environ["PCG_REFACTOR"])); print("pcg_refactor", os.environ["PCG_REFACTOR"])
if os.environ.get("FAR_SOLVER"):            # 3: conjugate gradients without the low-rank correction
    opt.debug_set(far_solver=int(os.environ["FAR_SOLVER"])); print("far_solver", os.environ["FAR_SOLVER"])
if tol_exp:
    opt.debug_set(pcg_tol_exp=tol_exp); print("pcg tolerance 1e-%d" % tol_exp)
t = time.time()
kw = dict(n_kf=nkf, n_pt=npt, band=10, far_frac=far)
if closures:
    kw["closures"] = closures
P = synth.config_global(**kw); print("synth s %.1f" % (time.time() - t), "observations", len(P.sobs_kf[0]), flush=True)
o = abi.options_global()
m0 = mem_used_mb()
t = time.time(); opt.upload(P, o); print("upload s %.3f" % (time.time() - t), flush=True)
info = opt.solver_info()
print("solver_info", info, flush=True)
print("device memory MB: before upload %.0f, after %.0f" % (m0, mem_used_mb()), flush=True)
for _ in range(nsolve):
    t = time.time(); rep = opt.solve()
    print("solve ms %.1f" % ((time.time() - t)*1e3), "iters", rep['iters'], "accepted", rep['accepted'], "term", rep['termination'], "cost", rep['cost0'], rep['cost1'],
          "blocks", rep['n_sblock'], "evals", rep['n_resid_evals'], "pcg", opt.pcg_stats(), flush=True)
