"""GPU diagnostic (not a pytest): the sharded global BA at world = N on ONE device through the in-process communicator (one thread per
rank): per-rank plan sizes, bytes handed to collectives per LM trial, and -- under `rocprofv3 --kernel-trace --stats` -- the kernel
durations of the rank-sized launches.  The collectives themselves go through host memory here, so their TIME is not representative of
RCCL over xGMI; everything else is what a rank executes.
usage: python tools/diag/gpu_multi_rank_profile.py WORLD [n_kf n_pt [far_frac]]"""
import sys, os, threading, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer, local_group_create, local_group_destroy
world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n_kf = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
n_pt = int(sys.argv[3]) if len(sys.argv) > 3 else 70000
far = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
P = synth.config_global(n_kf=n_kf, n_pt=n_pt, band=10, **({"far_frac": far} if far > 0 else {}))
o = abi.options_global()
out = [None]*world
def run(rank, group):
    g = Optimizer(0)
    if world > 1:
        g.comm_init_local(group, rank, world)
    g.upload(P, o)
    rep = g.solve()
    t = time.perf_counter(); rep = g.solve(); dt = time.perf_counter() - t
    out[rank] = {"rank": rank, "info": g.solver_info(), "exchange": g.exchange_bytes(), "iters": rep["iters"], "accepted": rep["accepted"],
                 "cost1": rep["cost1"], "n_sblock": rep["n_sblock"], "solve_ms_with_host_collectives": dt*1e3, "pcg": g.pcg_stats() if far > 0 else None}
    g.close()
group = local_group_create(world) if world > 1 else None
th = [threading.Thread(target=run, args=(r, group)) for r in range(world)]
for t in th: t.start()
for t in th: t.join()
if group is not None: local_group_destroy(group)
for r in out: print(json.dumps(r))
