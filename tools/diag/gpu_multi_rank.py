"""GPU diagnostic (not a pytest): ONE global BA sharded by landmark over N in-process ranks on one GPU (tsba_comm_init_local: the product's N > 1 path with the
collectives done in host memory), for rocprofv3 --kernel-trace --stats.  The ranks share the device, so wall time says nothing; the per-rank KERNEL time
(sum over all kernels / N / solves) is what a rank of an N-GPU node would spend computing, and with the 1-rank figure gives the ceiling of strong scaling
before any byte is exchanged.

    python tools/diag/gpu_multi_rank.py <world> <map> [solves]      map: dense500 = 5000 KF x 350 k points (~500 observations per keyframe)
                                                                          c5text   = C5 with its 1000 text planes (500 KF x 50 k points, use_text)
                                                                          c6       = the bench's 5000 KF x 70 k points
"""
import sys, os, json, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer, local_group_create, local_group_destroy

world, name = int(sys.argv[1]), sys.argv[2]
solves = int(sys.argv[3]) if len(sys.argv) > 3 else 2
o = abi.options_global()
if name == "dense500":
    P = synth.config_global(n_kf=5000, n_pt=350000, band=10)
elif name == "c5text":
    P = synth.make_problem(n_kf=500, n_pt=50000, n_text=1000, seed=7, feats=(64, 24, 12), band=12, n_levels=1, max_targets=8, text_targets=5, frozen_frac=0.0, rot_deg=0.2, trans_m=0.01)
    o.use_text = 1
else:
    P = synth.config_global(n_kf=5000, n_pt=70000, band=10)
group = local_group_create(world) if world > 1 else None
out = [None]*world


def run(rank):
    g = Optimizer(0)
    if world > 1:
        g.comm_init_local(group, rank, world)
    g.upload(P, o)
    rep = g.solve()                                  # warm
    t0 = time.perf_counter()
    for _ in range(solves):
        rep = g.solve()
    dt = (time.perf_counter() - t0)/solves
    info = g.solver_info(); ex = g.exchange_bytes()
    out[rank] = {"rank": rank, "world": world, "map": name, "n_pair": info["n_pair"], "n_sblock_plan": info["n_sblock"], "scene_candidates": info["n_scene_candidates"],
                 "point_slots": info["n_point_slots"], "exchange": ex, "iters": rep["iters"], "accepted": rep["accepted"], "cost1": rep["cost1"],
                 "blocks": [rep["n_sblock"][-1], rep["n_tblock"][-1]], "solve_ms_ranks_sharing_one_gpu": dt*1e3, "timed_solves": solves, "solves_total": solves + 1}
    g.close()


th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
for t in th: t.start()
for t in th: t.join(1200)
if group is not None:
    local_group_destroy(group)
for r in out:
    print(json.dumps(r))
