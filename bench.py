#!/usr/bin/env python3
"""bench.py -- the two north-star measurements of BASELINE.json, one JSON line on rank 0.

  N = 1 (default)   local BA: wall time and residuals/s of one complete optimizer::LocalBundleAdjustment (pyramid passes
                    2,1,0 x <= 10 LM iterations, mu/sigma, outlier passes) on the 20 KF x 5000 pts x 100 text-plane window
                    (SURVEY.md 8d, config C4), the window resident in HBM when the timed region starts.
  N > 1 (default)   global BA: residuals/s of one complete optimizer::GlobalBA on the synthetic 5000-KF / ~500 k-observation map
                    (config C6) sharded over the N GPUs (strong scaling: the map is fixed); the line carries the 1-GPU time of
                    the same map measured in the same run (`scaling_reference`), the RCCL rank count and the bytes every rank
                    exchanges per LM trial.
  --workload local_ba | global_ba | orb selects explicitly (local BA and ORB at N > 1 are independent replicas).

residuals/s (`value`, round 6) = SURVEY 8d's count / wall time; `residuals_per_s_all_evaluations` = scalar residuals evaluated (once per linearisation and once per LM trial step); `residuals_per_s_8d` (= value) counts
SURVEY 8d's way (one residual + Jacobian evaluation per LM trial).  At N = 1 the local-BA line also carries `also`: the other BASELINE configs
(C3, C5, C6 as an open chain / with 1 % long-range observations / after a loop closure, the ORB batch), each a few solves, no CPU leg.

Launch: `python bench.py --gpus N` spawns the N ranks itself (one process per GPU, RCCL over xGMI); under a launcher that already
set RANK / LOCAL_RANK / WORLD_SIZE (torch.distributed.run) it is one of the ranks -- --gpus must then agree with WORLD_SIZE.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s
F64_MFMA_PEAK_TFLOPS = 78.6      # dense fp64 matrix rate (= the fp64 vector rate on this part)


def _pmc_traffic(name):
    """HBM bytes per launch of a kernel from the PMC passes committed under profiles/ (counters cannot be read from inside the
    process): 2 x FETCH_SIZE (gfx950 correction of the micro-architecture guide) + WRITE_SIZE."""
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        try:
            with open(os.path.join(ROOT, "profiles", "%s_%s_pmc_traffic.json" % (rnd, name))) as f:
                pm = json.load(f)
            return (2.0*pm["fetch_size_kb_raw_max"] + pm["write_size_kb_raw_max"])*1024.0, pm["source"]
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def usable_cores():
    """CPUs this process can actually run on at once: the online CPUs, cut by the affinity mask and by the cgroup's CPU quota (the GPU
    boxes of this pool show 256 CPUs under a quota of 16: an OpenMP team of 256 there is 16 CPUs' worth of time slices)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(float(quota)/float(period) + 0.5)))
        except (OSError, ValueError):
            pass
    try:                                                  # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            n = min(n, max(1, int(q/per + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline_local(prob, opt_ref):
    """The CPU restatement of the reference path (oracle source, Ceres-style central-difference text Jacobians) timed on this host:
    the full LocalBundleAdjustment call on the same window, once with the reference's own setting (1 thread: num_threads = 1,
    optimizer.cc:1600) and once with every host core (OpenMP over the residual blocks).  Built here with -O3 -march=native."""
    import oracle
    from textslam_amd import abi
    L = oracle.baseline_lib()
    o = abi.TsbaOptions.from_buffer_copy(opt_ref)
    o.text_jacobian = 1                       # NumericDiffCostFunction<CENTRAL>, nume_BAText.h:97-100
    cores = usable_cores()
    res = {}
    for nt in sorted({1, min(16, cores), cores}):      # the restatement's accumulation is serial: a moderate team often beats all cores
        oracle.omp_set_threads(nt)
        q = prob.copy()
        t0 = time.perf_counter(); rep = oracle.solve(q, o, library=L); dt = time.perf_counter() - t0
        res[nt] = (_evals_8d(rep)/dt, dt)
    best = max(res, key=lambda k: res[k][0])
    return {"value": res[best][0], "unit": "residuals/s", "cores": best, "kind": "port", "counting": COUNTING,
            "sample": "the full LocalBundleAdjustment call (3 passes) on the same window, numeric-diff text Jacobians, gcc -O3 -march=native -fopenmp; "
                      "value = the best of the thread counts tried (1 = the reference's own num_threads setting, 16, all %d CPUs this process may use: "
                      "%d online, affinity mask and cgroup quota applied)" % (cores, os.cpu_count() or 1),
            "seconds": res[best][1], "single_thread_value": res[1][0], "single_thread_seconds": res[1][1],
            "all_core_value": res[cores][0], "all_cores": cores, "by_threads": {str(k): v[0] for k, v in res.items()}}


def cpu_baseline_global(prob, opt):
    """The same 5000-keyframe map through the CPU restatement (band storage of H_pp / S + band Cholesky), 1 thread and all cores."""
    import oracle
    L = oracle.baseline_lib()
    cores = usable_cores()
    res = {}
    for nt in sorted({1, cores}):
        oracle.omp_set_threads(nt)
        t0 = time.perf_counter(); rep = oracle.solve(prob.copy(), opt, library=L); dt = time.perf_counter() - t0
        res[nt] = (_evals_8d(rep)/dt, dt)
    best = max(res, key=lambda k: res[k][0])
    return {"value": res[best][0], "unit": "residuals/s", "cores": best, "kind": "port", "counting": COUNTING,
            "sample": "the complete GlobalBA call (20 LM iterations) on the SAME map; OpenMP parallelises the block evaluation only -- the "
                      "normal-equation accumulation and the Schur complement of the restatement are serial, so all cores gain little",
            "seconds": res[best][1], "single_thread_value": res[1][0], "single_thread_seconds": res[1][1],
            "all_core_value": res[cores][0], "all_cores": cores}


def bench_orb(args, rank, local_rank, world, dist, torch):
    """BASELINE config 2: ORBextractor (1000, 1.2, 8, 20, 7) on a batch of 64 frames 640x480 per GPU (replicas for N > 1)."""
    import numpy as np
    from textslam_amd.orbextractor import ORBextractor, synthetic_frame
    imgs = np.stack([synthetic_frame(1000*rank + s) for s in range(64)])
    ex = ORBextractor(1000, 1.2, 8, 20, 7, device=local_rank)
    ex.upload(imgs)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier(); torch.cuda.synchronize()
    for _ in range(args.warmup):
        ex.run()
    sync(); t0 = time.perf_counter()
    for _ in range(args.steps):
        ex.run()
    sync(); dt = time.perf_counter() - t0
    nk = sum(len(k) for k, _ in ex.download())
    if dist is not None:
        t = torch.tensor([dt, float(nk)], dtype=torch.float64, device="cuda")
        tm = t.clone(); dist.all_reduce(tm, op=dist.ReduceOp.MAX); dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, nk = float(tm[0]), float(t[1])
    if rank == 0:
        out = {"metric": "orb_keypoints_per_s", "value": nk*args.steps/dt, "unit": "keypoints/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": dt/args.steps*1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "u8", "data": "synthetic",
               "config": {"workload": "ORBextractor(1000,1.2,8,20,7), 64 frames 640x480 per GPU", "frames_per_s": 64*world*args.steps/dt}}
        # whole pipeline against HBM: SURVEY 8d counts 4.7 MB of algorithmic traffic per frame (input + pyramid + blur planes + taps)
        algo = 64*4.7e6
        traffic, src = _pmc_traffic("orb")
        out["roofline"] = {"bound": "hbm", "kernel": "whole ORB pipeline (pyramid, FAST, quadtree, orientation, blur, rBRIEF)", "achieved": algo*world/(dt/args.steps)/1e9,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algo/(dt/args.steps)/1e9/HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
                           "algorithmic_bytes_per_launch": algo}
        if not args.no_cpu_baseline and world == 1:
            import oracle
            t0 = time.perf_counter(); n = 0
            for f in range(16):
                n += len(oracle.orb_extract(imgs[f])[0])
            dtc = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": n/dtc, "unit": "keypoints/s", "cores": 1, "kind": "port", "sample": "16 of the 64 frames", "seconds": dtc}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


def spawn_ranks(args):
    """`python bench.py --gpus N` outside a launcher: one child process per GPU; rank 0's JSON line is passed through."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out0); sys.stdout.flush()
    if any(rcs):
        sys.exit("bench.py: rank exit codes %s" % rcs)


SOLVER_PATHS = {0: "LDS (one workgroup)", 1: "dense / wide-band Cholesky", 2: "streaming band", 3: "partitioned band", 4: "partitioned band + cyclic reduction",
                5: "ring (ghost rows)", 6: "band + low-rank correction", 7: "band-preconditioned conjugate gradients", 8: "pose-only (6x6)"}      # include/tsba.h TSBA_SOLVER_*


COUNTING = ("SURVEY 8d: scalar residuals of one residual + Jacobian evaluation per LM trial step (2 per scene block, 8 per text block, x LM iterations); the cost "
            "evaluation that the same speculative launch produces for every trial is NOT counted (with it: residuals_per_s_all_evaluations, ~1.9 x)")


def _evals_8d(rep):
    """SURVEY 8d's residual count: every residual evaluated in a residual + Jacobian pass, once per LM trial step (the library's
    n_resid_evals also counts the cost evaluation of every trial, which the same speculative launch produces: ~1.9 x)."""
    return float(sum(it*(2*ns + 8*nt) for it, ns, nt in zip(rep["iters"], rep["n_sblock"], rep["n_tblock"])))


def also_lines(gpu, local_rank, torch, steps=10):
    """The other BASELINE configs on this GPU, without their CPU legs (a few hundred ms of GPU time each; the synthetic maps take longer to
    build than to solve): C3 pose-only, C5 global BA 500 KF x 50 k points, C6 global BA 5000 KF / ~500 k observations as an open chain, with
    SURVEY 8d's 1 % long-range observations, right after a loop closure (ring) and with two separate closures, and the ORB batch of 64 frames."""
    import numpy as np
    from textslam_amd import synth, abi
    out = {}

    def run(name, prob, opt, extra=None, call=None):
        t0 = time.perf_counter(); gpu.upload(prob, opt); up = (time.perf_counter() - t0)*1e3
        rep = gpu.solve(); torch.cuda.synchronize()
        ts = []                                            # every solve timed on its own (tsba_solve returns after its last kernel): median, min and max of `steps`
        for _ in range(steps):
            t0 = time.perf_counter(); rep = gpu.solve(); ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize(); dt = float(np.median(ts))
        e = {"ms_per_solve": dt*1e3, "ms_per_solve_min": min(ts)*1e3, "ms_per_solve_max": max(ts)*1e3, "timed_solves": steps, "poll_timeouts": rep["poll_timeouts"],
             "residuals_per_s": _evals_8d(rep)/dt, "residuals_per_s_8d": _evals_8d(rep)/dt, "residuals_per_s_all_evaluations": float(rep["n_resid_evals"])/dt, "lm_iterations": rep["iters"],
             "accepted": rep["accepted"], "scene_blocks": rep["n_sblock"][-1], "text_blocks": rep["n_tblock"][-1], "upload_plan_ms": up, "cost": [rep["cost0"][0], rep["cost1"][-1]],
             "solver_path": SOLVER_PATHS.get(rep["solver_path"], rep["solver_path"])}
        if extra:
            e.update(extra(rep))
        if call:                                           # the one-shot entry point the reference's caller uses (plan on the host + upload + solve + download), context warm
            cold, lib_ms = [], []
            for _ in range(4):
                q = prob.copy(); t0 = time.perf_counter(); r1 = getattr(gpu, call)(q, options=opt); cold.append((time.perf_counter() - t0)*1e3)
                lib_ms.append(r1["t_upload_ms"] + r1["t_solve_ms"] + r1["t_download_ms"])
            e["cold_call_ms"] = min(cold[1:])              # through the Python mirror (ctypes struct + report conversion: ~0.09 ms of it)
            e["cold_call_library_ms"] = min(lib_ms[1:])    # inside the C ABI call: upload + solve + download as the library's own clock sees them
        out[name] = e

    def glob_extra(rep):
        info = gpu.solver_info()
        lin_ms, algo = gpu.time_linearize(0, 30)
        # (PMC passes: the 5000-keyframe chain, and -- round 6 -- C5)
        traffic, src = _pmc_traffic("c6_linearize") if gpu._resident.n_kf == 5000 else (_pmc_traffic("c5_linearize") if gpu._resident.n_kf == 500 else (None, None))
        e = {"band_rows": info["band_rows"], "interiors": info["interiors"], "ring_partition": info["ring"], "keyframes_reordered": info["kf_reordered"],
             "roofline": {"bound": "hbm", "kernel": "k_linearize (level 0)", "achieved": algo/(lin_ms*1e-3)/1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": algo/(lin_ms*1e-3)/1e9/HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src, "algorithmic_bytes_per_launch": algo, "avg_launch_us": lin_ms*1e3}}
        if info["far_band_blocks"]:
            e.update({"long_range_blocks": info["far_blocks"], "preconditioner_band_blocks": info["far_band_blocks"], "pcg_iterations": rep["pcg_iterations"],
                      "pcg_systems": rep["pcg_systems"], "pcg_max_iterations": rep["pcg_max_iterations"], "pcg_unconverged": rep["pcg_unconverged"],
                      "pcg_stagnated": rep["pcg_stagnated"]})
        else:
            e["solve_us_per_lm_trial"] = gpu.time_solve(10)*1e3
        return e
    run("c3_pose_only", synth.config_c3(), abi.options_pose(), call="PoseOptim")
    og = abi.options_global()
    run("c5_global_500kf_50kpts", synth.config_global(n_kf=500, n_pt=50000, band=12), og, glob_extra, call="GlobalBA")
    run("c6_global_5000kf_open_chain", synth.config_global(n_kf=5000, n_pt=70000, band=10), og, glob_extra, call="GlobalBA")
    run("c6_global_5000kf_1pct_long_range", synth.config_global(n_kf=5000, n_pt=70000, band=10, far_frac=0.01), og, glob_extra, call="GlobalBA")
    run("c6_global_5000kf_loop_closure_ring", synth.config_global(n_kf=5000, n_pt=70000, band=10, loop=True), og, glob_extra, call="GlobalBA")
    run("c6_global_5000kf_two_loop_closures", synth.config_global(n_kf=5000, n_pt=70000, band=10, closures=2), og, glob_extra, call="GlobalBA")
    # ORB batch of 64 frames (BASELINE config 2)
    from textslam_amd.orbextractor import ORBextractor, synthetic_frame
    imgs = np.stack([synthetic_frame(s) for s in range(64)])
    ex = ORBextractor(1000, 1.2, 8, 20, 7, device=local_rank)
    ex.upload(imgs)
    for _ in range(5):
        ex.run()
    ts = []                                                # (tsorb_run returns after its last kernel: every batch timed on its own, median of 20 as the other entries)
    for _ in range(20):
        t0 = time.perf_counter(); ex.run(); ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts))
    nk = sum(len(k) for k, _ in ex.download())
    out["c2_orb_64_frames"] = {"ms_per_batch": dt*1e3, "ms_per_batch_min": min(ts)*1e3, "ms_per_batch_max": max(ts)*1e3, "timed_batches": 20, "keypoints_per_s": nk/dt, "frames_per_s": 64/dt,
                               "roofline": {"bound": "hbm", "kernel": "whole ORB pipeline", "achieved": 64*4.7e6/dt/1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": 64*4.7e6/dt/1e9/HBM_PEAK_GBS, "algorithmic_bytes_per_launch": 64*4.7e6}}
    # ORB, ONE frame: the call frame.cc:328-331 makes per camera image (resident: the device chain alone; call: upload + chain + download)
    ex.upload(imgs[:1])
    for _ in range(5):
        ex.run()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); ex.run(); ts.append(time.perf_counter() - t0)
    tc = []
    for _ in range(20):
        t0 = time.perf_counter(); ex.extract_batch(imgs[:1]); tc.append(time.perf_counter() - t0)
    out["orb_one_frame"] = {"ms_resident": float(np.median(ts))*1e3, "ms_resident_min": min(ts)*1e3, "ms_resident_max": max(ts)*1e3,
                            "ms_call": float(np.median(tc))*1e3, "ms_call_min": min(tc)*1e3, "ms_call_max": max(tc)*1e3, "timed": 20, "keypoints": int(len(ex.download()[0][0]))}
    # the tracking thread's window search (tracking::SearchFrom3D*: frame::GetFeaturesInArea + DescriptorDistance): every keypoint of one frame searched in the next, 40-px radius
    (kpA, dA), (kpB, dB) = ex.extract_batch(imgs[:2])
    nq = kpA.shape[0]
    qxy = kpA[:, :2].copy(); qr = np.full(nq, 40.0, np.float32); oct_ = kpA[:, 5].astype(np.int32); qlev = np.stack([oct_ - 1, oct_ + 1], 1).astype(np.int32)
    tg, tq = [], []
    for _ in range(23):
        t0 = time.perf_counter(); ex.match_set_frame(1, (0.0, 640.0, 0.0, 480.0)); t1 = time.perf_counter(); res = ex.match_search(qxy, qr, qlev, dA, 32); t2 = time.perf_counter()
        tg.append(t1 - t0); tq.append(t2 - t1)
    out["orb_window_search"] = {"ms_search_call": float(np.median(tq[3:]))*1e3, "ms_search_call_min": min(tq[3:])*1e3, "ms_search_call_max": max(tq[3:])*1e3,
                                "ms_grid_per_frame": float(np.median(tg[3:]))*1e3, "queries": int(nq), "radius_px": 40, "mean_candidates": float(res["cand_cnt"].mean()), "timed": 20}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=["local_ba", "global_ba", "orb"],
                    help="default: local_ba at 1 GPU, global_ba (the sharded 5000-keyframe map) at N > 1")
    ap.add_argument("--kf", type=int, default=5000, help="global_ba: keyframes (BASELINE configs[4]: 5k KF / 500k observations)")
    ap.add_argument("--pts", type=int, default=70000, help="global_ba: map points (70k points -> ~500k observations)")
    ap.add_argument("--band", type=int, default=10, help="global_ba: keyframes after its host that observe a point (C5 in `also`: 12)")
    ap.add_argument("--far", type=float, default=0.0, help="global_ba: fraction of loop-closure-like long-range observations")
    ap.add_argument("--loop", action="store_true", help="global_ba: closed trajectory (the map right after a loop closure: ring-shaped co-visibility)")
    ap.add_argument("--loop-at", type=int, default=0, help="global_ba --loop: the loop starts at this keyframe (a tail before the loop)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="local_ba at 1 GPU: skip the `also` object (the other BASELINE configs: C3, C5, C6 incl. long-range / loop-closure maps, ORB)")
    ap.add_argument("--check-single", action="store_true", help="global_ba, N > 1: compare the N-rank result with the 1-rank solve")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            return spawn_ranks(args)
        rank, local_rank, world = 0, 0, 1
    else:
        rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            sys.exit("bench.py: --gpus %d disagrees with WORLD_SIZE=%d" % (args.gpus, world))
    workload = args.workload or ("local_ba" if world == 1 else "global_ba")
    dist = None
    import torch
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import numpy as np
    from textslam_amd import synth, abi
    from textslam_amd.optimizer import Optimizer

    if workload == "orb":
        return bench_orb(args, rank, local_rank, world, dist, torch)
    gpu = Optimizer(local_rank)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(opt_ctx, steps, warmup, barrier=True):
        rep = None
        for _ in range(warmup):
            rep = opt_ctx.solve()
        if barrier:
            sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            rep = opt_ctx.solve()                          # synchronous: returns after the last kernel
        if barrier:
            sync()
        return rep, time.perf_counter() - t0

    reference, check = None, None
    if workload == "global_ba":
        # one global BA sharded by landmark over the ranks: poses replicated, reduced normal equations exchanged over RCCL
        prob = synth.config_global(n_kf=args.kf, n_pt=args.pts, band=args.band, far_frac=args.far, loop=args.loop, loop_at=args.loop_at)      # identical on every rank
        opt = abi.options_global()
        if world > 1:
            if rank == 0:                                  # the 1-GPU time of the SAME map, same run: the strong-scaling reference
                gpu.upload(prob, opt)
                rep1, dt1 = timed(gpu, args.steps, args.warmup, barrier=False)
                reference = {"n_gpus": 1, "ms_per_step": dt1/args.steps*1e3, "value": _evals_8d(rep1)*args.steps/dt1}
                if args.check_single:
                    check = gpu.download(prob.copy())
            idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(gpu.comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, 0)
            gpu.comm_init(bytes(idt.cpu().numpy().tobytes()), rank, world)
    else:
        prob = synth.config_c4(seed=synth.SEED + rank)    # every replica gets its own window
        opt = abi.options_local()
    gpu.upload(prob, opt)
    rep, dt = timed(gpu, args.steps, args.warmup)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if workload == "global_ba":
            evals_all = float(rep["n_resid_evals"])        # block counts are already global (all-reduced in the library)
        else:
            ev = torch.tensor([float(rep["n_resid_evals"])], dtype=torch.float64, device="cuda")
            dist.all_reduce(ev, op=dist.ReduceOp.SUM)
            evals_all = float(ev.item())
    else:
        evals_all = float(rep["n_resid_evals"])
    ms_per_step = dt/args.steps*1e3
    # `value` = SURVEY 8d's residuals/s (round 6; rounds 1-5 reported the library's count of ALL evaluations there, ~1.9 x, and this one as residuals_per_s_8d).
    # Global BA: the block counts are global in every rank's report; local BA at N > 1: independent replicas of the same window shape
    value = _evals_8d(rep)*(1 if workload == "global_ba" else world)*args.steps/dt
    value_all = evals_all*args.steps/dt
    info = gpu.solver_info()

    out = None
    if workload == "global_ba":
        # every rank takes part in the timed launches of the linearisation (the pass set-up all-reduces in a sharded run)
        lin_ms, algo_bytes = gpu.time_linearize(0, 50)
        sol_ms = gpu.time_solve(20) if world == 1 else None
        ex = gpu.exchange_bytes()
        if rank == 0:
            n6 = 6*args.kf
            out = {"metric": "global_ba_residuals_per_s", "value": value, "unit": "residuals/s", "n_gpus": world, "steps": args.steps,
                   "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                   "dtype": "f64", "data": "synthetic", "residuals_per_s_8d": value, "residuals_per_s_all_evaluations": value_all, "counting": COUNTING,
                   "config": {"workload": "C6 global BA: %d KF x %d pts (scene only, 20 LM its, level 0), landmarks sharded over %d GPU(s)"
                                          % (args.kf, args.pts, world), "lm_iterations": rep["iters"], "scene_blocks": rep["n_sblock"],
                              "reduced_system_dim": n6, "band_rows": info["band_rows"], "interiors": info["interiors"],
                              "separator_solver": "cyclic reduction" if info["sep_cr"] else "streaming", "far_frac": args.far,
                              "loop_closure_map": bool(args.loop), "keyframes_reordered": bool(info["kf_reordered"]), "ring_partition": bool(info.get("ring", 0)),
                              "long_range_blocks": info.get("far_blocks", 0), "pcg": gpu.pcg_stats() if info.get("far_band_blocks") else None,
                              "rccl_ranks": ex["ranks"], "allreduce_bytes_per_lm_trial": ex["per_trial"],
                              "allreduce_bytes_per_linearisation": ex["per_linearisation"]}}
            out["rccl_ranks"] = ex["ranks"]                     # (top-level: the communicator size the library reports, 1 without tsba_comm_init)
            if reference:
                out["scaling_reference"] = reference
                out["config"]["speedup_vs_1gpu_same_run"] = reference["ms_per_step"]/ms_per_step
            if world > 1:
                out["scaling_note"] = ("strong scaling of ONE 5000-keyframe solve is expected to be flat: the reduced-system solve (two thirds of an LM trial on one GPU, "
                                       "a latency chain that every rank repeats) does not shard; per-rank kernel times give a ceiling of 1.42 x at 8 ranks before "
                                       "16.6 MB of all-reduce per trial (DESIGN.md 7).  INTEGRATION.md tells callers to keep a map on one GPU")
            traffic, src = _pmc_traffic("c6_linearize") if args.kf == 5000 else (_pmc_traffic("c5_linearize") if args.kf == 500 else (None, None))
            achieved = algo_bytes/(lin_ms*1e-3)/1e9
            out["roofline"] = {"bound": "hbm", "kernel": "k_linearize<FULL,4> (level 0, this rank's shard)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": achieved/HBM_PEAK_GBS, "traffic": traffic if world == 1 else None, "traffic_source": src if world == 1 else None,
                               "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_us": lin_ms*1e3}
            if sol_ms is not None and not args.loop and not info.get("far_band_blocks"):
                # the dominant phase of an LM trial: the band solve of the reduced camera system -- LDL^T of an n x n band of half width
                # bw: n (bw^2 + 3 bw) flops + two triangular sweeps 4 n bw
                bw = info["band_rows"] + 5
                flops = n6*(bw*bw + 3.0*bw) + 4.0*n6*bw
                out["roofline_dominant"] = {"bound": "mfma", "kernel": "reduced-system solve (k_bandp_factor + separator solve + back-substitution)",
                                            "achieved": flops/(sol_ms*1e-3)/1e12, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                            "frac": flops/(sol_ms*1e-3)/1e12/F64_MFMA_PEAK_TFLOPS, "flops_per_launch": flops, "avg_launch_us": sol_ms*1e3,
                                            "note": "latency-bound chain (interior factorisation + log2(P) cyclic-reduction levels), not a throughput kernel"}
            if check is not None:
                got = gpu.download(prob.copy())
                out["config"]["max_pose_diff_vs_single_rank"] = float(np.abs(got.pose - check.pose).max())
            if not args.no_cpu_baseline and world == 1:
                out["cpu_baseline"] = cpu_baseline_global(prob, opt)
    elif rank == 0:
        # the same window through the one-shot ABI entry point (what the TextSLAM adapter calls per keyframe)
        cold = []
        for _ in range(3):
            g2 = prob.copy(); tc = time.perf_counter(); gpu.LocalBundleAdjustment(g2, options=opt); cold.append((time.perf_counter() - tc)*1e3)
        cold_ms = min(cold)
        # the call TextSLAM makes once per new keyframe: the window slides by one (tracking.cc:828-842), 19 of its 20 keyframes were in the last call --
        # with the keyframes' identities (tsba_problem.kf_id) the context keeps their pyramid planes on the device and copies the new one only
        import numpy as np
        slide = []
        for it in range(4):
            g2 = prob.copy(); g2.kf_id = np.arange(20, dtype=np.int64) + 1000; g2.kf_id[19] = 5000 + it
            tc = time.perf_counter(); gpu.LocalBundleAdjustment(g2, options=opt); slide.append((time.perf_counter() - tc)*1e3)
        slide_ms = min(slide[1:])
        # end to end as the adapter sees it: object graph -> gather (adapter/tsba_gather.hpp) -> tsba_local_ba -> scatter, from C++ (tests/cxx/abi_from_cxx)
        adapter = None
        exe = os.path.join(ROOT, "tests", "cxx", "abi_from_cxx")
        if os.path.exists(exe):
            import tempfile
            from textslam_amd import abi as _abi
            with tempfile.TemporaryDirectory() as td:
                _abi.write_dump(os.path.join(td, "c4.bin"), prob.copy())
                r = subprocess.run([exe, os.path.join(td, "c4.bin"), "time_local", os.path.join(td, "t.json"), "5"], capture_output=True, text=True, timeout=300)
                if r.returncode == 0:
                    adapter = json.load(open(os.path.join(td, "t.json")))
        gpu.upload(prob, opt)
        gpu.solve()
        # roofline of the linearisation kernel (residual + Jacobian + IRLS weight + J^T J / J^T r sums), level 0
        lin_ms, algo_bytes = gpu.time_linearize(0, 200)
        achieved = algo_bytes/(lin_ms*1e-3)/1e9
        traffic, traffic_src = _pmc_traffic("c4")
        gpu.solve()
        sol_ms = gpu.time_solve(200)
        n = 6*(prob.n_kf - 3)                                      # 17 free poses: the first three keyframes of the window are the gauge
        flops = n**3/3.0 + 4.0*n*n                                 # LDL^T + forward / backward substitution
        out = {
            "metric": "local_ba_residuals_per_s", "value": value, "unit": "residuals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C4 local BA: 20 KF x 5000 pts x 100 text planes (%d (KF,plane) pairs x 64 features), "
                                   "passes 2,1,0 x <=10 LM its" % prob.n_tobs,
                       "parallelism": "replicas x%d" % world if world > 1 else "1 GPU",
                       "residual_blocks_level0": {"scene": rep["n_sblock"][-1], "text": rep["n_tblock"][-1]},
                       "lm_iterations": rep["iters"], "resid_evals_per_call": rep["n_resid_evals"]},
            "local_ba_wall_ms": ms_per_step,
            "residuals_per_s_8d": value, "residuals_per_s_all_evaluations": value_all, "counting": COUNTING,
            "local_ba_cold_call_ms": cold_ms,          # PCIe-inclusive: plan construction + upload + solve + download (never `value`)
            "local_ba_sliding_call_ms": slide_ms,      # the same call with keyframe identities: 19 of the 20 keyframes' planes already on the device
            "local_ba_adapter_call": adapter,          # C++: gather from the object graph + tsba_local_ba + scatter (what optimizer::LocalBundleAdjustment costs its caller)
            "roofline": {"bound": "hbm", "kernel": "k_linearize<FULL> (level 0)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved/HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_us": lin_ms*1e3},
            "roofline_dominant": {"bound": "mfma", "kernel": "reduced-system solve (%d x %d LDL^T in LDS), the largest share of an LM iteration" % (n, n),
                                  "achieved": flops/(sol_ms*1e-3)/1e12, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": flops/(sol_ms*1e-3)/1e12/F64_MFMA_PEAK_TFLOPS, "flops_per_launch": flops, "avg_launch_us": sol_ms*1e3,
                                  "note": "a dependent chain of %d pivots inside one workgroup: latency-bound, the FLOP fraction is not its figure of merit" % n},
        }
        if not args.no_cpu_baseline and world == 1:        # (the CPU baseline is reported at N = 1 only)
            out["cpu_baseline"] = cpu_baseline_local(prob, opt)
        if world == 1 and not args.no_also:                # the other BASELINE configs, outside the timed region of `value`
            try:
                out["also"] = also_lines(gpu, local_rank, torch)
            except Exception as e:                         # noqa: BLE001 -- the headline line must not die with an extra
                out["also"] = {"error": repr(e)}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
