#!/usr/bin/env python3
"""bench.py -- local-BA wall-time and residuals/s on the 20 KF x 5k pts x 100 text-plane window.

One "step" = one complete optimizer::LocalBundleAdjustment (pyramid passes 2,1,0 x <=10 LM iterations, mu/sigma,
outlier passes) on a synthetic window that is already resident in HBM when the timed region starts.
residuals/s = scalar residuals evaluated (once per linearisation and once per LM trial step) / wall time.

N > 1: local BA does not shard ("replicas only") -- every rank runs its own window on its own GPU,
value = sum over ranks (weak scaling), no data-path collective.  `--workload global_ba` runs the sharded global BA.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def cpu_baseline(prob, opt_ref, budget_s=30.0):
    """The CPU restatement of the reference path (oracle, Ceres-style central-difference text Jacobians, 1 thread),
    timed on this host on a bounded sample of the same window."""
    import oracle
    from textslam_amd import abi
    o = abi.TsbaOptions.from_buffer_copy(opt_ref)
    o.text_jacobian = 1                       # NumericDiffCostFunction<CENTRAL>, nume_BAText.h:97-100
    o.n_passes = 1
    o.levels[0] = 2                           # first pass of LocalBundleAdjustment (level 2), optimizer.cc:287
    o.its[0] = opt_ref.its[0]
    o.chi2_mono[0], o.chi2_text[0] = opt_ref.chi2_mono[0], opt_ref.chi2_text[0]
    q = prob.copy()
    t0 = time.perf_counter()
    rep = oracle.solve(q, o)
    dt = time.perf_counter() - t0
    sample = "pass 1 of 3 (pyramid level 2, <=%d LM its) of the same window, numeric-diff text Jacobians" % o.its[0]
    if dt < budget_s / 6:                     # cheap enough: time the whole call instead
        o = abi.TsbaOptions.from_buffer_copy(opt_ref)
        o.text_jacobian = 1
        q = prob.copy()
        t0 = time.perf_counter()
        rep = oracle.solve(q, o)
        dt = time.perf_counter() - t0
        sample = "the full LocalBundleAdjustment call (3 passes) on the same window, numeric-diff text Jacobians"
    return {"value": rep["n_resid_evals"] / dt, "unit": "residuals/s", "cores": 1, "kind": "port",
            "sample": sample, "seconds": dt}


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
        out["roofline"] = {"bound": "hbm", "kernel": "whole ORB pipeline (pyramid, FAST, quadtree, orientation, blur, rBRIEF)", "achieved": algo*world/(dt/args.steps)/1e9,
                           "peak": 8000.0, "unit": "GB/s", "frac": algo/(dt/args.steps)/1e9/8000.0, "traffic": None, "algorithmic_bytes_per_launch": algo}
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="local_ba", choices=["local_ba", "global_ba", "orb"])
    ap.add_argument("--kf", type=int, default=5000, help="global_ba: keyframes (BASELINE configs[4]: 5k KF / 500k observations)")
    ap.add_argument("--pts", type=int, default=70000, help="global_ba: map points (70k points -> ~500k observations)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    import torch
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from textslam_amd import synth, abi
    from textslam_amd.optimizer import Optimizer

    if args.workload == "orb":
        return bench_orb(args, rank, local_rank, world, dist, torch)
    gpu = Optimizer(local_rank)
    if args.workload == "global_ba":
        # one global BA sharded by landmark over the ranks: poses replicated, S and g all-reduced over RCCL
        prob = synth.config_global(n_kf=args.kf, n_pt=args.pts, band=10)      # identical on every rank
        opt = abi.options_global()
        if world > 1:
            idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(gpu.comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, 0)
            gpu.comm_init(bytes(idt.cpu().numpy().tobytes()), rank, world)
    else:
        prob = synth.config_c4(seed=synth.SEED + rank)    # every replica gets its own window
        opt = abi.options_local()
    gpu.upload(prob, opt)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    rep = None
    for _ in range(args.warmup):
        rep = gpu.solve()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rep = gpu.solve()                                  # synchronous: returns after the last kernel
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if args.workload == "global_ba":
            evals_all = float(rep["n_resid_evals"])        # block counts are already global (all-reduced in the library)
        else:
            ev = torch.tensor([float(rep["n_resid_evals"])], dtype=torch.float64, device="cuda")
            dist.all_reduce(ev, op=dist.ReduceOp.SUM)
            evals_all = float(ev.item())
    else:
        evals_all = float(rep["n_resid_evals"])
    ms_per_step = dt / args.steps * 1e3
    value = evals_all * args.steps / dt

    out = None
    if rank == 0 and args.workload == "global_ba":
        out = {"metric": "global_ba_residuals_per_s", "value": value, "unit": "residuals/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic",
               "config": {"workload": "global BA %d KF x %d pts (scene only, 20 LM its, level 0), landmarks sharded over %d GPU(s)"
                                      % (args.kf, args.pts, world), "lm_iterations": rep["iters"], "scene_blocks": rep["n_sblock"],
                          "reduced_system_dim": 6*args.kf}}
        # roofline of the linearisation kernel on this rank's shard (HBM-bound stream of scene blocks, SURVEY 8d: 44 B per block)
        if world == 1:                                     # (with a communicator the pass set-up all-reduces: every rank would have to take part)
            lin_ms, algo_bytes = gpu.time_linearize(0, 50)
            achieved = algo_bytes/(lin_ms*1e-3)/1e9
            out["roofline"] = {"bound": "hbm", "kernel": "k_linearize<FULL> (level 0)", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                               "frac": achieved/8000.0, "traffic": None, "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_us": lin_ms*1e3}
        if not args.no_cpu_baseline and world == 1:
            # the oracle's dense Schur solve is cubic in the keyframes: a bounded instance of the same generator stands in
            import oracle
            small = synth.config_global(n_kf=300, n_pt=30000, band=12)
            t0 = time.perf_counter(); rep_o = oracle.solve(small.copy(), opt); dtc = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": rep_o["n_resid_evals"]/dtc, "unit": "residuals/s", "cores": 1, "kind": "port",
                                   "sample": "the same generator at 300 KF x 30000 pts (the CPU restatement factors the reduced system densely)", "seconds": dtc}
    elif rank == 0:
        # the same window through the one-shot ABI entry point (what the TextSLAM adapter calls per keyframe)
        cold = []
        for _ in range(3):
            g2 = prob.copy(); tc = time.perf_counter(); gpu.LocalBundleAdjustment(g2, options=opt); cold.append((time.perf_counter() - tc)*1e3)
        cold_ms = min(cold)
        gpu.upload(prob, opt)
        # roofline of the linearisation kernel (residual + Jacobian + IRLS weight + J^T J / J^T r sums), level 0
        lin_ms, algo_bytes = gpu.time_linearize(0, 200)
        achieved = algo_bytes / (lin_ms * 1e-3) / 1e9
        # HBM traffic of the same kernel from the PMC passes committed under profiles/ (counters cannot be read from inside
        # the process): 2 x FETCH_SIZE (gfx950 correction of the micro-architecture guide) + WRITE_SIZE, bytes per launch
        traffic, traffic_src = None, None
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_c4_pmc_traffic.json")) as f:
                pm = json.load(f)
            traffic = (2.0 * pm["fetch_size_kb_raw_max"] + pm["write_size_kb_raw_max"]) * 1024.0
            traffic_src = pm["source"]
        except (OSError, KeyError, ValueError):
            pass
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
            "local_ba_cold_call_ms": cold_ms,          # PCIe-inclusive: plan construction + upload + solve + download (never `value`)
            "roofline": {"bound": "hbm", "kernel": "k_linearize<FULL> (level 0)", "achieved": achieved, "peak": 8000.0,
                         "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_us": lin_ms * 1e3},
        }
        if not args.no_cpu_baseline and world == 1:        # (the CPU baseline is reported at N = 1 only)
            out["cpu_baseline"] = cpu_baseline(prob, opt)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
