"""A/B timing of the window solve (C4 by default) under tsba_debug_options variants, alternating in one process.
    python tools/diag/gpu_ab_window.py [--n 40] name=key:val,key:val ...      (default set: production vs the round-4 paths)"""
import argparse
import json
import os
import sys
import time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer

DEFAULT = ["production=", "ldl_scratch=solve_variant:5", "pass_launches=pass_launches:1", "lin_mid_fused=trial_launches:2", "round4=solve_variant:5,pass_launches:1",
           "separate_launches=solve_variant:3"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=40)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("variants", nargs="*", default=DEFAULT)
    a = ap.parse_args()
    gpu = Optimizer(0)
    P, o = synth.config_c4(), abi.options_local()
    res = {}
    for rnd in range(a.rounds):
        for v in a.variants:
            name, _, kv = v.partition("=")
            kw = {k: int(x) for k, x in (p.split(":") for p in kv.split(",") if p)}
            gpu.debug_set(**kw)
            gpu.upload(P, o)
            for _ in range(3):
                rep = gpu.solve()
            ts = []
            for _ in range(a.n):
                t0 = time.perf_counter(); rep = gpu.solve(); ts.append((time.perf_counter() - t0)*1e3)
            res.setdefault(name, []).append((float(np.median(ts)), float(min(ts))))
            res.setdefault(name + "_rep", (rep["iters"], rep["accepted"], rep["cost1"][-1], rep["poll_timeouts"]))
    gpu.debug_set()
    for k, v in res.items():
        print(k, v)
    print(json.dumps({k: v for k, v in res.items()}))


if __name__ == "__main__":
    main()
