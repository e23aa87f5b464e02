"""Run-to-run reproducibility of the 5000-keyframe solves alone and beside a second context that keeps the device busy (an ORB extractor looping on
another host thread), under solver-path switches: which launch, if any, makes the LM run depend on timing.
    python tools/diag/gpu_nondet.py [open_chain|long_range] [runs]"""
import sys
import os
import threading
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
from textslam_amd.orbextractor import ORBextractor, synthetic_frame

which = sys.argv[1] if len(sys.argv) > 1 else "open_chain"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 20
P = synth.config_global(n_kf=5000, n_pt=70000, band=10, far_frac=0.01 if which == "long_range" else 0.0)
o = abi.options_global(); o.its[0] = 6
gpu = Optimizer(0)


def solve():
    gpu.upload(P, o); rep = gpu.solve()
    return rep, gpu.lm_trace(0)


def first_diff(ta, tb):
    for k, (a, b) in enumerate(zip(ta, tb)):
        if not np.array_equal(a, b, equal_nan=True):
            return k, a, b
    return None


def batch(tag, ref=None):
    rep0, tr0 = solve() if ref is None else ref
    bad = 0; seen = []
    for _ in range(runs):
        rep, tr = solve()
        d = first_diff(tr, tr0)
        if d is not None:
            bad += 1
            if len(seen) < 3:
                k, a, b = d
                seen.append(f"trial {k}: cost {a[0]:.17g} vs {b[0]:.17g} (rel {abs(a[0] - b[0])/abs(b[0]):.1e}), verdict {a[3]} vs {b[3]}, time-outs {rep['poll_timeouts']}")
    print(f"{tag}: {bad} of {runs} runs differ from the first" + ("".join("\n      " + s for s in seen)), flush=True)
    return rep0, tr0


class Busy:
    def __init__(self):
        self.ex = ORBextractor(device=0); self.ex.upload(np.stack([synthetic_frame(s) for s in range(16)]))
        self.stop = threading.Event(); self.n = 0
        self.th = threading.Thread(target=self.loop, daemon=True)
    def loop(self):
        while not self.stop.is_set():
            self.ex.run(); self.n += 1
    def __enter__(self): self.th.start(); return self
    def __exit__(self, *a): self.stop.set(); self.th.join(timeout=30)


VARIANTS = [("production", {}), ("no_schur_quad", dict(no_schur_quad=1)), ("no_small_pairs", dict(no_small_pairs=1)), ("band_parts=1 (one workgroup streams the band)", dict(band_parts=1)),
            ("sep_solver=3 (pivot / update / back kernels)", dict(sep_solver=3)), ("sep_solver=1 (sequential separator solve)", dict(sep_solver=1)),
            ("no_band_stream (wide-band Cholesky)", dict(no_band_stream=1)), ("assume_cus=4", dict(assume_cus=4))]
if len(sys.argv) > 3:
    VARIANTS = [v for v in VARIANTS if v[0].split()[0].split("=")[0] in sys.argv[3:]]
refs = {}
for name, kw in VARIANTS:
    gpu.debug_set(**kw)
    refs[name] = batch("alone, " + name)
with Busy() as b:
    while b.n < 2:
        pass
    for name, kw in VARIANTS:
        gpu.debug_set(**kw)
        batch("beside a busy context, " + name, refs[name])
    print("ORB batches meanwhile:", b.n)
gpu.debug_set()
