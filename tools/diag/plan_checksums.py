"""Host-only: checksums over EVERY list of the plans of a set of problems (windows, maps, loop closures, text planes, shards, ring / reordering
plans) for several plan-thread counts -- run it with two builds of the library (TSBA_LIB=...) and diff the output: a change of the plan builder
that is meant to keep the plans must keep every line.
usage: python tools/diag/plan_checksums.py > sums.txt"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import load_library
L = load_library()
L.tsba_debug_plan_checksum.argtypes = [C.POINTER(abi.TsbaProblem), C.POINTER(abi.TsbaOptions), C.c_int, C.c_int]; L.tsba_debug_plan_checksum.restype = C.c_ulonglong
L.tsba_debug_plan_knob.argtypes = [C.c_int, C.c_int]; L.tsba_debug_plan_knob.restype = None
og, ol, op = abi.options_global(), abi.options_local(), abi.options_pose()
def gtext():
    o = abi.options_global(); o.use_text = 1; return o
cases = [
    ("tiny", synth.tiny(seed=3, n_kf=8, n_pt=300, n_text=6), ol, (0, 1, 2)),
    ("tiny5", synth.tiny(), ol, (0, 1, 2)),
    ("c1", synth.config_c1(), ol, (0, 1, 2)),
    ("c3", synth.config_c3(), op, (0, 1, 2)),
    ("c4", synth.config_c4(), ol, (0, 1, 2)),
    ("init", synth.init_pair(), abi.options_init(), (0,)),
    ("landmarker", synth.landmark_refine(), abi.options_landmarker(), (0,)),
    ("g700", synth.config_global(n_kf=700, n_pt=20000, band=9), og, (0,)),
    ("g60", synth.config_global(n_kf=60, n_pt=3000, band=6), og, (0,)),
    ("g40", synth.config_global(n_kf=40, n_pt=2000, band=6), og, (0,)),
    ("gtext", synth.make_problem(n_kf=60, n_pt=3000, n_text=40, seed=5), gtext(), (0,)),
    ("gfar", synth.config_global(n_kf=600, n_pt=12000, band=8, far_frac=0.01), og, (0,)),
    ("gloop", synth.config_global(n_kf=600, n_pt=12000, band=8, loop=True), og, (0,)),
    ("gtail", synth.config_global(n_kf=600, n_pt=12000, band=8, loop=True, loop_at=200), og, (0,)),
]
if len(sys.argv) > 1 and sys.argv[1] == "big":
    cases += [("g5000", synth.config_global(n_kf=5000, n_pt=70000, band=10), og, (0,)),
              ("g5000loop", synth.config_global(n_kf=5000, n_pt=70000, band=10, loop=True), og, (0,)),
              ("g5000tail", synth.config_global(n_kf=5000, n_pt=70000, band=10, loop=True, loop_at=1500), og, (0,))]
for name, P, o, levels in cases:
    s = P.struct()
    for ring in (0, 13):
        L.tsba_debug_plan_knob(2, ring)
        for nshard in (1, 3):
            for shard in range(nshard):
                o.lm_shard, o.lm_nshard = shard, nshard
                for lev in levels:
                    if lev >= P.n_levels: continue
                    for t in (1, 3, 16):
                        print(name, "ring", ring, "shard %d/%d" % (shard, nshard), "level", lev, "threads", t, hex(L.tsba_debug_plan_checksum(C.byref(s), C.byref(o), lev, t)), flush=True)
    o.lm_shard, o.lm_nshard = 0, 1
