"""Host-only diagnostic (no GPU work): lap times of the plan builder on the 5000-keyframe map, open chain and ring, with the S-block keys
marked by the calling thread (mark_mt 0) or by the plan threads (1), and for 1 .. 16 plan threads.
usage: python tools/diag/host_plan_time.py"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import load_library
L = load_library()
L.tsba_debug_plan_time.argtypes = [C.POINTER(abi.TsbaProblem), C.POINTER(abi.TsbaOptions), C.c_int, C.c_int, C.POINTER(C.c_double)]
L.tsba_debug_plan_knob.argtypes = [C.c_int, C.c_int]; L.tsba_debug_plan_knob.restype = None
o = abi.options_global()
def topo():
    try:
        cpu = os.sched_getaffinity(0); import glob
        me = open("/proc/self/stat").read().split()[38]
        l3 = open("/sys/devices/system/cpu/cpu%s/cache/index3/shared_cpu_list" % me).read().strip()
        quota = open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?"
        print("host: %d CPUs online, %d in the affinity mask, this thread on cpu %s, its L3 domain: %s, cgroup cpu.max: %s" % (os.cpu_count(), len(cpu), me, l3, quota), flush=True)
    except Exception as e:
        print("host topology: n/a (%s)" % e, flush=True)
topo()
for loop in (False, True):
    P = synth.config_global(n_kf=5000, n_pt=70000, band=10, loop=loop); s = P.struct(); ms = C.c_double(0)
    for mt in (0, 1, 0, 1):
        L.tsba_debug_plan_knob(1, mt)
        L.tsba_debug_plan_time(C.byref(s), C.byref(o), 0, -8, C.byref(ms))
        print("loop %d mark_mt %d: plan %.2f ms (recycled plan object, mean of 8)" % (loop, mt, ms.value), flush=True)
    for pin in (1, 0, 1, 0):
        L.tsba_debug_plan_knob(3, pin)
        L.tsba_debug_plan_time(C.byref(s), C.byref(o), 0, -8, C.byref(ms))
        print("loop %d pinned to the caller's L3 domain %d: plan %.2f ms (recycled plan object, mean of 8)" % (loop, pin, ms.value), flush=True)
    L.tsba_debug_plan_knob(3, 0)
    if not loop:
        for t in (1, 2, 4, 8, 16, 32, 0):
            L.tsba_debug_plan_knob(0, t)
            L.tsba_debug_plan_time(C.byref(s), C.byref(o), 0, 8, C.byref(ms))
            print("threads %d: plan %.2f ms (fresh plan object)" % (t, ms.value), flush=True)
        L.tsba_debug_plan_knob(0, 0)
