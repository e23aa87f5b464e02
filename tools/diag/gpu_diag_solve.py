"""GPU diagnostic (not a pytest): k_solve cycle stamps on the C4 window. Run with TSBA_LIB=textslam_amd/libtsba_stamps.so."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer

opt = Optimizer(0)
opt.debug_set(solve_variant=int(os.environ.get("SOLVE_VARIANT", "0")))
P = synth.config_c4(); o = abi.options_local()
opt.upload(P, o)
for _ in range(3):
    rep = opt.solve()
print("t_solve_ms", rep['t_solve_ms'])
st = (ctypes.c_longlong * 64)()
opt.lib.tsba_debug_stamps(opt.ctx, st)
print("k_solve stamps (cycles): load %d factor %d backsub %d nfree %d" % (st[0], st[1], st[2], st[6]))
for i, nm in enumerate(("P wave0 (look-ahead | scratch+ldl | solve | barrier)", "T wave2 (- | - | trailing | barrier)", "T last")):
    print("   %s: lookahead+scratch %d  ldl %d  solve/trailing %d  barrier-wait %d" % ((nm,) + tuple(st[8+4*i:12+4*i])))
print("k_decide stamps (cycles): sums_local %d  pose_scale %d  partials %d  decision %d" % tuple(st[32:36]))
print("   fused: partials %d  offsets+range sums %d  exchange+finalize %d  reduction %d" % tuple(st[40:44]))
print("k_linearize text group 0 (cycles): st+record %d  level-2 loads %d  feature+tap fetch %d  tap math %d  reduce %d" % tuple(st[48:53]))
ms = ctypes.c_double()
opt.lib.tsba_debug_time_solve.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
rc = opt.lib.tsba_debug_time_solve(opt.ctx, 200, ctypes.byref(ms))
print("k_solve back-to-back: rc %d, %.2f us per launch" % (rc, ms.value*1e3))

if hasattr(opt.lib, "tsba_debug_step_stamps"):
    ss = (ctypes.c_longlong * 128)()
    opt.lib.tsba_debug_step_stamps(opt.ctx, ss)
    print("per step (cycles): jb | P0 load+apply+scratch (D: apply+solve) | P0 scratch-read+ldl (D: diag update) | P0 solve+store (D: ldl) | T wave2 tiles; P wave (look-ahead schedule) steps 0-3:", list(st[44:48]))
    for jb in range(int(st[6]) or 17):
        print("   %2d | %5d | %5d | %5d | %5d" % (jb, ss[jb], ss[32 + jb], ss[64 + jb], ss[96 + jb]))
