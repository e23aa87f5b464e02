"""GPU diagnostic (not a pytest; needs the -DMID_STAMPS build copied over textslam_amd/libtsba.so, tools/mid_stamps.sh): cycles the blocks of k_mid spend per kind."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from textslam_amd import synth, abi
from textslam_amd.optimizer import Optimizer
opt = Optimizer(0)
P = synth.config_c4(); o = abi.options_local()
opt.upload(P, o)
for k in range(5): rep = opt.solve()
v = (C.c_longlong*64)()
opt.lib.tsba_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
assert opt.lib.tsba_debug_stamps(opt.ctx, v) == 0
v = np.array(list(v), dtype=np.int64)
print("k_linearize (speculative launches), mean cycles per workgroup from the arrival of the LM state, by kind:")
for kind, name in enumerate(("scene pairs (2 per workgroup)", "text group")):
    s_ = v[8*kind: 8*kind + 8]; n = max(int(s_[7]), 1)
    print("  %-30s workgroups %6d   [1] %7.0f   [2] %7.0f   end %7.0f    (scene: [1] = candidates evaluated; text: [1] = operands there, taps requested, [2] = taps evaluated)" % (name, n, s_[1]/n, s_[2]/n, s_[3]/n))
print("k_mid (speculative launches of 5 C4 solves; level mix 2,1,0), mean cycles per block from its start, by kind:")
for kind, name in enumerate(("point blocks", "plane blocks", "pair blocks")):
    s = v[16 + 8*kind: 24 + 8*kind]; n = max(int(s[7]), 1)
    print("  %-13s blocks %6d   state + offsets there %7.0f   records summed and stored %7.0f   end (after the block reductions) %7.0f" % (name, n, s[0]/n, s[1]/n, s[2]/n))
s = v[32:40]; n = max(int(s[7]), 1)
print("  pair blocks in detail: loads there %7.0f   lanes' sums met %7.0f   products formed and stored (host pairs) %7.0f" % (s[3]/n, s[4]/n, s[5]/n))
print("k_schur_t<4> (all launches), mean cycles per workgroup from its start, by kind:")
s = v[56:64]; n = max(int(s[7]), 1)
print("mu / sigma of a text observation (k_pass_begin / k_pass_end), mean cycles per workgroup: operands there %6.0f   corners projected %6.0f   mask cleared %6.0f   quad rasterised %6.0f   histogram %6.0f   end %6.0f   (%d workgroups)" % (s[0]/n, s[1]/n, s[2]/n, s[3]/n, s[4]/n, s[5]/n, n))
print("   inside the quad's rasterisation (thread 0, cycles from its start): boundary lines %6.0f   edge slopes %6.0f   end of the scanline fill %6.0f" % (v[48]/n, v[49]/n, v[50]/n))
for kind, name in enumerate(("diagonal S blocks",)):
    s = v[40 + 8*kind: 48 + 8*kind]; n = max(int(s[7]), 1)
    print("  %-22s workgroups %6d   decision taken %7.0f   slot pairs gathered %7.0f   waves' sums met %7.0f   end %7.0f" % (name, n, s[0]/n, s[1]/n, s[2]/n, s[3]/n))
print("(clock64 at 100 MHz on gfx950? compare with the kernel's 10.4 us)")
