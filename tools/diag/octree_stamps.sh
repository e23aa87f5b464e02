#!/bin/bash
# k_octree by phase (a -DQ_STAMPS build of libtsorb: thread 0's cycles per phase, one (frame, level) problem per row), one resident frame
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DQ_STAMPS ${ORB_DEFS:-} -o /tmp/libtsorb_stamps.so textslam_amd/csrc/tsorb.hip || exit 1
TSORB_LIB=/tmp/libtsorb_stamps.so python - <<'P' 2>&1 | tee gpurun_out/octree_stamps.txt
import ctypes as C, numpy as np
from textslam_amd.orbextractor import ORBextractor, synthetic_frame
ex = ORBextractor(1000, 1.2, 8, 20, 7); ex.upload(synthetic_frame(100)[None])
for _ in range(5): ex.run()
out = np.zeros(32*8, np.int32); ex.lib.tsorb_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int]
ex.lib.tsorb_debug_stamps(ex.ctx, out.ctypes.data_as(C.POINTER(C.c_int32)), out.size)
print("level: gather first-nodes | order counts cut partition lists | arg-max   (cycles of thread 0; passes, candidates, nodes)")
for l in range(8):
    o = out[32*l:32*l+32]; print(l, o[:2], o[2:7], o[7], " passes %d cand %d nodes %d  total %d" % (o[8], o[9], o[10], o[:8].sum() + o[16:24].sum()), " full generations in one step: histogram %d, counts %d, scan %d, nodes %d, scatter %d" % tuple(o[16:21]))
P
