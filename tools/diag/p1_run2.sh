#!/bin/bash
# ORB per-frame chain: tests, A/B chain (0) / few-frames path (1), timeline, then k_octree with 512 / 1024 threads (alternative builds)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_orb.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/p1_tests.txt; cat gpurun_out/p1_tests.txt
( for pyr in 0 1; do ORB_AB_PYR=$pyr ORB_AB_FRAMES=1,2,4,8,12 timeout 300 python tools/diag/gpu_orb_ab.py; done ) 2>&1 | tee gpurun_out/p1_ab.txt
bash tools/diag/timeline_orb_single.sh 2>&1 | tee gpurun_out/p1_timeline.txt
for qt in 512 1024; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DQT=$qt -o /tmp/libtsorb_qt$qt.so textslam_amd/csrc/tsorb.hip 2>/dev/null || { echo "build QT=$qt failed"; continue; }
  TSORB_LIB=/tmp/libtsorb_qt$qt.so ORB_AB_PYR=-1 ORB_AB_FRAMES=1,4,64 timeout 300 python tools/diag/gpu_orb_ab.py
  TSORB_LIB=/tmp/libtsorb_qt$qt.so timeout 300 python -m pytest tests/test_gpu_orb.py -x -q -m gpu 2>&1 | tail -2
done 2>&1 | tee gpurun_out/p1_qt.txt
