#!/bin/bash
# ORB per-frame chain: tests, A/B chain (0) / few-frames path (default), timeline
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_orb.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/p1_tests.txt; cat gpurun_out/p1_tests.txt
( for pyr in 0 -1; do ORB_AB_PYR=$pyr ORB_AB_FRAMES=1,2,4,8,12,64 timeout 300 python tools/diag/gpu_orb_ab.py; done ) 2>&1 | tee gpurun_out/p1_ab.txt
bash tools/diag/timeline_orb_single.sh 2>&1 | tee gpurun_out/p1_timeline.txt
