#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_orb.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/p1_tests.txt; cat gpurun_out/p1_tests.txt
( for pyr in 0 3 0 3; do ORB_AB_PYR=$pyr ORB_AB_FRAMES=16,32,64 timeout 300 python tools/diag/gpu_orb_ab.py; done ) 2>&1 | tee gpurun_out/p1_pairs_ab.txt
