#!/bin/bash
# round 6, final tree: the -m gpu suite, smoke, the bench line (with the also block), the ORB profile legs
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/r06_gpu_suite.log; tail -3 gpurun_out/r06_gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a gpurun_out/r06_gpu_suite.log
timeout 900 python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench.err; tail -c 1200 gpurun_out/r06_bench_line.json
bash tools/profile_r06.sh orb > gpurun_out/r06_profile_orb.log 2>&1; cat gpurun_out/r06_orb_one_frame_timeline.txt; head -16 gpurun_out/r06_orb_batch64_kernel_stats.txt
