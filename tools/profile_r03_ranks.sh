#!/bin/bash
# The sharded global BA on the map with 1 % long-range points at world = 1, 2, 4, 8 through the in-process communicator (one rank on the device at a
# time): kernel statistics per world size -- which kernels shrink with the rank count (linearisation, Schur assembly) and which are replicated (the
# conjugate gradients and their preconditioner).  The collectives go through host memory here: their time is not RCCL's.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
for w in 1 2 4 8; do
  rm -rf /tmp/prof_w$w; ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_w$w -o w$w -- python $OLDPWD/tools/diag/gpu_multi_rank_profile.py $w 5000 70000 0.01 > $OLDPWD/$OUT/r03_long_range_ranks_w${w}.jsonl 2> /tmp/prof_w$w.err )
  python profiles/rocpd_top_kernels.py $(find /tmp/prof_w$w -name "*.db" | head -1) > $OUT/r03_long_range_ranks_w${w}_kernel_stats.txt 2>&1
  head -12 $OUT/r03_long_range_ranks_w${w}_kernel_stats.txt | cut -c1-110
done
