#!/bin/bash
# The sharded global BA at world = 1, 2, 4, 8 through the in-process communicator (one rank on the device at a time): kernel statistics per world size.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
for w in 1 2 4 8; do
  rm -rf /tmp/prof_w$w; rocprofv3 --kernel-trace --stats -d /tmp/prof_w$w -o w$w -- python tools/diag/gpu_multi_rank_profile.py $w > $OUT/r02_multi_rank_w${w}_ranks.jsonl 2> /tmp/prof_w$w.err
  python profiles/rocpd_top_kernels.py $(find /tmp/prof_w$w -name "*.db" | head -1) > $OUT/r02_multi_rank_w${w}_kernel_stats.txt 2>&1
done
ls $OUT/r02_multi_rank_*
