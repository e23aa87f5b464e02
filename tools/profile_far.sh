#!/bin/bash
# per-kernel statistics of one gpu_diag_far.py run (rocprofv3 --kernel-trace --stats): tools/profile_far.sh <name> <gpu_diag_far args...>
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
name=$1; shift
rm -rf /tmp/prof_$name; ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- python $OLDPWD/tools/diag/gpu_diag_far.py "$@" > $OLDPWD/$OUT/r04_${name}_run.txt 2> /tmp/prof_$name.err )
python profiles/rocpd_top_kernels.py $(find /tmp/prof_$name -name "*.db" | head -1) > $OUT/r04_${name}_kernel_stats.txt 2>&1
head -30 $OUT/r04_${name}_kernel_stats.txt
