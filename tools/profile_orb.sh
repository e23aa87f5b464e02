#!/bin/bash
# per-kernel statistics of the ORB batch (rocprofv3 --kernel-trace --stats of bench.py --workload orb): tools/profile_orb.sh <name>
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
name=${1:-orb_batch64}
rm -rf /tmp/prof_$name; ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- python $OLDPWD/bench.py --workload orb --steps 20 --warmup 3 --no-cpu-baseline --no-also > $OLDPWD/$OUT/r03_${name}_bench_under_rocprof.json 2> /tmp/prof_$name.err )
python profiles/rocpd_top_kernels.py $(find /tmp/prof_$name -name "*.db" | head -1) > $OUT/r03_${name}_kernel_stats.txt 2>&1
head -14 $OUT/r03_${name}_kernel_stats.txt; python profiles/rocpd_kernel_sequence.py $(find /tmp/prof_$name -name "*.db" | head -1) k_resize 14
