#!/bin/bash
# per-kernel statistics of the C4 bench line alone (rocprofv3 --kernel-trace --stats): gpurun_out/r04_c4_local_ba_kernel_stats.txt
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
name=c4_local_ba
rm -rf /tmp/prof_$name; ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- python $OLDPWD/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also > $OLDPWD/$OUT/r04_${name}_bench_under_rocprof.json 2> /tmp/prof_$name.err )
python profiles/rocpd_top_kernels.py $(find /tmp/prof_$name -name "*.db" | head -1) > $OUT/r04_${name}_kernel_stats.txt 2>&1
head -12 $OUT/r04_${name}_kernel_stats.txt
