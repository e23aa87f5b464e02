#!/bin/bash
# Round-3 ORB profiles: per-kernel statistics of the 64-frame batch, the durations of the seven resize launches, FETCH_SIZE / WRITE_SIZE in separate passes.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
name=orb_batch64
rm -rf /tmp/prof_$name; ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- python $OLDPWD/bench.py --workload orb --steps 20 --warmup 3 --no-cpu-baseline --no-also > $OLDPWD/$OUT/r03_${name}_bench_under_rocprof.json 2> /tmp/prof_$name.err )
db=$(find /tmp/prof_$name -name "*.db" | head -1)
python profiles/rocpd_top_kernels.py $db > $OUT/r03_${name}_kernel_stats.txt 2>&1
python profiles/rocpd_kernel_sequence.py $db k_resize 7 >> $OUT/r03_${name}_kernel_stats.txt 2>&1
pmc() { local name=$1 ctr=$2
  rm -rf /tmp/pmc_$name; ( cd /tmp && rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_$name -o $name -- python $OLDPWD/bench.py --workload orb --steps 2 --warmup 1 --no-cpu-baseline --no-also > /dev/null 2> /tmp/pmc_$name.err )
  python profiles/rocpd_pmc_by_kernel.py $(find /tmp/pmc_$name -name "*.db" | head -1) > $OUT/r03_${name}.txt 2>&1; }
pmc orb_pmc_fetch FETCH_SIZE
pmc orb_pmc_write WRITE_SIZE
cat $OUT/r03_${name}_kernel_stats.txt; cat $OUT/r03_orb_pmc_fetch.txt; cat $OUT/r03_orb_pmc_write.txt
