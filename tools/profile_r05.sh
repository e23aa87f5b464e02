#!/bin/bash
# Round-5 profiles on the GPU box (one call): per-kernel statistics (rocprofv3 --kernel-trace --stats) of the bench workloads and of the maps with long-range
# coupling / two closures, then FETCH_SIZE / WRITE_SIZE in separate --pmc passes (never combined with other trace domains) for the linearisation kernel of
# C4 and C6 and for the whole ORB pipeline (its last PMC pass was round 3).  Output: gpurun_out/r05_*.txt (the cited ones are copied to profiles/).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
stats() {   # name, bench args...
  local name=$1; shift
  rm -rf /tmp/prof_$name; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- python $OLDPWD/bench.py "$@" --no-cpu-baseline --no-also > $OLDPWD/$OUT/r05_${name}_bench_under_rocprof.json 2> /tmp/prof_$name.err )
  python profiles/rocpd_top_kernels.py $(find /tmp/prof_$name -name "*.db" | head -1) > $OUT/r05_${name}_kernel_stats.txt 2>&1
}
diag() {    # name, gpu_diag_far args...
  local name=$1; shift
  rm -rf /tmp/prof_$name; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- python $OLDPWD/tools/diag/gpu_diag_far.py "$@" > $OLDPWD/$OUT/r05_${name}_run.txt 2> /tmp/prof_$name.err )
  python profiles/rocpd_top_kernels.py $(find /tmp/prof_$name -name "*.db" | head -1) > $OUT/r05_${name}_kernel_stats.txt 2>&1
}
pmc() {     # name, counter, kernel filter, bench args...
  local name=$1 ctr=$2 filt=$3; shift 3
  rm -rf /tmp/pmc_$name; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_$name -o $name -- python $OLDPWD/bench.py "$@" --no-cpu-baseline --no-also > /dev/null 2> /tmp/pmc_$name.err )
  python profiles/rocpd_pmc_by_kernel.py $(find /tmp/pmc_$name -name "*.db" | head -1) $filt > $OUT/r05_${name}.txt 2>&1
}
ONLY=${1:-all}      # "orb": just the ORB legs (after a change of that pipeline)
if [ "$ONLY" = all ]; then
stats c4_local_ba --steps 20 --warmup 3
stats c6_global_ba --workload global_ba --steps 3 --warmup 1
fi
stats orb_batch64 --workload orb --steps 20 --warmup 3
if [ "$ONLY" = all ]; then
diag c6_long_range 5000 70000 0.01 2
diag c6_two_closures 5000 70000 0.0 2 2
pmc c4_pmc_fetch FETCH_SIZE k_linearize --steps 3 --warmup 1
pmc c4_pmc_write WRITE_SIZE k_linearize --steps 3 --warmup 1
pmc c6_pmc_fetch FETCH_SIZE k_linearize --workload global_ba --steps 2 --warmup 1
pmc c6_pmc_write WRITE_SIZE k_linearize --workload global_ba --steps 2 --warmup 1
fi
pmc orb_pmc_fetch FETCH_SIZE "" --workload orb --steps 2 --warmup 1
pmc orb_pmc_write WRITE_SIZE "" --workload orb --steps 2 --warmup 1
ls -la $OUT/r05_* | head -60
