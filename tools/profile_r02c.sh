#!/bin/bash
# Round-2 profiles, end of the round: kernel statistics of the three bench workloads (+ the ring map), the cyclic-reduction levels by grid,
# and the two PMC passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) of the global-BA line for the linearisation's traffic.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
stats() {   # name, bench args...
  local name=$1; shift
  rm -rf /tmp/prof_$name; rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- python bench.py "$@" --no-cpu-baseline > $OUT/r02_${name}_bench_under_rocprof.json 2> /tmp/prof_$name.err
  python profiles/rocpd_top_kernels.py $(find /tmp/prof_$name -name "*.db" | head -1) > $OUT/r02_${name}_kernel_stats.txt 2>&1
}
pmc() {     # name, counter, bench args...
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/pmc_$name; rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_$name -o $name -- python bench.py "$@" --no-cpu-baseline > /dev/null 2> /tmp/pmc_$name.err
  python profiles/rocpd_pmc_by_kernel.py $(find /tmp/pmc_$name -name "*.db" | head -1) k_linearize > $OUT/r02_${name}.txt 2>&1
}
stats c4_local_ba --steps 20 --warmup 3
stats c6_global_ba --workload global_ba --steps 3 --warmup 1
python profiles/rocpd_kernel_by_grid.py $(find /tmp/prof_c6_global_ba -name "*.db" | head -1) k_cre > $OUT/r02_c6_cre_by_level.txt 2>&1
stats c6_ring_global_ba --workload global_ba --loop --steps 3 --warmup 1
pmc c6_pmc_fetch FETCH_SIZE --workload global_ba --steps 1 --warmup 1
pmc c6_pmc_write WRITE_SIZE --workload global_ba --steps 1 --warmup 1
python bench.py > $OUT/r02_bench_c4.json 2> /tmp/b1.err
python bench.py --workload global_ba > $OUT/r02_bench_c6.json 2> /tmp/b2.err
python bench.py --workload global_ba --loop --no-cpu-baseline > $OUT/r02_bench_c6_ring.json 2> /tmp/b3.err
cat $OUT/r02_c6_pmc_fetch.txt $OUT/r02_c6_pmc_write.txt
head -14 $OUT/r02_c6_global_ba_kernel_stats.txt
