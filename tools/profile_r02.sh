#!/bin/bash
# Round-2 profiles on the GPU box: per-kernel statistics (rocprofv3 --kernel-trace --stats) of the three bench workloads, then the PMC
# passes (FETCH_SIZE and WRITE_SIZE in separate runs: they do not fit one pass; SQ counters of the reduced-system solve in a third).
# Output: gpurun_out/r02_*.txt (copied to profiles/ by hand after inspection).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
stats() {   # name, bench args...
  local name=$1; shift
  rm -rf /tmp/prof_$name; rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- python bench.py "$@" --no-cpu-baseline > $OUT/r02_${name}_bench_under_rocprof.json 2> /tmp/prof_$name.err
  python profiles/rocpd_top_kernels.py $(find /tmp/prof_$name -name "*.db" | head -1) > $OUT/r02_${name}_kernel_stats.txt 2>&1
}
pmc() {     # name, counter list, bench args...
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/pmc_$name; rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_$name -o $name -- python bench.py "$@" --no-cpu-baseline > /dev/null 2> /tmp/pmc_$name.err
  python profiles/rocpd_pmc_by_kernel.py $(find /tmp/pmc_$name -name "*.db" | head -1) > $OUT/r02_${name}.txt 2>&1
}
stats c4_local_ba --steps 20 --warmup 3
stats c6_global_ba --workload global_ba --steps 3 --warmup 1
stats orb_batch64 --workload orb --steps 20 --warmup 3
pmc c4_pmc_fetch FETCH_SIZE --steps 3 --warmup 1
pmc c4_pmc_write WRITE_SIZE --steps 3 --warmup 1
pmc c6_pmc_fetch FETCH_SIZE --workload global_ba --steps 1 --warmup 1
pmc c6_pmc_write WRITE_SIZE --workload global_ba --steps 1 --warmup 1
pmc orb_pmc_fetch FETCH_SIZE --workload orb --steps 2 --warmup 1
pmc orb_pmc_write WRITE_SIZE --workload orb --steps 2 --warmup 1
rocprofv3 -L > $OUT/r02_counters_available.txt 2>&1
# SQ counters of the reduced-system solve, in small passes (an unknown counter name only costs its own pass)
pmc c4_pmc_sq_time "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" --steps 3 --warmup 1
pmc c4_pmc_sq_lds "SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" --steps 3 --warmup 1
pmc c4_pmc_sq_mfma "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" --steps 3 --warmup 1
ls -la $OUT/r02_* | head -30
