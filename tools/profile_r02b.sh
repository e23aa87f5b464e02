#!/bin/bash
# Round-2 profiles, second half of the round (after the cyclic-reduction / interior rewrite): per-kernel statistics of the three bench
# workloads and of the sharded global BA at world = 1, 2, 4, 8 through the in-process communicator (one rank on the device at a time).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
stats() {   # name, bench args...
  local name=$1; shift
  rm -rf /tmp/prof_$name; rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- python bench.py "$@" --no-cpu-baseline > $OUT/r02_${name}_bench_under_rocprof.json 2> /tmp/prof_$name.err
  python profiles/rocpd_top_kernels.py $(find /tmp/prof_$name -name "*.db" | head -1) > $OUT/r02_${name}_kernel_stats.txt 2>&1
}
stats c4_local_ba --steps 20 --warmup 3
stats c6_global_ba --workload global_ba --steps 3 --warmup 1
python profiles/rocpd_kernel_by_grid.py $(find /tmp/prof_c6_global_ba -name "*.db" | head -1) k_cre > $OUT/r02_c6_cre_by_level.txt 2>&1
stats orb_batch64 --workload orb --steps 20 --warmup 3
for w in 1 2 4 8; do
  rm -rf /tmp/prof_w$w; rocprofv3 --kernel-trace --stats -d /tmp/prof_w$w -o w$w -- python tools/diag/gpu_multi_rank_profile.py $w > $OUT/r02_multi_rank_w${w}_ranks.jsonl 2> /tmp/prof_w$w.err
  python profiles/rocpd_top_kernels.py $(find /tmp/prof_w$w -name "*.db" | head -1) > $OUT/r02_multi_rank_w${w}_kernel_stats.txt 2>&1
done
python bench.py > $OUT/r02_bench_c4.json 2> /tmp/b1.err
python bench.py --workload global_ba > $OUT/r02_bench_c6.json 2> /tmp/b2.err
ls -la $OUT/r02_* | head -40
