set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
rm -rf /tmp/prof_c6; rocprofv3 --kernel-trace --stats -d /tmp/prof_c6 -o c6 -- python bench.py --workload global_ba --steps 3 --warmup 1 --no-cpu-baseline > $OUT/r02_c6_global_ba_bench_under_rocprof.json 2> /tmp/prof_c6.err
python profiles/rocpd_top_kernels.py $(find /tmp/prof_c6 -name "*.db" | head -1) > $OUT/r02_c6_global_ba_kernel_stats.txt 2>&1
python bench.py --workload global_ba --no-cpu-baseline > $OUT/c6_plain.json 2>&1
python bench.py --no-cpu-baseline > $OUT/c4_plain.json 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $OUT/gpu_tests.log
tail -3 $OUT/gpu_tests.log; head -20 $OUT/r02_c6_global_ba_kernel_stats.txt; cat $OUT/c6_plain.json | tail -1 | cut -c1-400
