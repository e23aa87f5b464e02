set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_global.py -x -q > $OUT/gpu_global.log 2>&1; echo "pytest rc=$?" >> $OUT/gpu_global.log
tail -4 $OUT/gpu_global.log
timeout 600 python tools/diag/gpu_diag_c6_ab.py "$@" 2>&1 | tee $OUT/c6_ab.log
timeout 300 python tools/diag/gpu_diag_cre.py 2>&1 | grep bandp | tee $OUT/cre_stamps.log
rm -rf /tmp/prof_c6; rocprofv3 --kernel-trace --stats -d /tmp/prof_c6 -o c6 -- python bench.py --workload global_ba --steps 3 --warmup 1 --no-cpu-baseline > $OUT/c6_rocprof.json 2> /tmp/prof_c6.err
DB=$(find /tmp/prof_c6 -name "*.db" | head -1)
python profiles/rocpd_top_kernels.py $DB > $OUT/c6_kernel_stats_new.txt 2>&1
head -12 $OUT/c6_kernel_stats_new.txt
