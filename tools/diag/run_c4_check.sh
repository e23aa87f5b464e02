set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_loop.py tests/test_gpu_context_reuse.py -x -q > $OUT/gpu_parity.log 2>&1; echo "pytest rc=$?" >> $OUT/gpu_parity.log
tail -3 $OUT/gpu_parity.log
rm -rf /tmp/prof_c4; rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o c4 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/c4_rocprof.json 2> /tmp/prof_c4.err
python profiles/rocpd_top_kernels.py $(find /tmp/prof_c4 -name "*.db" | head -1) 2>&1 | head -9
python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-260
