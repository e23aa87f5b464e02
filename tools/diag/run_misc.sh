cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/gpu_tests.log
tail -3 gpurun_out/gpu_tests.log
timeout 600 python tools/diag/gpu_diag_global_text.py 2>&1 | tail -6 | tee gpurun_out/c5.log
timeout 600 python bench.py --workload global_ba --loop --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400 | tee gpurun_out/ring.log
timeout 600 python bench.py --workload global_ba --kf 1000 --pts 100000 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300 | tee gpurun_out/kf1000.log
