set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $OUT/gpu_tests.log
tail -4 $OUT/gpu_tests.log
python bench.py --workload global_ba --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
