# last seconds of the round's GPU budget: the cold global BA call with the pinned plan threads, and the tests that go through the large plans
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 40 python tools/diag/gpu_diag_cold_global.py > $OUT/r02_cold_global.log 2>&1
grep -h "^call\|tsba_upload" $OUT/r02_cold_global.log
timeout 70 python -m pytest tests/test_gpu_context_reuse.py tests/test_gpu_parity.py tests/test_cxx_adapter.py tests/test_gpu_global.py -m gpu -x -q -k "not test_gpu_global or c6_full_size or c5_full_size or ring_partition or multi_rank" > $OUT/r02_last_tests.log 2>&1
grep -E "passed|failed" $OUT/r02_last_tests.log | tail -1
