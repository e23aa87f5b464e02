# after a change of the host-side plan builder: its lap times on the GPU box's CPU, the cold global BA call, a GPU subset through the new plans
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
if [ "${1:-}" != "short" ]; then
timeout 90 python tools/diag/host_plan_time.py > $OUT/r02_host_plan_time.log 2>&1
grep -v "^\[build" $OUT/r02_host_plan_time.log
grep "^\[build" $OUT/r02_host_plan_time.log | tail -11
fi
timeout 90 python tools/diag/gpu_diag_cold_global.py > $OUT/r02_cold_global.log 2>&1
timeout 90 python tools/diag/gpu_diag_cold_global.py loop > $OUT/r02_cold_global_loop.log 2>&1
grep -h "^call\|tsba_upload" $OUT/r02_cold_global.log $OUT/r02_cold_global_loop.log
timeout 200 python -m pytest tests/test_gpu_context_reuse.py tests/test_gpu_parity.py tests/test_cxx_adapter.py tests/test_gpu_global.py -m gpu -x -q -k 'not full_size and not c6 and not c5' > $OUT/r02_plan_check_tests.log 2>&1
grep -E "passed|failed" $OUT/r02_plan_check_tests.log | tail -1
