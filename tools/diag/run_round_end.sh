# last GPU call of a round: the whole GPU suite on the final code, the host-side plan timings on the GPU box's CPU, the cold global BA call
# (open chain and ring) with the upload's lap times, one local-BA bench line
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 400 python -m pytest tests -m gpu -x -q > $OUT/r02_final_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $OUT/r02_final_gpu_tests.log
tail -3 $OUT/r02_final_gpu_tests.log
timeout 90 python tools/diag/host_plan_time.py > $OUT/r02_host_plan_time.log 2>&1
grep -v "^\[build" $OUT/r02_host_plan_time.log
timeout 90 python tools/diag/gpu_diag_cold_global.py > $OUT/r02_cold_global.log 2>&1
timeout 90 python tools/diag/gpu_diag_cold_global.py loop > $OUT/r02_cold_global_loop.log 2>&1
grep -h "^call\|tsba_upload" $OUT/r02_cold_global.log $OUT/r02_cold_global_loop.log
timeout 120 python bench.py --no-cpu-baseline > $OUT/r02_bench_c4_final.json 2> /tmp/b1.err
python -c "
import json
d=json.loads(open('$OUT/r02_bench_c4_final.json').read().strip().splitlines()[-1]); print('c4', round(d['ms_per_step'],3), round(d['local_ba_cold_call_ms'],3), round(d['roofline']['frac'],4))"
