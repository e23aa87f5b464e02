#!/bin/bash
# round 5, GPU call B: the window-path tests after the k_lin_mid change, the window A/B, per-kernel statistics of the C4 bench run (launch count per solve)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_polling.py tests/test_gpu_context_reuse.py tests/test_cxx_adapter.py "tests/test_gpu_fullsize_oracle.py::test_converged_answers_against_the_oracle" -m gpu -q -p no:cacheprovider --timeout 600 2>&1 | tail -150 > $OUT/r05b_tests.log
tail -8 $OUT/r05b_tests.log
timeout 300 python tools/diag/gpu_ab_window.py > $OUT/r05b_ab.log 2>&1
grep -v "^{" $OUT/r05b_ab.log | tail -14
name=c4_local_ba
rm -rf /tmp/prof_$name; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- python $OLDPWD/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also > $OLDPWD/$OUT/r05b_${name}_bench_under_rocprof.json 2> /tmp/prof_$name.err )
python profiles/rocpd_top_kernels.py $(find /tmp/prof_$name -name "*.db" | head -1) > $OUT/r05b_${name}_kernel_stats.txt 2>&1
cat $OUT/r05b_${name}_kernel_stats.txt | head -30
