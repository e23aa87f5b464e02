#!/bin/bash
# round 5, GPU call C: the whole -m gpu suite, the window A/B, kernel statistics of the C4 bench run, the default bench line
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 -s 2>&1 | tail -150 > $OUT/r05c_tests.log
grep -v "^ \|^$" $OUT/r05c_tests.log | tail -25
timeout 300 python tools/diag/gpu_ab_window.py > $OUT/r05c_ab.log 2>&1
grep -v "^{" $OUT/r05c_ab.log | tail -14
name=c4_local_ba
rm -rf /tmp/prof_$name; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- python $OLDPWD/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also > $OLDPWD/$OUT/r05c_${name}_bench_under_rocprof.json 2> /tmp/prof_$name.err )
python profiles/rocpd_top_kernels.py $(find /tmp/prof_$name -name "*.db" | head -1) > $OUT/r05c_${name}_kernel_stats.txt 2>&1
cat $OUT/r05c_${name}_kernel_stats.txt | head -24
timeout 600 python bench.py > $OUT/r05c_bench.json 2> $OUT/r05c_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05c_bench.json"))
print("ms_per_step", d["ms_per_step"], "adapter", d.get("local_ba_adapter_call"))
for k,v in d.get("also",{}).items():
    if isinstance(v,dict): print(k, {a:v[a] for a in v if a.startswith("ms_per") or a in ("cold_call_ms","poll_timeouts","solve_us_per_lm_trial")})
PY
