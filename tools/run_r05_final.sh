#!/bin/bash
# round 5, final GPU call: the whole -m gpu suite, the round's profiles (kernel statistics + PMC passes), the sanitizer leg that needs a device, the bench line
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 -s 2>&1 | tail -150 > $OUT/r05_final_tests.log
grep -v "^ \|^$" $OUT/r05_final_tests.log | tail -12
timeout 600 python bench.py > $OUT/r05_bench_line.json 2> $OUT/r05_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_bench_line.json"))
print("ms_per_step", d["ms_per_step"], "roofline", d["roofline"]["frac"], "adapter", d.get("local_ba_adapter_call", {}).get("local_ba_adapter_call_ms"))
for k,v in d.get("also",{}).items():
    if isinstance(v,dict): print(k, {a:v[a] for a in v if a.startswith("ms_per") or a in ("cold_call_ms","poll_timeouts","solve_us_per_lm_trial")})
PY
bash tools/profile_r05.sh > $OUT/r05_profile_run.log 2>&1
tail -5 $OUT/r05_profile_run.log
head -12 $OUT/r05_c4_local_ba_kernel_stats.txt
bash tools/sanitize.sh gpuonly > /dev/null 2>&1; cat $OUT/r05_sanitizers_gpu.log | grep SANITIZE
