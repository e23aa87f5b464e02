#!/bin/bash
# k_bandp_factor variants (round 6 experiment: the kernel spills 248 bytes per lane at 768 threads / 168 registers): C6 open chain, kernel statistics per variant
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
cp textslam_amd/libtsba.so /tmp/libtsba_prod.so
for v in prod "$@"; do
  if [ $v = prod ]; then cp /tmp/libtsba_prod.so textslam_amd/libtsba.so; else cp tools/bin/libtsba_bandp_$v.so textslam_amd/libtsba.so; fi
  rm -rf /tmp/prof_$v; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o $v -- python $OLDPWD/bench.py --workload global_ba --steps 3 --warmup 1 --no-cpu-baseline > /tmp/run_$v.json 2> /tmp/prof_$v.err )
  echo "== $v: $(python -c "import json; d=json.load(open('/tmp/run_$v.json')); print(d['ms_per_step'], d['config']['lm_iterations'])")"
  python profiles/rocpd_top_kernels.py $(find /tmp/prof_$v -name "*.db" | head -1) 2>&1 | grep -E "k_bandp_factor|k_cre_elim|TOTAL"
done
cp /tmp/libtsba_prod.so textslam_amd/libtsba.so
