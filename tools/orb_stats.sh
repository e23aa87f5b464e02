#!/bin/bash
# per-kernel statistics of the ORB batch of 64 (rocprofv3 --kernel-trace --stats): quick look after a kernel change
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
rm -rf /tmp/prof_orb; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_orb -o orb -- python $OLDPWD/bench.py --workload orb --steps 20 --warmup 3 --no-cpu-baseline > /tmp/run_orb.json 2> /tmp/prof_orb.err )
python -c "import json; d=json.load(open('/tmp/run_orb.json')); print('ms_per_step', d['ms_per_step'])"
python profiles/rocpd_top_kernels.py $(find /tmp/prof_orb -name "*.db" | head -1) 2>&1 | head -${1:-14}
