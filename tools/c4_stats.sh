#!/bin/bash
# per-kernel statistics of resident C4 solves (rocprofv3 --kernel-trace --stats): quick look after a kernel change
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
rm -rf /tmp/prof_c4; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o c4 -- python $OLDPWD/tools/diag/gpu_c4_loop.py 20 > /tmp/run_c4.txt 2> /tmp/prof_c4.err )
cat /tmp/run_c4.txt
python profiles/rocpd_top_kernels.py $(find /tmp/prof_c4 -name "*.db" | head -1) 2>&1 | head -${1:-12}
