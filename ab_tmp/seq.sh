#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf /tmp/p_full; ( cd /tmp && ORB_AB_FRAMES=64 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_full -o full -- python $OLDPWD/tools/diag/gpu_orb_ab.py > /dev/null 2>/tmp/p_full.err )
DB=$(find /tmp/p_full -name "*.db" | head -1)
python ab_tmp/seq.py $DB
cp $DB gpurun_out/orb_full.db
