#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf /tmp/p_pose; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_pose -o pose -- python $OLDPWD/ab_tmp/pose_dbg.py > /dev/null 2>/tmp/p_pose.err )
DB=$(find /tmp/p_pose -name "*.db" | head -1); cp $DB gpurun_out/pose.db; python profiles/rocpd_top_kernels.py $DB | head -14
