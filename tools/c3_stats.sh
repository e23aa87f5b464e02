set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
rm -rf /tmp/prof_c3; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o c3 -- python $OLDPWD/tools/diag/gpu_diag_pose.py > /tmp/run_c3.txt 2> /tmp/prof_c3.err )
cat /tmp/run_c3.txt
python profiles/rocpd_top_kernels.py $(find /tmp/prof_c3 -name "*.db" | head -1) 2>&1 | head -12
