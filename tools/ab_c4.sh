#!/bin/bash
# A/B of libtsba.so builds on the resident C4 window: tools/bin/libtsba_<v>.so put in place of the product library for one run each,
# kernel statistics (rocprofv3 --kernel-trace --stats) of 20 solves.  usage: tools/ab_c4.sh v1 v2 ...   (the product build is always run first)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
cp textslam_amd/libtsba.so /tmp/libtsba_prod.so
for v in prod "$@"; do
  if [ $v = prod ]; then cp /tmp/libtsba_prod.so textslam_amd/libtsba.so; else cp tools/bin/libtsba_$v.so textslam_amd/libtsba.so; fi
  rm -rf /tmp/prof_$v; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o $v -- python $OLDPWD/tools/diag/gpu_c4_loop.py 20 > /tmp/run_$v.txt 2> /tmp/prof_$v.err )
  echo "== $v: $(cat /tmp/run_$v.txt)"
  python profiles/rocpd_top_kernels.py $(find /tmp/prof_$v -name "*.db" | head -1) 2>&1 | head -${TOPN:-9}
  python tools/diag/gpu_c4_loop.py 40
done
cp /tmp/libtsba_prod.so textslam_amd/libtsba.so
