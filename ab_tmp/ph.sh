#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for v in full ph1 ph2 ph3 ph4; do
  if [ $v = full ]; then unset TSORB_LIB; else export TSORB_LIB=$PWD/ab_tmp/libtsorb_$v.so; fi
  rm -rf /tmp/p_$v; ( cd /tmp && ORB_AB_FRAMES=64 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_$v -o $v -- python $OLDPWD/tools/diag/gpu_orb_ab.py > /dev/null 2>/tmp/p_$v.err )
  echo "== $v"; tail -5 /tmp/p_$v.err; python profiles/rocpd_top_kernels.py $(find /tmp/p_$v -name "*.db" | head -1) 2>&1 | head -12
done
