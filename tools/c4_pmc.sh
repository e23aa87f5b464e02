#!/bin/bash
# PMC counters by kernel for resident C4 solves (separate pass per counter group; --pmc only with --kernel-trace)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for grp in "$@"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-60)
  rm -rf /tmp/pmc_$name; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_$name -o p -- python $OLDPWD/tools/diag/gpu_c4_loop.py 3 > /tmp/run_pmc.txt 2> /tmp/pmc_$name.err )
  echo "== $grp"
  python profiles/rocpd_pmc_by_kernel.py $(find /tmp/pmc_$name -name "*.db" | head -1) "" 2>&1 | grep -E "kernel|k_mid|k_linearize|k_schur_t|k_solve_back|k_decide"
done
