#!/bin/bash
# SQ counters of the cyclic-reduction level kernel and the interior factorisation (global BA, 5000 keyframes), small passes.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT; : > $OUT/r02_c6_solver_pmc_sq.txt
pmc() {     # counters...
  rm -rf /tmp/pmc_x; rocprofv3 --kernel-trace --pmc $1 -d /tmp/pmc_x -o x -- python bench.py --workload global_ba --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/pmc_x.err
  for k in k_cre_elim k_bandp_factor k_schur_quad; do python profiles/rocpd_pmc_by_kernel.py $(find /tmp/pmc_x -name "*.db" | head -1) $k | tail -n +2 >> $OUT/r02_c6_solver_pmc_sq.txt; done
}
pmc "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"
pmc "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES"
pmc "SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
cat $OUT/r02_c6_solver_pmc_sq.txt
