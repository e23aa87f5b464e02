# A/B over differently built libraries: bash tools/diag/run_ab_libs.sh libtsba.so libtsba_x.so ...   (C6 resident solve, ms)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for L in "$@"; do echo "== $L"; TSBA_LIB=textslam_amd/$L timeout 300 python tools/diag/gpu_diag_c6_ab.py 2>&1 | grep default | head -2; done 2>&1 | tee gpurun_out/ab_libs.log
