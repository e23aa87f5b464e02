cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python tools/diag/gpu_diag_c6_ab.py "$@" 2>&1 | tee gpurun_out/c6_ab.log
