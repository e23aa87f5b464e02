cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
TSBA_LIB=textslam_amd/libtsba_stamps.so timeout 300 python tools/diag/gpu_diag_solve.py 2>&1 | tee gpurun_out/solve_stamps.log
