cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
TSBA_LIB=textslam_amd/libtsba_stamps.so timeout 300 python tools/diag/gpu_diag_cre.py 2>&1 | tee gpurun_out/cre_stamps.log
timeout 300 python tools/diag/gpu_diag_cre.py 2>&1 | tee -a gpurun_out/cre_stamps.log
