cd "${GRAFT_REPO_ROOT:-.}"
timeout 300 python tools/diag/gpu_diag_c6_ab.py band_parts=127 band_parts=128 2>&1 | tail -4
