cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for P in "$@"; do
rm -rf /tmp/prof_p; rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o p -- python tools/diag/gpu_diag_c6_ab.py band_parts=$P > /tmp/p.log 2>&1
echo "== band_parts=$P"; grep band_parts /tmp/p.log
python profiles/rocpd_top_kernels.py $(find /tmp/prof_p -name "*.db" | head -1) 2>&1 | head -9
done 2>&1 | tee gpurun_out/prof_parts.log
