cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for L in libtsba.so libtsba_u1.so; do
echo "== $L"
rm -rf /tmp/prof_p; TSBA_LIB=textslam_amd/$L rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o p -- python tools/diag/gpu_diag_c6_ab.py > /tmp/p.log 2>&1
grep default /tmp/p.log | head -1
python profiles/rocpd_top_kernels.py $(find /tmp/prof_p -name "*.db" | head -1) 2>&1 | grep "schur"
done 2>&1 | tee gpurun_out/ab2.log
