#!/bin/bash
# The leg of tools/sanitize.sh that needs the GPU box: the HOST side of libtsba.so under clang UBSan + bounds, driven by the GPU parity tests (real uploads / solves through the
# instrumented host code; the device code is not instrumented).  UBSan only: the memory-error sanitizer of the device toolchain needs xnack, which this pool does not enable, and its
# runtime aborts the first device allocation on a node without it -- tools/sanitize.sh (the CPU legs, with that sanitizer) is listed in .gpurunignore and never travels to the GPU box.
# Usage (on the GPU box): bash tools/sanitize_gpu.sh  -> gpurun_out/r06_sanitizers_gpu.log
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/r06_sanitizers_gpu.log; : > $LOG
say() { echo "SANITIZE $*" | tee -a $LOG; }
(cd textslam_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -fsanitize=undefined,bounds -fno-omit-frame-pointer -Wno-unused-variable -o /tmp/libtsba_ubsan.so tsba.hip 2>/dev/null) || say "libtsba UBSan build: FAILED"
CUBSAN=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.ubsan_standalone-x86_64.so)
export UBSAN_OPTIONS=print_stacktrace=0:halt_on_error=0
TSBA_LIB=/tmp/libtsba_ubsan.so LD_PRELOAD="$CUBSAN" timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_global.py tests/test_gpu_context_reuse.py tests/test_gpu_far.py -m gpu -q -s -p no:cacheprovider > /tmp/san_gpu.log 2>&1
say "libtsba host code (clang UBSan + bounds), GPU parity tests through the instrumented host side: $(grep -E 'passed|failed' /tmp/san_gpu.log | tail -1) ; reports: $(grep -c 'runtime error' /tmp/san_gpu.log)"
grep -B2 -A12 'runtime error' /tmp/san_gpu.log | head -60 >> $LOG
# libtsorb.so's host side the same way (round 6's last session rewrote its launch plans: the few-frames pyramid's tilings, the fallback word in pinned memory, the results' pinned copy)
(cd textslam_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fsanitize=undefined,bounds -fno-omit-frame-pointer -Wno-unused-variable -o /tmp/libtsorb_ubsan.so tsorb.hip 2>/dev/null) || say "libtsorb UBSan build: FAILED"
TSORB_LIB=/tmp/libtsorb_ubsan.so LD_PRELOAD="$CUBSAN" timeout 900 python -m pytest tests/test_gpu_orb.py -m gpu -q -s -p no:cacheprovider > /tmp/san_gpu_orb.log 2>&1
say "libtsorb host code (clang UBSan + bounds), the ORB GPU tests through the instrumented host side: $(grep -E 'passed|failed' /tmp/san_gpu_orb.log | tail -1) ; reports: $(grep -c 'runtime error' /tmp/san_gpu_orb.log)"
grep -B2 -A12 'runtime error' /tmp/san_gpu_orb.log | head -40 >> $LOG
