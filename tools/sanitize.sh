#!/bin/bash
# Sanitizer pass over this repository's own host code (aux row "race / memory checking"):
#   1. the CPU oracle (plain C) under gcc ASan + UBSan, driven by its whole CPU test-suite;
#   2. the adapter gather / scatter templates + the C++ ABI driver under g++ ASan + UBSan (six problem types);
#   3. the HOST side of libtsba.so (plan builder, reordering, index validation, upload staging, C ABI) under clang ASan + UBSan --
#      the CPU tests here (the leg that needs a device -- the GPU parity tests through a UBSan build of the host side -- is tools/sanitize_gpu.sh:
#      this file carries the memory-error sanitizer's flags, which the GPU pool refuses to run, and is listed in .gpurunignore);
#   4. the plan builder's host threads under clang TSan.
# Usage: bash tools/sanitize.sh   -> gpurun_out/r06_sanitizers.log (summary lines "SANITIZE <what>: <result>"); runs in the build container (no GPU)
set -u
cd "$(dirname "$0")/.."
MODE=${1:-}
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/r06_sanitizers.log; [ "$MODE" = "gpuonly" ] && LOG=$OUT/r06_sanitizers_gpu.log; : > $LOG
GASAN=$(gcc -print-file-name=libasan.so); GUBSAN=$(gcc -print-file-name=libubsan.so)
CASAN=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=0:halt_on_error=0
say() { echo "SANITIZE $*" | tee -a $LOG; }

if [ "$MODE" != "gpuonly" ]; then
# ---- 1. oracle
mkdir -p /tmp/san_oracle && cp oracle/*.so /tmp/san_oracle/ 2>/dev/null
for f in tsba tsorb tsframe tsloop; do gcc -O1 -g -fPIC -std=c11 -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o oracle/lib${f}_oracle.so oracle/${f}_oracle.c -lm || say "oracle build $f: FAILED"; done
LD_PRELOAD="$GASAN $GUBSAN" python -m pytest tests/test_oracle.py tests/test_orb_oracle.py tests/test_frame_oracle.py tests/test_loop_oracle.py tests/test_golden.py tests/test_multi_gpu_sharding.py -q -s -p no:cacheprovider > /tmp/san_oracle.log 2>&1
say "oracle (gcc ASan+UBSan, CPU suite): $(grep -E 'passed|failed' /tmp/san_oracle.log | tail -1) ; reports: $(grep -c 'ERROR: AddressSanitizer\|runtime error' /tmp/san_oracle.log)"
cp /tmp/san_oracle/*.so oracle/ 2>/dev/null; make -s -C oracle

# ---- 2. adapter + C++ ABI driver
g++ -std=c++11 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude -Iadapter -o /tmp/abi_from_cxx_san tests/cxx/abi_from_cxx.cpp -Ltextslam_amd -ltsba -L/opt/rocm/lib -Wl,-rpath,$PWD/textslam_amd -Wl,-rpath,/opt/rocm/lib || say "adapter build: FAILED"
python - > /tmp/san_adapter.log 2>&1 <<'PY'
import subprocess, sys, os, tempfile
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_cxx_adapter as t
bad = 0
for name, (P, mode) in t._cases().items():
    d = tempfile.mkdtemp(); dump, out = os.path.join(d, "p.bin"), os.path.join(d, "o.bin")
    t._write_dump(dump, P)
    # (the device is hidden: the gather / scatter code is what is under test here, and the HIP runtime does not start under ASan)
    r = subprocess.run(["/tmp/abi_from_cxx_san", dump, mode, out], capture_output=True, text=True, env=dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1"))
    ok = r.returncode in (0, 3) and "gather identical" in r.stdout and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr
    print(name, "rc", r.returncode, "OK" if ok else "BAD\n" + r.stderr[-2000:]); bad += not ok
print("bad", bad)
PY
say "adapter gather/scatter + C++ driver (g++ ASan+UBSan, 6 problem types): $(tail -1 /tmp/san_adapter.log)"
# the sliding-window gather with the segment cache (round 5): six windows, a mutated graph, invalidate() -- C4-sized
python - > /tmp/san_slide.log 2>&1 <<'PY'
import subprocess, sys, os, tempfile
sys.path.insert(0, ".")
from textslam_amd import synth, abi
d = tempfile.mkdtemp(); dump = os.path.join(d, "c4.bin")
abi.write_dump(dump, synth.config_c4())
r = subprocess.run(["/tmp/abi_from_cxx_san", dump, "slide_check", os.path.join(d, "o.bin")], capture_output=True, text=True, env=dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1"))
ok = r.returncode == 0 and "gathers identical" in r.stdout and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr
print(r.stdout.strip()); print("OK" if ok else "BAD\n" + r.stderr[-2000:])
PY
say "adapter gather cache, sliding window + mutated graph (g++ ASan+UBSan): $(tail -1 /tmp/san_slide.log)"
g++ -std=c++11 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude -Iadapter -o /tmp/loop_from_cxx_san tests/cxx/loop_from_cxx.cpp -Ltextslam_amd -ltsloop -L/opt/rocm/lib -Wl,-rpath,$PWD/textslam_amd -Wl,-rpath,/opt/rocm/lib || say "loop adapter build: FAILED"
python - > /tmp/san_loop_adapter.log 2>&1 <<'PY'
import subprocess, sys, os, tempfile
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_cxx_adapter as t
d = tempfile.mkdtemp(); bad = 0
m = t._sim3_case()
t._put_records(os.path.join(d, "s.bin"), [("P1", m["P1"], 0), ("P2", m["P2"], 0), ("uv1", m["uv1"], 0), ("uv2", m["uv2"], 0), ("inliers", m["inliers"], 2), ("sim0", m["sim0"], 0), ("K", m["K"], 0)])
g = t._loop_case(); t._write_loop_dump(os.path.join(d, "l.bin"), g)
for mode, f in (("sim3", "s.bin"), ("loop", "l.bin")):
    r = subprocess.run(["/tmp/loop_from_cxx_san", os.path.join(d, f), mode, os.path.join(d, "o.bin")], capture_output=True, text=True, env=dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1"))
    ok = r.returncode in (0, 3) and "gather" in r.stdout and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr
    print(mode, "rc", r.returncode, "OK" if ok else "BAD\n" + r.stderr[-2000:]); bad += not ok
    if ok and mode == "loop": t._check_loop_gather(t._read_out(os.path.join(d, "o.bin")), g)
print("bad", bad)
PY
say "loop-closing adapter gather + C++ driver (g++ ASan+UBSan, sim3 + pose graph): $(tail -1 /tmp/san_loop_adapter.log)"

# ---- 3. host side of libtsba
fi
if [ "$MODE" != "gpuonly" ]; then
(cd textslam_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -Wno-unused-variable -o /tmp/libtsba_san.so tsba.hip) || say "libtsba build: FAILED"
TSBA_LIB=/tmp/libtsba_san.so LD_PRELOAD="$CASAN" HIP_VISIBLE_DEVICES=-1 ROCR_VISIBLE_DEVICES=-1 python -m pytest tests/test_band_partition.py tests/test_abi.py -q -s -p no:cacheprovider > /tmp/san_host.log 2>&1
say "libtsba host code (clang ASan+UBSan), CPU tests (plan, reordering, partition tables, ABI): $(grep -E 'passed|failed' /tmp/san_host.log | tail -1) ; reports: $(grep -c 'ERROR: AddressSanitizer\|runtime error' /tmp/san_host.log)"
fi
if [ "$MODE" != "gpuonly" ]; then
# ---- 4. the plan builder's host threads (fork-join pool, shared key bitmaps, atomic min / max, stable bucket placement) under clang TSan:
#         every plan of tools/diag/plan_checksums.py (windows, maps, loop closures, text planes, shards) built with 1, 3 and 16 threads
CTSAN=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.tsan-x86_64.so)
(cd textslam_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -fsanitize=thread -fno-omit-frame-pointer -Wno-unused-variable -Wno-unused-value -o /tmp/libtsba_tsan.so tsba.hip 2>/dev/null) || say "libtsba TSan build: FAILED"
TSBA_LIB=/tmp/libtsba_tsan.so LD_PRELOAD="$CTSAN" HIP_VISIBLE_DEVICES=-1 ROCR_VISIBLE_DEVICES=-1 TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" timeout 1500 python tools/diag/plan_checksums.py > /tmp/san_tsan.txt 2> /tmp/san_tsan.err
python tools/diag/plan_checksums.py > /tmp/san_plain.txt 2>/dev/null
say "libtsba plan builder (clang TSan, 1 / 3 / 16 host threads): $(wc -l < /tmp/san_tsan.txt) plans built, $(diff /tmp/san_plain.txt /tmp/san_tsan.txt | grep -c '^[<>]') checksum differences against the plain build ; reports: $(grep -c 'WARNING: ThreadSanitizer' /tmp/san_tsan.err)"
grep -A12 'WARNING: ThreadSanitizer' /tmp/san_tsan.err | head -40 >> $LOG
grep -B2 -A12 'ERROR: AddressSanitizer\|runtime error' /tmp/san_oracle.log /tmp/san_host.log 2>/dev/null | head -80 >> $LOG
fi
cat $LOG | grep SANITIZE
