#!/bin/bash
# round 5, GPU call E: ORB tests + A/B of the pyramid tail, polling tests, ORB kernel statistics
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_orb.py tests/test_gpu_polling.py tests/test_gpu_frame.py -m gpu -q -p no:cacheprovider --timeout 600 2>&1 | tail -40 > $OUT/r05e_tests.log
tail -6 $OUT/r05e_tests.log
python - <<'PY' > $OUT/r05e_orb_ab.txt 2>&1
import time, numpy as np
from textslam_amd.orbextractor import ORBextractor, synthetic_frame
import torch
imgs = np.stack([synthetic_frame(s) for s in range(64)])
for n in (64, 1):
    for rnd in range(2):
        for mode in (0, 1):
            ex = ORBextractor(1000, 1.2, 8, 20, 7, device=0); ex.debug_set(pyramid_launches=mode); ex.upload(imgs[:n])
            for _ in range(3): ex.run()
            ts = []
            for _ in range(30):
                t0 = time.perf_counter(); ex.run(); ts.append((time.perf_counter() - t0)*1e3)
            print(f"frames {n:2d}  pyramid_launches {mode}: median {np.median(ts):.4f} ms  min {min(ts):.4f} ms", flush=True)
            ex.close()
PY
cat $OUT/r05e_orb_ab.txt
name=orb_batch64
rm -rf /tmp/prof_$name; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- python $OLDPWD/bench.py --workload orb --steps 20 --warmup 3 --no-cpu-baseline > $OLDPWD/$OUT/r05e_${name}_bench_under_rocprof.json 2> /tmp/prof_$name.err )
python profiles/rocpd_top_kernels.py $(find /tmp/prof_$name -name "*.db" | head -1) > $OUT/r05e_${name}_kernel_stats.txt 2>&1
cat $OUT/r05e_${name}_kernel_stats.txt | head -16
