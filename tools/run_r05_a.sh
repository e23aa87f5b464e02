#!/bin/bash
# round 5, GPU call A: the whole -m gpu suite, the window A/B (production vs the round-4 paths), the default bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 2>&1 | tail -80 > gpurun_out/r05a_tests.log
tail -5 gpurun_out/r05a_tests.log
timeout 300 python tools/diag/gpu_ab_window.py > gpurun_out/r05a_ab.log 2>&1
tail -12 gpurun_out/r05a_ab.log
timeout 600 python bench.py > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err
head -c 1500 gpurun_out/r05a_bench.json
