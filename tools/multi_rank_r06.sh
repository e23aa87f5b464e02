#!/bin/bash
# Round 6, verdict item 7: per-rank kernel time of ONE sharded global BA at N = 1 / 2 / 4 / 8 in-process ranks on one GPU, on maps where sharding has something
# to shard: 5000 keyframes with ~500 observations each, and C5 with its 1000 text planes.  Output: gpurun_out/r06_ranks_<map>_w<N>{.jsonl,_kernel_stats.txt}
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for map in "$@"; do for w in 1 2 4 8; do
  n=r06_ranks_${map}_w$w
  rm -rf /tmp/prof_$n; ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$n -o $n -- python $OLDPWD/tools/diag/gpu_multi_rank.py $w $map 2 > $OLDPWD/gpurun_out/$n.jsonl 2> /tmp/prof_$n.err )
  python profiles/rocpd_top_kernels.py $(find /tmp/prof_$n -name "*.db" | head -1) > gpurun_out/${n}_kernel_stats.txt 2>&1
  echo "== $n: $(head -c 300 gpurun_out/$n.jsonl)"; tail -3 /tmp/prof_$n.err | head -3
  head -4 gpurun_out/${n}_kernel_stats.txt
done; done
