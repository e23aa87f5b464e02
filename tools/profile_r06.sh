#!/bin/bash
# Round-6 profiles on the GPU box (one call; the r05 script + C5's PMC pass + the *_pmc_traffic.json files written from the text files): per-kernel statistics (rocprofv3 --kernel-trace --stats) of the bench workloads and of the maps with long-range
# coupling / two closures, then FETCH_SIZE / WRITE_SIZE in separate --pmc passes (never combined with other trace domains) for the linearisation kernel of
# C4 and C6 and for the whole ORB pipeline (its last PMC pass was round 3).  Output: gpurun_out/r06_*.txt (the cited ones are copied to profiles/).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
stats() {   # name, bench args...
  local name=$1; shift
  rm -rf /tmp/prof_$name; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- python $OLDPWD/bench.py "$@" --no-cpu-baseline --no-also > $OLDPWD/$OUT/r06_${name}_bench_under_rocprof.json 2> /tmp/prof_$name.err )
  python profiles/rocpd_top_kernels.py $(find /tmp/prof_$name -name "*.db" | head -1) > $OUT/r06_${name}_kernel_stats.txt 2>&1
}
diag() {    # name, gpu_diag_far args...
  local name=$1; shift
  rm -rf /tmp/prof_$name; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- python $OLDPWD/tools/diag/gpu_diag_far.py "$@" > $OLDPWD/$OUT/r06_${name}_run.txt 2> /tmp/prof_$name.err )
  python profiles/rocpd_top_kernels.py $(find /tmp/prof_$name -name "*.db" | head -1) > $OUT/r06_${name}_kernel_stats.txt 2>&1
}
pmc() {     # name, counter, kernel filter, bench args...
  local name=$1 ctr=$2 filt=$3; shift 3
  rm -rf /tmp/pmc_$name; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_$name -o $name -- python $OLDPWD/bench.py "$@" --no-cpu-baseline --no-also > /dev/null 2> /tmp/pmc_$name.err )
  python profiles/rocpd_pmc_by_kernel.py $(find /tmp/pmc_$name -name "*.db" | head -1) $filt > $OUT/r06_${name}.txt 2>&1
}
ONLY=${1:-all}      # "orb": just the ORB legs (after a change of that pipeline)
if [ "$ONLY" = all ]; then
stats c4_local_ba --steps 20 --warmup 3
stats c6_global_ba --workload global_ba --steps 3 --warmup 1
fi
stats orb_batch64 --workload orb --steps 20 --warmup 3
if [ "$ONLY" = all ]; then
diag c6_long_range 5000 70000 0.01 2
diag c6_two_closures 5000 70000 0.0 2 2
pmc c4_pmc_fetch FETCH_SIZE k_linearize --steps 3 --warmup 1
pmc c4_pmc_write WRITE_SIZE k_linearize --steps 3 --warmup 1
pmc c6_pmc_fetch FETCH_SIZE k_linearize --workload global_ba --steps 2 --warmup 1
pmc c6_pmc_write WRITE_SIZE k_linearize --workload global_ba --steps 2 --warmup 1
# round 6: C5 (500 KF x 50 k points) had no PMC pass: its roofline.traffic was null
stats c5_global_ba --workload global_ba --kf 500 --pts 50000 --band 12 --steps 5 --warmup 1
pmc c5_pmc_fetch FETCH_SIZE k_linearize --workload global_ba --kf 500 --pts 50000 --band 12 --steps 2 --warmup 1
pmc c5_pmc_write WRITE_SIZE k_linearize --workload global_ba --kf 500 --pts 50000 --band 12 --steps 2 --warmup 1
# the JSONs bench.py reads, from the text files just written (copy both to profiles/)
python tools/pmc_traffic_json.py $OUT/r06_c4_pmc_fetch.txt $OUT/r06_c4_pmc_write.txt k_linearize $OUT/r06_c4_pmc_traffic.json "k_linearize<FULL> level 0"
python tools/pmc_traffic_json.py $OUT/r06_c6_pmc_fetch.txt $OUT/r06_c6_pmc_write.txt k_linearize $OUT/r06_c6_linearize_pmc_traffic.json "k_linearize<FULL,4> level 0, 5000 keyframes"
python tools/pmc_traffic_json.py $OUT/r06_c5_pmc_fetch.txt $OUT/r06_c5_pmc_write.txt k_linearize $OUT/r06_c5_linearize_pmc_traffic.json "k_linearize<FULL,4> level 0, 500 keyframes x 50 k points"
sed -i 's#gpurun_out/#profiles/#g' $OUT/r06_*_pmc_traffic.json
fi
pmc orb_pmc_fetch FETCH_SIZE "" --workload orb --steps 2 --warmup 1
pmc orb_pmc_write WRITE_SIZE "" --workload orb --steps 2 --warmup 1
python tools/pmc_traffic_json.py $OUT/r06_orb_pmc_fetch.txt $OUT/r06_orb_pmc_write.txt "" $OUT/r06_orb_pmc_traffic.json "whole ORB pipeline, one batch of 64 frames (sum over its kernels' mean per dispatch)" sum
sed -i 's#gpurun_out/#profiles/#g' $OUT/r06_orb_pmc_traffic.json
# the per-frame call: kernel timeline of ONE resident frame
bash tools/diag/timeline_orb_single.sh > $OUT/r06_orb_one_frame_timeline.txt 2>&1
ls -la $OUT/r06_* | head -60
