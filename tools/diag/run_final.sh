set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/gpu_tests.log
grep -E "passed|failed" $OUT/gpu_tests.log | tail -1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
stats() { local name=$1; shift
  rm -rf /tmp/prof_$name; rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- python bench.py "$@" --no-cpu-baseline > $OUT/r02_${name}_bench_under_rocprof.json 2> /tmp/prof_$name.err
  python profiles/rocpd_top_kernels.py $(find /tmp/prof_$name -name "*.db" | head -1) > $OUT/r02_${name}_kernel_stats.txt 2>&1; }
stats c6_global_ba --workload global_ba --steps 3 --warmup 1
python profiles/rocpd_kernel_by_grid.py $(find /tmp/prof_c6_global_ba -name "*.db" | head -1) k_cre > $OUT/r02_c6_cre_by_level.txt 2>&1
stats c6_ring_global_ba --workload global_ba --loop --steps 3 --warmup 1
stats c6_loop_tail_global_ba --workload global_ba --loop --loop-at 1500 --steps 3 --warmup 1
python bench.py > $OUT/r02_bench_c4.json 2> /tmp/b1.err
python bench.py --workload global_ba > $OUT/r02_bench_c6.json 2> /tmp/b2.err
python bench.py --workload global_ba --loop --no-cpu-baseline > $OUT/r02_bench_c6_ring.json 2> /tmp/b3.err
python bench.py --workload global_ba --loop --loop-at 1500 --no-cpu-baseline > $OUT/r02_bench_c6_loop_tail.json 2> /tmp/b4.err
for f in c4 c6 c6_ring c6_loop_tail; do python -c "
import json
d=json.loads(open('$OUT/r02_bench_$f.json').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],3), round(d['value']/1e9,3), round(d['roofline']['frac'],4))"; done
head -8 $OUT/r02_c6_global_ba_kernel_stats.txt
