#!/bin/bash
# GPU-side kernel timeline of one single-frame ORB extraction (rocprofv3 --kernel-trace, csv)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
python tools/diag/gpu_orb_single_loop.py
rm -rf /tmp/prof_os; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_os -o os -- python $OLDPWD/tools/diag/gpu_orb_single_loop.py > /tmp/run_os.txt 2> /tmp/prof_os.err )
python - <<'P'
import csv, glob
ev = []
for f in glob.glob("/tmp/prof_os/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50]))
ev.sort()
first = lambda n: n.startswith("k_level0") or "k_pyramid_one" in n
starts = [i for i, e in enumerate(ev) if first(e[2]) and (i == 0 or not first(ev[i-1][2]))]
i0, i1 = starts[12], starts[13]
base = ev[i0][0]; prev = base
for s, e, n in ev[i0:i1]:
    print("%8.1f us  gap %5.1f  dur %6.1f  %s" % ((s - base)/1e3, (s - prev)/1e3, (e - s)/1e3, n)); prev = e
print("total %.1f us" % ((ev[i1-1][1] - base)/1e3))
P
