#!/bin/bash
# GPU-side timeline (kernels + copies, rocprofv3 traces) of the last one-shot tsba_pose_optim calls of tools/diag/gpu_diag_pose_cold.py
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
rm -rf /tmp/prof_tl; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d /tmp/prof_tl -o tl -- python $OLDPWD/${1:-tools/diag/gpu_diag_pose_cold.py} > /tmp/run_tl.txt 2> /tmp/prof_tl.err )
tail -3 /tmp/run_tl.txt
python - <<'P'
import csv, glob
ev = []
for f in glob.glob("/tmp/prof_tl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:60]))
for f in glob.glob("/tmp/prof_tl/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "?"))))
for f in glob.glob("/tmp/prof_tl/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Function"]
        if n.startswith("hipGetLastError") or n.startswith("__hip") or n in ("hipGetDevice", "hipSetDevice", "hipPeekAtLastError"): continue
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "  A " + n))
ev.sort()
# one call: from the slab clearing of an upload (hipMemsetAsync after a stream synchronisation) to the next one
ms = [e[0] for e in ev if e[2] == "  A hipMemsetAsync"]
starts = [ms[i] for i in range(len(ms)) if i == 0 or ms[i] - ms[i-1] > 500000]
t0, t1 = (starts[-2] - 30000, starts[-1] - 30000) if len(starts) >= 2 else (ev[0][0], ev[-1][1])
base = None; nk = 0
for s, e, n in ev:
    if s < t0 or s > t1 or n == "  A hipLaunchKernel": continue
    if base is None: base = s
    if n.startswith("K void k_") or n.startswith("K k_decide"):
        nk += 1
        if nk > 8: continue
    print("%9.1f us  +%7.1f us  %s" % ((s - base)/1e3, (e - s)/1e3, n))
P
