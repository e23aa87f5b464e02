"""Durations of the dispatches of one kernel in launch order (rocprofv3 rocpd database): rocpd_kernel_sequence.py <db> <name substring> [count]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pat = sys.argv[2]; n = int(sys.argv[3]) if len(sys.argv) > 3 else 16
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t and "rocpd_kernel_dispatch" in t]
ks = [t for t in tabs if "kernel_symbol" in t or "info_kernel_symbol" in t]
try:
    rows = list(cur.execute("select name, start, end from kernels where name like ? order by start", ("%" + pat + "%",)))
except Exception as e:
    print("tables:", tabs); raise
print("%d dispatches of *%s*; last %d (us):" % (len(rows), pat, n))
print(" ".join("%.1f" % ((r[2] - r[1])/1e3) for r in rows[-n:]))
