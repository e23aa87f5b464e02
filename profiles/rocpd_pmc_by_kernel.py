"""Summarise one rocprofv3 --pmc pass (rocpd sqlite): per (kernel, grid size) the mean counter value per dispatch.
usage: python rocpd_pmc_by_kernel.py <results.db> [kernel-substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); pat = sys.argv[2] if len(sys.argv) > 2 else ""
q = ("select kernel_name, grid_size, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
     "where kernel_name like ? group by kernel_name, grid_size, counter_name order by kernel_name, grid_size")
print("%-44s %9s %-12s %6s %14s %14s %14s" % ("kernel", "grid", "counter", "n", "mean", "min", "max"))
for k, g, c, n, a, lo, hi in db.execute(q, ("%" + pat + "%",)):
    print("%-44s %9d %-12s %6d %14.2f %14.2f %14.2f" % (k[:44], g, c, n, a, lo, hi))
