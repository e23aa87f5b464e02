"""Per-dispatch durations of the kernels whose name contains argv[2], grouped by grid size (rocprofv3 rocpd database argv[1])."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
if not cols:
    print("no `kernels` view; tables:", [r[0] for r in cur.execute("select name from sqlite_master")]); sys.exit(0)
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
if gx is None: print("columns:", cols); sys.exit(0)
dur = "duration" if "duration" in cols else "(end - start)"
print("%-50s %8s %6s %10s %10s" % ("kernel", gx, "calls", "avg_us", "min_us"))
for r in cur.execute("select name, %s, count(*), avg(%s), min(%s) from kernels where name like ? group by name, %s order by name, %s" % (gx, dur, dur, gx, gx), ("%" + pat + "%",)):
    print("%-50s %8d %6d %10.2f %10.2f" % (r[0][:50], r[1], r[2], r[3]/1e3, r[4]/1e3))
