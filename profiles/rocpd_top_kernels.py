import sqlite3, sys
db=sqlite3.connect(sys.argv[1]); cur=db.cursor()
print("%-60s %7s %12s %10s %6s"%("kernel","calls","total_us","avg_us","%"))
tot = 0.0
for r in cur.execute("select * from top_kernels"):
    tot += r[2]/1e3 if r[2]>1e6 else r[2]
    print("%-60s %7d %12.1f %10.2f %6.2f"%(r[0][:60],r[1],r[2]/1e3 if r[2]>1e6 else r[2],r[3]/1e3 if r[2]>1e6 else r[3],r[4]))
print("%-60s %7s %12.1f" % ("TOTAL (all kernels)", "", tot))
