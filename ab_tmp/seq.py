import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'kernel' in t.lower()][:20])
v = [t for t in tabs if t.lower() == 'kernels'] or [t for t in tabs if 'kernel_dispatch' in t.lower()]
print(v)
t = v[0]
cols = [r[1] for r in cur.execute(f"pragma table_info({t})")]
print(cols)
