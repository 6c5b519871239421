import sqlite3, glob, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print(cols)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
t0 = rows[0][1]
last_end = {}
for n, s, e in rows[-40:]:
    short = n.split("(")[0].split("::")[-1]
    print("%-16s start %10.3f ms  dur %8.3f ms" % (short, (s - t0) / 1e6, (e - s) / 1e6))
