"""per-grid durations of the wgrad_reduce / colsum launches in a rocprofv3 kernel trace (sqlite)"""
import sys, glob, sqlite3, collections
db = glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0]
cur = sqlite3.connect(db).cursor()
names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
view = [n for n in names if n == 'kernels'] or [n for n in names if 'kernels' in n]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % view[0])]
print(cols)
ncol = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
gcol = [c for c in cols if 'grid' in c.lower()]
wcol = [c for c in cols if 'workgroup' in c.lower() or 'block' in c.lower()]
sel = ", ".join([ncol, "start", "end"] + gcol + wcol)
agg = collections.defaultdict(lambda: [0, 0.0])
for row in cur.execute("select %s from %s" % (sel, view[0])):
    n = row[0]
    if 'wgrad_reduce' in n or 'colsum' in n or 'zero_kernel' in n:
        key = (n.split('(')[0][-40:],) + tuple(row[3:])
        agg[key][0] += 1
        agg[key][1] += (row[2] - row[1]) / 1e3
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%6d %9.1f %8.1f  %s" % (c, t, t / c, k))
