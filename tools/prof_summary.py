"""summarise a rocprofv3 --kernel-trace sqlite/csv output dir into a per-kernel table (stdout)"""
import sys, glob, sqlite3, csv, collections, re
root = sys.argv[1]
rows = collections.defaultdict(lambda: [0, 0.0])
dbs = glob.glob(root + '/**/*.db', recursive=True)
if dbs:
    con = sqlite3.connect(dbs[0])
    cur = con.cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
    view = [n for n in names if n == 'kernels'] or [n for n in names if 'kernels' in n]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % view[0])]
    ncol = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    for name, s, e in cur.execute("select %s, start, end from %s" % (ncol, view[0])):
        rows[name][0] += 1
        rows[name][1] += (e - s) / 1e3
else:
    for f in glob.glob(root + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            rows[r['Kernel_Name']][0] += 1
            rows[r['Kernel_Name']][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in rows.values())
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = n.replace('__hip_bfloat16', 'bf16').replace('void ', '')
    return n[:150]
print('   calls   total_us    avg_us    pct  kernel')
for n, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:45]:
    print('%8d %10.1f %9.1f %6.1f  %s' % (c, t, t / c, 100 * t / tot, short(n)))
print('# total kernel time %.1f us' % tot)
