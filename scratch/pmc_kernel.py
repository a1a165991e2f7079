"""per-kernel means of the counters found in rocprofv3 --pmc output dirs: pmc_kernel.py <dir> [<dir> ...]"""
import sys, glob, sqlite3, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for root in sys.argv[1:]:
    for db in glob.glob(root + '/**/*.db', recursive=True):
        cur = sqlite3.connect(db).cursor()
        q = "select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection group by dispatch_id, counter_name"
        for name, cname, _, v in cur.execute(q):
            a = acc[name][cname]
            a[0] += 1
            a[1] += v
for name, cs in acc.items():
    if not any(k in name for k in ('gemm', 'wgrad_rows', 'pool', 'stem_', 'conv_rows64')):
        continue
    short = re.sub(r'\(anonymous namespace\)::|vlfb::', '', name).replace('unsigned short', 'bf16')[:110]
    print(short)
    for c, (n, v) in sorted(cs.items()):
        print('    %-28s %16.1f  (mean of %d dispatches)' % (c, v / n, n))
