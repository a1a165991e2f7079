"""GPU timeline of the last training step in a rocprofv3 --kernel-trace DB: busy/idle time, overlap, and
the kernels that run with few workgroups (serialisation points).  usage: timeline.py <dir>"""
import sys, glob, sqlite3, re, collections
db = glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0]
cur = sqlite3.connect(db).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = list(cur.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, queue_id from kernels order by start"))
# last step = from the last sgd_kernel backwards to the previous one
sgd = [i for i, r in enumerate(rows) if 'sgd_kernel' in r[0]]
lo, hi = sgd[-3] + 1, sgd[-2] + 1          # a full step not touched by the event-bracketed one
step = rows[lo:hi]
t0, t1 = step[0][1], max(r[2] for r in step)
print('kernels in step: %d, span %.2f ms' % (len(step), (t1 - t0) / 1e6))
ev = []
for r in step:
    ev.append((r[1], 1)); ev.append((r[2], -1))
ev.sort()
act, last, busy, over = 0, t0, 0, 0
for t, d in ev:
    if act >= 1: busy += t - last
    if act >= 2: over += t - last
    act += d; last = t
print('busy %.2f ms, idle %.2f ms, >=2 kernels in flight %.2f ms' % (busy / 1e6, (t1 - t0 - busy) / 1e6, over / 1e6))
byq = collections.defaultdict(float)
for r in step: byq[r[7]] += (r[2] - r[1]) / 1e6
print('kernel time per queue (ms):', dict(byq))
small = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    wgs = (r[3] // max(r[6], 1)) * max(r[4], 1) * max(r[5], 1)
    if wgs < 256:
        n = re.sub(r'\(anonymous namespace\)::|vlfb::|void ', '', r[0]).split('(')[0][:70]
        small[n][0] += 1; small[n][1] += (r[2] - r[1]) / 1e3
print('kernels launched with < 256 workgroups:')
for n, (c, t) in sorted(small.items(), key=lambda kv: -kv[1][1])[:14]:
    print('  %8.1f us  x%-3d %s' % (t, c, n))
# gaps on the timeline
gaps = []
ends = t0
for r in sorted(step, key=lambda r: r[1]):
    if r[1] > ends: gaps.append((r[1] - ends, r[0]))
    ends = max(ends, r[2])
gaps.sort(reverse=True)
print('largest idle gaps (us, next kernel):')
for g, n in gaps[:8]:
    print('  %7.1f  %s' % (g / 1e3, re.sub(r'\(anonymous namespace\)::|vlfb::|void ', '', n)[:80]))
