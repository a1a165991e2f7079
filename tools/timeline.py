"""GPU timeline of one steady-state training step from a rocprofv3 --kernel-trace DB:
phases (forward / backward / solver), busy vs idle, per-queue kernel time and the per-kernel table of
that step.  usage: tools/timeline.py <dir> [out.txt] [k = which step from the end]"""
import sys, glob, sqlite3, re, collections
db = glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, queue_id from kernels order by start"))
short = lambda n: re.sub(r'\(anonymous namespace\)::|vlfb::|void |unsigned short|__hip_bfloat16', lambda m: {'unsigned short': 'bf16'}.get(m.group(0), ''), n).split('(')[0][:86]
sgd = [i for i, r in enumerate(rows) if 'sgd_kernel' in r[0]]
k = int(sys.argv[3]) if len(sys.argv) > 3 else 3
lo, hi = sgd[-k - 1] + 1, sgd[-k] + 1
step = rows[lo:hi]
t0, t1 = step[0][1], max(r[2] for r in step)
out = []
P = out.append
P('kernels in step: %d, span %.3f ms' % (len(step), (t1 - t0) / 1e6))
loss = [i for i, r in enumerate(step) if 'sigmoid_ce' in r[0] or 'softmax_ce' in r[0]]
if loss:
    tl = step[loss[0]][2]
    P('forward %.3f ms | backward+solver %.3f ms' % ((tl - t0) / 1e6, (t1 - tl) / 1e6))
ev = []
for r in step:
    ev.append((r[1], 1)); ev.append((r[2], -1))
ev.sort()
act, last, busy, over = 0, t0, 0, 0
for t, d in ev:
    if act >= 1: busy += t - last
    if act >= 2: over += t - last
    act += d; last = t
P('busy %.3f ms, idle %.3f ms, >=2 kernels in flight %.3f ms, sum of kernel durations %.3f ms' % (
    busy / 1e6, (t1 - t0 - busy) / 1e6, over / 1e6, sum(r[2] - r[1] for r in step) / 1e6))
byq = collections.defaultdict(float)
for r in step: byq[r[7]] += (r[2] - r[1]) / 1e6
P('kernel time per queue (ms): %s' % dict(byq))
agg = collections.OrderedDict()
for r in step:
    a = agg.setdefault(short(r[0]), [0, 0.0])
    a[0] += 1; a[1] += (r[2] - r[1]) / 1e3
P('%7s %9s %8s  kernel' % ('calls', 'total_us', 'avg_us'))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    P('%7d %9.1f %8.1f  %s' % (c, t, t / c, n))
# which kernels occupy which queue in which phase (the main queue carries the forward and the DGRAD chain)
if loss:
    for q in sorted(byq, key=lambda q: -byq[q]):
        for phase, sel in (('forward', lambda r: r[1] < tl), ('backward', lambda r: r[1] >= tl)):
            part = [r for r in step if r[7] == q and sel(r)]
            if not part:
                continue
            tot = sum(r[2] - r[1] for r in part) / 1e3
            a2 = collections.OrderedDict()
            for r in part:
                e = a2.setdefault(short(r[0]).split('<')[0], [0, 0.0])
                e[0] += 1; e[1] += (r[2] - r[1]) / 1e3
            P('queue %s, %s: %d kernels, %.1f us: %s' % (q, phase, len(part), tot, ', '.join(
                '%s x%d %.0f' % (n, c, t) for n, (c, t) in sorted(a2.items(), key=lambda kv: -kv[1][1])[:18])))
# backward: when is only ONE of the queues busy (the head before any parameter gradient exists, the tail of either queue)?
if loss and len(byq) >= 2:
    qs = sorted(byq, key=lambda q: -byq[q])[:2]
    evq = []
    for r in step:
        if r[1] >= tl and r[7] in qs and 'sgd_kernel' not in r[0]:
            evq.append((r[1], 1, r[7])); evq.append((r[2], -1, r[7]))
    evq.sort()
    actq = {q: 0 for q in qs}
    only = {q: 0 for q in qs}
    both = 0
    lastt = tl
    for t, d, q in evq:
        a, b = actq[qs[0]] > 0, actq[qs[1]] > 0
        if a and b: both += t - lastt
        elif a: only[qs[0]] += t - lastt
        elif b: only[qs[1]] += t - lastt
        actq[q] += d; lastt = t
    lastk = {q: max(r[2] for r in step if r[7] == q and r[1] >= tl and 'sgd_kernel' not in r[0]) for q in qs}
    P('backward: both queues busy %.3f ms, only queue %s %.3f ms, only queue %s %.3f ms; last kernel before the solver ends at +%.3f ms (queue %s) / +%.3f ms (queue %s)' % (
        both / 1e6, qs[0], only[qs[0]] / 1e6, qs[1], only[qs[1]] / 1e6, (lastk[qs[0]] - tl) / 1e6, qs[0], (lastk[qs[1]] - tl) / 1e6, qs[1]))
gaps = []
ends = t0
for r in sorted(step, key=lambda r: r[1]):
    if r[1] > ends: gaps.append((r[1] - ends, r[0]))
    ends = max(ends, r[2])
P('idle gaps: n=%d total %.1f us; largest:' % (len(gaps), sum(g for g, _ in gaps) / 1e3))
for g, n in sorted(gaps, reverse=True)[:10]:
    P('  %7.1f us before %s' % (g / 1e3, short(n)))
txt = '\n'.join(out)
print(txt)
if len(sys.argv) > 2: open(sys.argv[2], 'w').write(txt + '\n')
