"""compare two per-launch GEMM tables written by `bench.py --detail`:  python scratch/cmp_detail.py old.txt new.txt [min_us]"""
import sys

def load(fn):
    d = {}
    for l in open(fn).read().splitlines()[1:]:
        p = l.split(None, 4)
        d[p[4]] = (float(p[0]), float(p[1]), int(p[2]), float(p[3]))
    return d

a, b = load(sys.argv[1]), load(sys.argv[2])
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
tot = {}
rows = []
for k in sorted(set(a) | set(b)):
    ta, tb = a.get(k, (0, 0, 0, 0))[0], b.get(k, (0, 0, 0, 0))[0]
    kind = k.split()[0]
    t = tot.setdefault(kind, [0.0, 0.0])
    t[0] += ta; t[1] += tb
    rows.append((tb - ta, k, ta, tb))
rows.sort()
for d, k, ta, tb in rows:
    if abs(d) >= thr:
        print("%-72s %8.1f -> %8.1f  %+8.1f" % (k, ta, tb, d))
for kind, (ta, tb) in tot.items():
    print("%-8s %9.1f -> %9.1f  %+8.1f us" % (kind, ta, tb, tb - ta))
