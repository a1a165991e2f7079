"""The vector-memory path (TA -> TCP (L1) -> TCC (L2)) per kernel family from rocprofv3 PMC passes of bench.py: what the LDS-DMA
kernels wait for.  Any number of pass directories (each a `rocprofv3 --kernel-trace --pmc <counters> GRBM_GUI_ACTIVE` run); the
raw per-launch sums of every collected counter are printed per family and per kernel instance, plus the ratios that can be formed:
  TA busy            TA_BUSY_avr / GRBM_GUI_ACTIVE (both averaged / summed the same way by rocprofv3: a plain ratio of the sums)
  L2 read latency    TCP_TCC_READ_REQ_LATENCY_sum / TCP_TCC_READ_REQ_sum  (cycles per L1 -> L2 read request)
  stall shares       TA_ADDR_STALLED_BY_TC / TA_DATA_STALLED_BY_TC / TCP_PENDING_STALL cycles per GUI-active cycle and TA / TCP instance
usage: tools/pmc_mem.py <out.txt> "<command line>" <dir> [<dir> ...]"""
import sys, glob, sqlite3, collections, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import family

out_txt, cmd = sys.argv[1:3]
fam = collections.defaultdict(lambda: collections.defaultdict(float))
per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
gui_of = {}
for root in sys.argv[3:]:
    db = glob.glob(root + "/**/*.db", recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    q = "select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection group by dispatch_id, counter_name"
    rows = list(cur.execute(q))
    gui = {did: v for _, cname, did, v in rows if cname == "GRBM_GUI_ACTIVE"}
    for name, cname, did, v in rows:
        k = family(name)
        if k is None or cname == "GRBM_GUI_ACTIVE":
            continue
        short = name.replace("(anonymous namespace)::", "").replace("vlfb::", "").replace("void ", "").split("(")[0][:100]
        for d in (fam[k], per_kernel[(k, short)]):
            d[cname] += v
            d["GUI@" + cname] += gui.get(did, 0.0)        # GUI-active cycles of the launches this counter was collected on


def line(c):
    parts = []
    names = sorted(n for n in c if not n.startswith("GUI@"))
    for n in names:
        g = c.get("GUI@" + n, 0.0)
        parts.append("%s %.4g (%.3f per GUI cycle)" % (n, c[n], c[n] / g if g else 0.0))
    if c.get("TCP_TCC_READ_REQ_sum") and c.get("TCP_TCC_READ_REQ_LATENCY_sum"):
        parts.append("=> L1->L2 read latency %.0f cycles / request" % (c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"]))
    return " | ".join(parts)


lines = ["# " + cmd, "# " + __doc__.split("usage:")[0].strip().replace("\n", "\n# ")]
for k in sorted(fam):
    lines.append("%-9s %s" % (k, line(fam[k])))
lines.append("# per kernel instance (largest GUI-active share first)")
key = lambda kv: -max([v for n, v in kv[1].items() if n.startswith("GUI@")] + [0.0])
for (k, short), c in sorted(per_kernel.items(), key=key)[:12]:
    lines.append("  %-9s %-100s\n      %s" % (k, short, line(c)))
open(out_txt, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
