"""MFMA-pipe utilisation, LDS bank conflicts and wave stall shares per kernel family from ONE rocprofv3 PMC pass of bench.py:
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \\
            SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU -d <dir> -o q -- python bench.py ...
usage: tools/pmc_sq.py <dir> <out.txt> "<command line>"
Units (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD, summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is
summed over the 8 XCDs; SQ_WAVE_CYCLES / SQ_WAIT_* are quad-cycles per wave.  MFMA pipe busy = MFMA_BUSY / (1024 x GUI_ACTIVE / 8);
parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES (s_waitcnt / barrier), issue-stalled = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES."""
import sys, glob, sqlite3, collections, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import family

root, out_txt, cmd = sys.argv[1:4]
db = glob.glob(root + "/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
fam = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
q = "select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection group by dispatch_id, counter_name"
for name, cname, did, v in cur.execute(q):
    k = family(name)
    if k is None:
        continue
    fam[k][cname] += v
    disp[k].add(did)
    short = name.replace("(anonymous namespace)::", "").replace("vlfb::", "").replace("void ", "").split("(")[0][:100]
    per_kernel[(k, short)][cname] += v
    per_kernel[(k, short)]["_n"] += (cname == "GRBM_GUI_ACTIVE")


def line(c):
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    lds_a = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
    return ("MFMA pipe busy %5.1f %% | LDS bank-conflict cycles %6.3f %% of LDS-active | waves parked %4.1f %% issue-stalled %4.1f %% | "
            "VALU (incl. MFMA) per MFMA %.2f" % (
                100.0 * mf / (128.0 * gui) if gui else 0.0, 100.0 * c.get("SQ_LDS_BANK_CONFLICT", 0.0) / lds_a if lds_a else 0.0,
                100.0 * c.get("SQ_WAIT_ANY", 0.0) / wc if wc else 0.0, 100.0 * c.get("SQ_WAIT_INST_ANY", 0.0) / wc if wc else 0.0,
                c.get("SQ_INSTS_VALU", 0.0) / c["SQ_INSTS_MFMA"] if c.get("SQ_INSTS_MFMA") else 0.0))


lines = ["# " + cmd, "# " + __doc__.split("usage:")[1].split("\n", 1)[1].strip().replace("\n", "\n# ")]
for k in sorted(fam):
    lines.append("%-9s (%5d launches)  %s" % (k, len(disp[k]), line(fam[k])))
lines.append("# per kernel (instances with the largest share of GPU-active time first)")
tot = {k: fam[k].get("GRBM_GUI_ACTIVE", 1.0) for k in fam}
for (k, short), c in sorted(per_kernel.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0))[:14]:
    lines.append("  %-9s %5.1f %% of family time  %-86s %s" % (k, 100.0 * c.get("GRBM_GUI_ACTIVE", 0.0) / tot[k], short, line(c)))
open(out_txt, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
