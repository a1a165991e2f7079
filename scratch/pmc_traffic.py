"""HBM bytes per launch of the GEMM kernel families from two rocprofv3 PMC passes
(--pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs as MI355X_MICROARCH.md prescribes).
usage: pmc_traffic.py <fetch_dir> <write_dir> <out.txt> <out.json> "<command line>"
Units: the counters are KiB; FETCH_SIZE is doubled (gfx950 counts 128-byte requests of wide
coalesced reads as 64 B)."""
import sys, glob, sqlite3, json, collections, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_family(root, counter):
    db = glob.glob(root + '/**/*.db', recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    fam = collections.defaultdict(lambda: [0, 0.0])
    q = "select kernel_name, dispatch_id, sum(value) from counters_collection where counter_name = ? group by dispatch_id"
    for name, _, v in cur.execute(q, (counter,)):
        key = 'gemm_nt' if ('gemm_nt_kernel' in name or 'gemm_nt_sp_kernel' in name or 'gemm_nt8_kernel' in name or 'gemm_nts_kernel' in name or 'gemm_skinny_nt_kernel' in name or 'stem_fprop_kernel' in name or 'conv_rows64_kernel' in name) else 'gemm_tn' if ('gemm_tn_' in name or 'gemm_tn8_' in name or 'stem_wgrad' in name or 'wgrad_rows' in name or 'wgrad_reduce' in name) else None
        if key:
            if 'wgrad_reduce' not in name:
                fam[key][0] += 1
            fam[key][1] += v
    return fam


fetch, write, out_txt, out_json, cmd = sys.argv[1:6]
f, w = per_family(fetch, 'FETCH_SIZE'), per_family(write, 'WRITE_SIZE')
lines = ["# rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- " + cmd,
         "# separate passes as MI355X_MICROARCH.md prescribes; units KiB; FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B for wide coalesced reads)",
         "# gemm_nt = gemm_nt_ / gemm_nt8_ / gemm_nts_ / gemm_skinny_nt_ / stem_fprop_ / conv_rows64_kernel (every vlfb_conv_run FPROP / DGRAD launch); gemm_tn = gemm_tn_* + stem_wgrad + wgrad_rows* launches, including their wgrad_reduce launches (slab reads) in the byte count"]
js = {}
for k in ('gemm_nt', 'gemm_tn'):
    n = max(f[k][0], 1)
    fm, wm = f[k][1] / n / 1024.0, w[k][1] / max(w[k][0], 1) / 1024.0
    tot = 2 * fm + wm
    lines.append("%-8s launches %5d  FETCH_SIZE/launch %8.2f MiB (x2 corrected %8.2f MiB)  WRITE_SIZE/launch %8.2f MiB  -> HBM traffic/launch %8.2f MiB"
                 % (k, n, fm, 2 * fm, wm, tot))
    js[k] = {"launches": n, "bytes_per_launch": tot * 1048576.0}
from bench import kernel_source_hash
js["csrc_sha256"] = kernel_source_hash()      # bench.py reports these bytes only for the kernel sources they were measured on
js["command"] = cmd
import re as _re
_m = _re.search(r"--dtype\s+(\w+)", cmd)
js["dtype"] = _m.group(1) if _m else "fp16"      # bench.py's default dtype
open(out_txt, 'w').write('\n'.join(lines) + '\n')
json.dump(js, open(out_json, 'w'))
print('\n'.join(lines))
