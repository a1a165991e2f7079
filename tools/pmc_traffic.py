"""HBM bytes per launch of the GEMM kernel families from two rocprofv3 PMC passes
(--pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs as MI355X_MICROARCH.md prescribes) of bench.py's command line.
usage: tools/pmc_traffic.py <fetch_dir> <write_dir> <out.txt> <out.json> "<command line>"
Units: the counters are KiB; FETCH_SIZE is doubled (gfx950 counts 128-byte requests of wide coalesced reads as 64 B).
Families are bench.py's (hip.conv_family): nt_pair = the two-plane fp16 forward kernels (csrc/vlfb_gemm_pair.hip), nt_split / tn_split = the split-bf16 kernels of csrc/vlfb_gemm_split.hip (and
gemm_tn_tr_kernel<..., SP>), nt_16 / tn_16 = the 16-bit families, nt_f32 / tn_f32 the exact-fp32 kernels; the bytes of a
split-K launch include its wgrad_reduce / wgrad_bias_reduce launches (slab reads)."""
import sys, glob, sqlite3, json, collections, os, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def family(name):
    # the two-plane fp16 instances (vlfb_gemm_pair.hip): PAIR is the 13th template argument of gemm_nt_kernel (W2I follows it),
    # the 7th of gemm_nt8_kernel
    m = re.search(r"gemm_(nt8?)_kernel<([^>]*)>\(", name)
    if m:
        targs = [a.strip() for a in m.group(2).split(",")]
        at = 12 if m.group(1) == "nt" else 6
        if len(targs) > at and targs[at] == "true":
            return "nt_pair"
    if "stem_fprop_pair_kernel" in name:          # conv1 of the two-plane forward (vlfb_stem.hip)
        return "nt_pair"
    if "gemm_nt_sp_kernel" in name or "gemm_nt_pl_kernel" in name or "gemm_skinny_nt_sp_kernel" in name:
        return "nt_split"
    if "gemm_tn_sp_kernel" in name or re.search(r"gemm_tn_tr_kernel<.*, true>\(", name):
        return "tn_split"
    f32 = re.search(r"gemm_(nt|tn)_kernel<float", name)
    if f32:
        return f32.group(1) + "_f32"
    if any(k in name for k in ("gemm_nt_kernel", "gemm_nt8_kernel", "gemm_nts_kernel", "gemm_skinny_nt_kernel", "stem_fprop_kernel", "conv_rows64_kernel")):
        return "nt_16"
    if any(k in name for k in ("gemm_tn_", "gemm_tn8_", "stem_wgrad", "wgrad_rows")):
        return "tn_16"
    return None


def per_family(root, counter):
    db = glob.glob(root + "/**/*.db", recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    fam = collections.defaultdict(lambda: [0, 0.0])
    reduce_bytes, last_tn = 0.0, "tn_16"
    q = "select kernel_name, dispatch_id, sum(value) from counters_collection where counter_name = ? group by dispatch_id order by dispatch_id"
    for name, _, v in cur.execute(q, (counter,)):
        if "wgrad_reduce" in name or "wgrad_bias_reduce" in name:
            fam[last_tn][1] += v             # (the reduce of the split-K launch in front of it: bytes, no launch count)
            continue
        key = family(name)
        if key:
            fam[key][0] += 1
            fam[key][1] += v
            if key.startswith("tn"):
                last_tn = key
    return fam


if __name__ == "__main__":
    fetch, write, out_txt, out_json, cmd = sys.argv[1:6]
    f, w = per_family(fetch, "FETCH_SIZE"), per_family(write, "WRITE_SIZE")
    lines = ["# rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- " + cmd,
             "# separate passes as MI355X_MICROARCH.md prescribes; units KiB; FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B for wide coalesced reads)",
             "# families as in bench.py's roofline_families; a TN family's bytes include the wgrad_reduce / wgrad_bias_reduce launches behind its split-K launches"]
    js = {"families": {}}
    for k in sorted(set(f) | set(w)):
        n = max(f[k][0], 1)
        fm, wm = f[k][1] / n / 1024.0, w[k][1] / max(w[k][0], 1) / 1024.0
        tot = 2 * fm + wm
        lines.append("%-9s launches %5d  FETCH_SIZE/launch %8.2f MiB (x2 corrected %8.2f MiB)  WRITE_SIZE/launch %8.2f MiB  -> HBM traffic/launch %8.2f MiB"
                     % (k, n, fm, 2 * fm, wm, tot))
        js["families"][k] = {"launches": n, "bytes_per_launch": tot * 1048576.0}
    from bench import kernel_source_hash
    js["csrc_sha256"] = kernel_source_hash()      # bench.py reports these bytes only for the kernel sources they were measured on
    js["command"] = cmd
    m = re.search(r"--dtype\s+(\w+)", cmd)
    js["dtype"] = m.group(1) if m else "mix"      # bench.py's default dtype
    open(out_txt, "w").write("\n".join(lines) + "\n")
    json.dump(js, open(out_json, "w"))
    print("\n".join(lines))
