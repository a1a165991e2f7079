"""One-screen summary of a tools/final_runs.sh output directory: clips/s of every bench line, the per-family roofline records of
the driver line, the SQ-counter families, the step timeline header and the PMC traffic.  usage: tools/batch_summary.py <dir>"""
import glob, json, os, sys
d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, "bench_*.json"))):
    try:
        j = json.load(open(f))
    except Exception as e:
        print("%-34s unreadable (%s)" % (os.path.basename(f), e))
        continue
    fam = j.get("roofline_families", {})
    print("%-34s %-5s %8.1f clips/s %7.2f ms  MFU %.3f / MFMA-FU %.3f  | %s" % (
        os.path.basename(f), j["dtype"], j["value"], j["ms_per_step"], j.get("model_flops_utilisation", 0), j.get("mfma_flops_utilisation", 0),
        "  ".join("%s %.1f ms %.0f TF %.3f" % (k, v["ms_per_step"], v["achieved"], v["frac"]) for k, v in fam.items())))
    if "driver" in f:
        for k in ("fp16_path", "bf16_path", "split_path", "fp32_path"):
            if k in j and j[k].get("value"):
                print("    %-10s %8.1f clips/s (%d steps)  NT frac %.3f" % (k, j[k]["value"], j[k]["steps"], j[k]["roofline"]["frac"]))
        r = j["roofline"]
        print("    roofline: %s" % {k: r[k] for k in ("achieved", "peak", "frac", "traffic", "traffic_ratio", "launches_per_step", "attainable_ms_per_step", "ms_per_step")})
        print("    parity: %s" % json.dumps(j.get("parity"))[:400])
        print("    cpu_baseline: %s" % json.dumps(j.get("cpu_baseline"))[:400])
        print("    host enqueue %.2f ms (%s)" % (j["host_enqueue_ms_per_step"], j["host_enqueue_path"]))
for name in ("timeline_step_mix.txt", "timeline_step_fp16.txt"):
    p = os.path.join(d, name)
    if os.path.exists(p):
        print(name + ": " + " | ".join(open(p).read().split("\n")[:4]))
for name in ("pmc_sq_mix.txt", "pmc_sq_fp16.txt"):
    p = os.path.join(d, name)
    if os.path.exists(p):
        print(name)
        for l in open(p).read().split("\n"):
            if l and not l.startswith("#") and not l.startswith("  "):
                print("   " + l[:200])
p = os.path.join(d, "pmc_hbm_traffic.txt")
if os.path.exists(p):
    print(open(p).read())
for f in sorted(glob.glob(os.path.join(d, "parity_fullsize_*.txt"))):
    print(os.path.basename(f))
    for l in open(f).read().split("\n"):
        if l.startswith("==") or "parameter gradients" in l:
            print("   " + l[:230])
