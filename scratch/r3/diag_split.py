"""Development diagnostic: per-tensor difference between the split-bf16 engine and the exact-fp32 engine on ONE full-size
clip, in backward-completion order (where does a backward error first appear?).  usage: diag_split.py [preset]"""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "video-long-term-feature-banks_amd", "lib")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_model_gpu import build, FULL, rel

preset = sys.argv[1] if len(sys.argv) > 1 else "ava_r50_lfb_nl"
res = {}
for dtype in ("fp32", "split"):
    cfg, model, eng, inputs, params, seed_fn = build(preset, dtype, FULL)
    eng.forward(); eng.backward(); torch.cuda.synchronize()
    g = collections.OrderedDict((n, eng.fetch_grad(n)) for n in eng.train_order)
    acts = {}
    for st in eng.steps:
        for b in st.outputs:
            if b.root is b and b.kind == "act" and b.slot is not None and b.slot.cur is not None and not getattr(b, "dead", False):
                try:
                    acts[b.name] = eng.fetch(b.name + "_grad")
                except Exception:
                    pass
    res[dtype] = (g, acts)
    order = [b.name for st in reversed(eng.steps) for b in st.outputs if b.name in acts]
    del eng; torch.cuda.empty_cache()
print("== parameter gradients, backward-completion order (split vs fp32 engine)")
for n in res["fp32"][0]:
    print("  %-40s %.3e" % (n, rel(res["split"][0][n], res["fp32"][0][n])))
print("== activation gradients, backward order")
for n in order:
    if n in res["split"][1]:
        print("  %-40s %.3e" % (n, rel(res["split"][1][n], res["fp32"][1][n])))
