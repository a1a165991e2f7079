"""Round 5, first GPU call: the "mix" dtype's default-off parity switches (Engine.MIX_HEAD_F32, Engine.MIX_W2_SKIP) against
the fp64 oracle at the benchmarked clip size, all variants in ONE process per preset (the split forward and therefore the
discrete decisions are the same for every variant, so the oracle runs twice per preset, not twice per variant).
Usage: [MIX_SMALL=1] python scratch/r5/mix_variants.py preset ..."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "video-long-term-feature-banks_amd", "lib")); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_model_gpu import build, rel, CHECK_BLOBS, SMALL
from vlfb.engine import Engine
from oracle import model as om

def same_dec(a, b):
    for k in ("relu", "pool"):
        if set(a[k]) != set(b[k]) or any(not np.array_equal(a[k][n], b[k][n]) for n in a[k]):
            return False
    return (a["roi_bin"] is None and b["roi_bin"] is None) or np.array_equal(a["roi_bin"], b["roi_bin"])


presets = sys.argv[1:] or ["ava_r50_lfb_nl", "charades_r50_baseline"]
FULL = ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 1, "TRAIN.VIDEO_LENGTH", 32, "TRAIN.CROP_SIZE", 224]
size = SMALL if os.environ.get("MIX_SMALL") else FULL
VARIANTS = [(True, ())]     # (round 5: the switches are gone -- the fp32 head is part of `mix`, W2_SKIP was deleted)
for preset in presets:
    ref = None; dec0 = None; g2 = None
    for head, skip in VARIANTS:
        cfg, model, eng, inputs, params, seed_fn = build(preset, "mix", size)
        eng.forward(); eng.backward(); torch.cuda.synchronize()
        if ref is None:
            torch.set_num_threads(min(32, os.cpu_count()))
            ref = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn)
        blobs, grads = ref
        acts = [rel(eng.fetch(n), blobs[n].detach().numpy().reshape(eng.fetch(n).shape)) for n in CHECK_BLOBS if n in blobs]
        gmax = max(float(g.norm()) for g in grads.values())
        names = [n for n in eng.trainable if np.linalg.norm(grads[n].numpy()) >= 1e-9 * gmax]
        raw = np.array([rel(eng.fetch_grad(n), grads[n].numpy()) for n in names])
        dec = eng.discrete_decisions()
        same = dec0 is not None and same_dec(dec, dec0)
        if not same:
            if dec0 is not None:
                print("   (decisions differ from the first variant: oracle re-run)", flush=True)
            _, g2 = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn, decisions=dec)
            if dec0 is None:
                dec0 = dec
        cond = sorted(((rel(eng.fetch_grad(n), g2[n].numpy()), n) for n in names), reverse=True)
        e = np.array([x for x, _ in cond])
        print("[%s head_f32=%s w2_skip=%s] act max %.2e | raw median %.2e max %.2e | identical decisions: median %.2e p90 %.2e max %.2e (%s) 2nd %.2e (%s) 3rd %.2e (%s) | loss_scale %g"
              % (preset, head, ",".join(skip) or "-", max(acts), np.median(raw), raw.max(), np.median(e), np.sort(e)[int(0.9 * (len(e) - 1))],
                 e[0], cond[0][1], e[1], cond[1][1], e[2], cond[2][1], eng.loss_scale), flush=True)
        for x, n in cond[:12]:
            print("   %-44s %.3e" % (n, x))
        del eng
        torch.cuda.empty_cache()
