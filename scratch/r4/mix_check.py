"""GPU check of the "mix" dtype (split-bf16 forward + fp16 backward) against the fp64 oracle: raw and on identical
ReLU / max-pool / RoI-bin decisions.  Usage: python scratch/r4/mix_check.py [preset ...]"""
import sys, os, collections, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "video-long-term-feature-banks_amd", "lib")); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_model_gpu import build, rel, CHECK_BLOBS, SMALL
from vlfb.engine import Engine
from oracle import model as om

presets = sys.argv[1:] or ["ava_r50_lfb_nl", "charades_r50_baseline"]
FULL = ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 1, "TRAIN.VIDEO_LENGTH", 32, "TRAIN.CROP_SIZE", 224]
size = FULL if os.environ.get("MIX_FULL") else SMALL
variants = [("mix", True, True, True), ("mix", True, True, False)]
if os.environ.get("MIX_FULL"):
    variants = [("mix", True, True, True)]
for preset in presets:
    ref = None
    for dtype, w2, nl, t2 in variants:
        if w2 is not None:
            Engine.MIX_W2, Engine.MIX_NL_F32, Engine.MIX_TRUNK2 = w2, nl, t2
        cfg, model, eng, inputs, params, seed_fn = build(preset, dtype, size)
        eng.forward(); eng.backward(); torch.cuda.synchronize()
        if ref is None:
            torch.set_num_threads(min(32, os.cpu_count()))
            blobs, grads = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn)
            ref = (blobs, grads)
        blobs, grads = ref
        acts = []
        for name in CHECK_BLOBS:
            if name in blobs:
                got = eng.fetch(name)
                acts.append(rel(got, blobs[name].detach().numpy().reshape(got.shape)))
        gmax = max(float(g.norm()) for g in grads.values())
        names = [n for n in eng.trainable if np.linalg.norm(grads[n].numpy()) >= 1e-9 * gmax]
        raw = sorted(((rel(eng.fetch_grad(n), grads[n].numpy()), n) for n in names), reverse=True)
        dec = eng.discrete_decisions()
        _, g2 = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn, decisions=dec)
        cond = sorted(((rel(eng.fetch_grad(n), g2[n].numpy()), n) for n in names), reverse=True)
        e = np.array([x for x, _ in cond]); r = np.array([x for x, _ in raw])
        print("[%s %s w2=%s nl_f32=%s trunk2=%s] act max %.2e | raw median %.2e max %.2e | identical decisions: median %.2e p90 %.2e max %.2e (%s) 2nd %.2e (%s) | loss_scale %g"
              % (preset, dtype, w2, nl, t2, max(acts), np.median(r), r.max(), np.median(e), np.sort(e)[int(0.9 * (len(e) - 1))], e[0], cond[0][1], e[1], cond[1][1], eng.loss_scale), flush=True)
        if os.environ.get("MIX_FULL"):
            for x, n in cond:
                print("   %-44s %.3e" % (n, x))
        eng.sgd_step(0.01); torch.cuda.synchronize()
        del eng
