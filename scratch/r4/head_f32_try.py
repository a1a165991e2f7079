"""one tiny GPU run of Engine.MIX_HEAD_F32: does the path execute, and how far do its gradients move from the default mix path?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "video-long-term-feature-banks_amd", "lib")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_model_gpu import build, rel
from vlfb.engine import Engine
SMALL8 = ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 2, "TRAIN.VIDEO_LENGTH", 8, "TRAIN.CROP_SIZE", 64]
res = {}
for on in (False, True):
    Engine.MIX_HEAD_F32 = on
    cfg, model, eng, inputs, params, seed_fn = build("ava_r50_lfb_nl", "mix", SMALL8)
    eng.forward(); eng.backward(); torch.cuda.synchronize()
    res[on] = {n: eng.fetch_grad(n) for n in eng.trainable}
    print("head_f32=%s marked=%s loss=%.6f" % (on, eng.head_f32, float(eng.fetch("loss").reshape(-1)[0])), flush=True)
    del eng
d = sorted(((rel(res[True][n], res[False][n]), n) for n in res[False] if np.linalg.norm(res[False][n]) > 0), reverse=True)
print("finite:", all(np.isfinite(g).all() for g in res[True].values()), "| on vs off: median %.2e max %.2e (%s)" % (np.median([x for x, _ in d]), d[0][0], d[0][1]))
