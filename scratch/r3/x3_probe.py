"""forward precision of the split path: bf16x6 (default) against bf16x3 forward products -- activations, raw and
decision-conditioned gradient errors on one full-size AVA clip"""
import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import conftest  # noqa
from test_model_gpu import build, rel, FULL, CHECK_BLOBS
from oracle import model as om
from vlfb.engine import Engine
torch.set_num_threads(32)
ref = None
for math in ((6, 3), (3, 3)):
    Engine.SPLIT_MATH = math
    cfg, model, eng, inputs, params, seed_fn = build("ava_r50_lfb_nl", "split", FULL)
    eng.forward(); eng.backward(); torch.cuda.synchronize()
    if ref is None:
        ref = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn)
    blobs, grads = ref
    acts = [(n, rel(eng.fetch(n), blobs[n].detach().numpy().reshape(eng.fetch(n).shape))) for n in CHECK_BLOBS if n in blobs]
    gmax = max(float(g.norm()) for g in grads.values())
    names = [n for n in eng.trainable if float(grads[n].norm()) > 1e-9 * gmax]
    got = {n: eng.fetch_grad(n) for n in names}
    e = np.sort([rel(got[n], grads[n].numpy()) for n in names])
    dec = eng.discrete_decisions()
    nflip = sum(int(((blobs[n].detach().numpy() > 0) != m).sum()) for n, m in dec["relu"].items() if n in blobs)
    _, g2 = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn, decisions=dec)
    ce = np.sort([rel(got[n], g2[n].numpy()) for n in names])
    print("SPLIT_MATH", math, "acts max %.2e" % max(x for _, x in acts), "raw med %.2e max %.2e" % (np.median(e), e[-1]),
          "flips", nflip, "conditioned med %.2e p90 %.2e max %.2e" % (np.median(ce), ce[int(.9 * (len(ce) - 1))], ce[-1]), flush=True)
    del eng; torch.cuda.empty_cache()
