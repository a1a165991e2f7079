"""Where do the activation gradients of the benchmarked model sit in the fp16 range?  (test infrastructure, CPU only)

fp32 oracle forward + backward of ONE full-size clip (the configuration of test_full_size_clip_matches_oracle: ava_r50_lfb_nl,
32 x 224^2, 2 RoIs, synthetic weights), gradient of every recorded blob retained.  For the loss scale S the engine picks for
that plan (2^13) it prints, per blob: max |g| * S against the fp16 maximum 65504 (head-room in binades) and the share of the
non-zero entries that are fp16 subnormals (< 2^-14) or flush to zero (< 2^-25) after scaling.
Usage: python scratch/r4/grad_headroom.py [log2 S]"""
import sys
sys.path.insert(0, 'video-long-term-feature-banks_amd/lib'); sys.path.insert(0, '.')
import math
import numpy as np, torch
from collections import OrderedDict
from vlfb.presets import load_preset
from core.config import config as cfg
from oracle import model as om
from vlfb import rng as vrng

LOG2S = float(sys.argv[1]) if len(sys.argv) > 1 else 13.0
FR, CROP = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (32, 224)
torch.set_num_threads(8)
load_preset("ava_r50_lfb_nl", ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 1, "TRAIN.VIDEO_LENGTH", FR, "TRAIN.CROP_SIZE", CROP])
inputs = om.synth_inputs(cfg, 1, "train", seed=cfg.RNG_SEED, rois_per_clip=[2], crop=CROP, frames=FR)
params = om.synth_params(cfg, seed=cfg.RNG_SEED)
dtype = torch.float32
P = OrderedDict()
spec = om.param_spec(cfg, False)
for k, v in params.items():
    t = torch.from_numpy(np.asarray(v)).to(dtype)
    if spec.get(k, {}).get("trainable", False):
        t.requires_grad_(True)
    P[k] = t
I = {}
for k, v in inputs.items():
    t = torch.from_numpy(np.asarray(v))
    I[k] = t if t.dtype in (torch.int32, torch.int64) or k == "proposals" else t.to(dtype)
blobs = om.forward(cfg, P, I, "train", False, dtype, lambda name: vrng.dropout_seed(cfg.RNG_SEED, name, 0))
kept = []
for n, b in blobs.items():
    if torch.is_tensor(b) and b.requires_grad and b.dtype == dtype:
        b.retain_grad()
        kept.append(n)
blobs["loss"].backward()
S = 2.0 ** LOG2S
print("loss scale 2^%g; %d blobs with gradients" % (LOG2S, len(kept)))
print("%-34s %11s %9s %8s %8s" % ("blob", "max|g|*S", "headroom", "subnorm", "flushed"))
rows = []
for n in kept:
    g = blobs[n].grad
    if g is None:
        continue
    a = (g.abs() * S).reshape(-1)
    nz = a[a > 0]
    if nz.numel() == 0:
        continue
    mx = float(nz.max())
    rows.append((math.log2(65504.0 / mx), n, mx, float((nz < 2.0 ** -14).float().mean()), float((nz < 2.0 ** -25).float().mean())))
for hr, n, mx, sub, fl in sorted(rows):
    print("%-34s %11.4g %8.1fb %7.2f%% %7.2f%%" % (n, mx, hr, 100 * sub, 100 * fl))
