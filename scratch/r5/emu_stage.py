"""CPU (fp64 oracle) sensitivity experiment: round the stored activation gradients of ONE stage's convs to fp16 precision
(11 significant bits) and see what that does to every parameter gradient -- which stage's backward roundings are the
ones that survive in conv1_w?  Full-size clip by default.  Usage: python scratch/r5/emu_stage.py [preset]"""
import sys, os, collections, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "video-long-term-feature-banks_amd", "lib")); sys.path.insert(0, ROOT)
import numpy as np, torch
from vlfb.presets import load_preset
from core.config import config as cfg
from vlfb import rng as vrng
from oracle import model as om

preset = sys.argv[1] if len(sys.argv) > 1 else "ava_r50_lfb_nl"
FR, CROP = int(os.environ.get("EMU_FRAMES", 32)), int(os.environ.get("EMU_CROP", 224))
load_preset(preset, ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 1, "TRAIN.VIDEO_LENGTH", FR, "TRAIN.CROP_SIZE", CROP])
inputs = om.synth_inputs(cfg, 1, "train", seed=cfg.RNG_SEED, rois_per_clip=[2] if cfg.DATASET == "ava" else None, crop=CROP, frames=FR)
params = om.synth_params(cfg, seed=cfg.RNG_SEED)
seed_fn = lambda name: vrng.dropout_seed(cfg.RNG_SEED, name, 0)
spec = om.param_spec(cfg)


def rq(t, bits=11):
    m, e = torch.frexp(t)
    s = float(1 << bits)
    return torch.ldexp(torch.round(m * s) / s, e)


SELECT = [lambda k: False]
NHOOK = [0]
_orig_conv = om._conv


def _conv_hooked(x, P, name, *a, **kw):
    """every conv output's gradient is what the engine stores in fp16 (the masked gradient of a branch-internal blob) or
    feeds to a DGRAD as its 11-bit operand (the hi term of the two-term trunk gradient at branch2c / branch1 / theta ...)"""
    y = _orig_conv(x, P, name, *a, **kw)
    if y.requires_grad and SELECT[0](name) and not name.endswith(("_theta", "_phi", "_g")):   # (MIX_NL_F32: those stay fp32)
        y.register_hook(rq); NHOOK[0] += 1
    return y


om._conv = _conv_hooked


def run(select):
    SELECT[0] = select; NHOOK[0] = 0
    P = collections.OrderedDict()
    for k, v in params.items():
        t = torch.from_numpy(v).double()
        if spec[k]["trainable"]:
            t.requires_grad_(True)
        P[k] = t
    I = {k: (torch.from_numpy(v) if v.dtype != np.float32 or k == "proposals" else torch.from_numpy(v).double()) for k, v in inputs.items()}
    B = om.forward(cfg, P, I, "train", False, torch.float64, seed_fn)
    for k, t in B.items():
        if isinstance(t, torch.Tensor) and t.requires_grad and select("blob:" + k):
            t.register_hook(rq); NHOOK[0] += 1
    B["loss"].backward()
    return {k: t.grad.numpy().copy() for k, t in P.items() if t.requires_grad and t.grad is not None}, NHOOK[0]


def rel(a, b):
    d = np.linalg.norm(b.ravel())
    return np.linalg.norm((a - b).ravel()) / (d if d > 0 else 1.0)


t0 = time.time()
ref, _ = run(lambda k: False)
print("exact backward: %.1f s" % (time.time() - t0), flush=True)
gmax = max(np.linalg.norm(g) for g in ref.values())
names = [n for n in ref if np.linalg.norm(ref[n]) >= 1e-9 * gmax]
CASES = [("res5 convs", lambda k: k.startswith("res5_")),
         ("res4 + nl4 convs", lambda k: k.startswith("res4_") or k.startswith("nonlocal_conv4")),
         ("res3 + nl3 convs", lambda k: k.startswith("res3_") or k.startswith("nonlocal_conv3")),
         ("res2 convs", lambda k: k.startswith("res2_")),
         ("pool2", lambda k: k == "blob:pool2"),
         ("pool1 + res_conv1_bn", lambda k: k in ("blob:pool1", "conv1")),
         ("head convs (fbo, lfb)", lambda k: k.startswith("lfb") or "fbonl" in k),
         ("all of the above", lambda k: k in ("blob:pool1", "blob:pool2", "conv1") or not k.startswith("blob:"))]
only = os.environ.get("EMU_ONLY")
for title, sel in CASES:
    if only and only not in title:
        continue
    g, n = run(sel)
    e = sorted(((rel(g[k], ref[k]), k) for k in names), reverse=True)
    v = np.array([x for x, _ in e])
    print("[%s] %d blobs rounded: conv1_w %.2e | median %.2e p90 %.2e max %.2e (%s) 2nd %.2e (%s)"
          % (title, n, rel(g["conv1_w"], ref["conv1_w"]), np.median(v), np.sort(v)[int(0.9 * (len(v) - 1))], e[0][0], e[0][1], e[1][0], e[1][1]), flush=True)
