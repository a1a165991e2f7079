import sys, numpy as np, torch
sys.path.insert(0, "/root/repo/video-long-term-feature-banks_amd/lib"); sys.path.insert(0, "/root/repo")
from vlfb.presets import load_preset
from core.config import config as cfg
from oracle import model as om
torch.set_num_threads(16)
BN = ["MODEL.USE_AFFINE", False, "NONLOCAL.USE_BN", True, "NONLOCAL.USE_AFFINE", False]
for tag, ov in (("affine", []), ("bn", BN)):
    load_preset("ava_r50_lfb_nl", ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 2, "TRAIN.VIDEO_LENGTH", 16, "TRAIN.CROP_SIZE", 64] + ov)
    params = om.synth_params(cfg, seed=cfg.RNG_SEED)
    inputs = om.synth_inputs(cfg, 2, "train", seed=cfg.RNG_SEED, rois_per_clip=[2, 3], crop=64, frames=16)
    b32, g32 = om.run(cfg, params, inputs, "train", torch.float32, True, lambda n: 0)
    b64, g64 = om.run(cfg, params, inputs, "train", torch.float64, True, lambda n: 0)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    e = sorted((rel(g32[k], g64[k]), k) for k in g64 if float(g64[k].norm()) > 0)
    print(tag, "raw fp32-oracle vs fp64-oracle: median %.2e max %.2e" % (np.median([x for x, _ in e]), e[-1][0]), e[-3:])
    # condition the fp64 run on the fp32 run's ReLU masks
    dec = {"relu": {}, "pool": {}}
    probe = {"relu": {}, "pool": {}}
    om.run(cfg, params, inputs, "train", torch.float64, False, lambda n: 0, decisions=probe)
    for name in probe["_missing"]:
        if name in b32 and not name.startswith("pool") and "_pool" not in name:
            dec["relu"][name] = (b32[name].detach().numpy() > 0)
    _, gc = om.run(cfg, params, inputs, "train", torch.float64, True, lambda n: 0, decisions=dec)
    e = sorted((rel(g32[k], gc[k]), k) for k in gc if float(gc[k].norm()) > 0)
    print(tag, "relu-conditioned: median %.2e max %.2e" % (np.median([x for x, _ in e]), e[-1][0]), e[-3:], "missing", sorted(dec["_missing"])[:6])
