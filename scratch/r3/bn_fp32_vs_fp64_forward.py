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
    b32, _ = om.run(cfg, params, inputs, "train", torch.float32, False, lambda n: 0)
    b64, _ = om.run(cfg, params, inputs, "train", torch.float64, False, lambda n: 0)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    print(tag, ["%s %.1e" % (k, rel(b32[k], b64[k])) for k in ("res_conv1_bn", "res2_2_branch2c_bn", "res3_3_branch2c_bn", "res4_5_branch2c_bn", "res5_2_branch2c_bn", "prob")])
    if tag == "bn":
        x = b64["res4_5_branch2c_bn"]
        # spread of per-channel inverse std of the last BN inputs is not directly available; report min channel std of block outputs
        print("min / median channel std of res4_5 output:", float(x.std((0,2,3,4)).min()), float(x.std((0,2,3,4)).median()))
