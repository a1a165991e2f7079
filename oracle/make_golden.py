"""Generates tests/golden/<preset>.npz: a compact fingerprint of the fp64 oracle on the small
synthetic problem the GPU parity tests use (2 clips, 16 frames, 64x64 crop, seed 2).

    python -m oracle.make_golden            # rewrites the fixtures (only when the oracle changes)

The reference itself has no golden vectors and cannot run (see oracle/__init__.py), so these
vectors pin the ORACLE (regressions, torch-version drift), not the reference.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 2, "TRAIN.VIDEO_LENGTH", 16, "TRAIN.CROP_SIZE", 64]
PRESETS = ["charades_r50_baseline", "ava_r50_lfb_nl"]


def compute(preset):
    sys.path.insert(0, os.path.join(ROOT, "video-long-term-feature-banks_amd", "lib"))
    from vlfb.presets import load_preset
    from vlfb import rng as vrng
    from core.config import config as cfg
    from oracle import model as om
    load_preset(preset, SMALL)
    inputs = om.synth_inputs(cfg, 2, "train", seed=2, rois_per_clip=[2, 3] if cfg.DATASET == "ava" else None,
                             crop=64, frames=16)
    params = om.synth_params(cfg, seed=2)
    blobs, grads = om.run(cfg, params, inputs, "train", torch.float64, True,
                          lambda name: vrng.dropout_seed(2, name, 0), threads=4)
    out = {"loss": np.asarray(float(blobs["loss"].detach())), "prob": blobs["prob"].detach().numpy(),
           "pred": blobs["pred"].detach().numpy(), "pool5": blobs["pool5"].detach().numpy().reshape(blobs["pool5"].shape[0], -1),
           "res5_mean_abs": np.asarray(float(blobs["res5_2_branch2c_bn"].abs().mean()))}
    for name in ["pred_w", "pred_b", "conv1_w", "res3_0_branch2b_w", "nonlocal_conv4_1_theta_w", "res5_2_branch2c_w"]:
        flat = grads[name].numpy().reshape(-1)
        out["grad/" + name] = flat[::max(1, flat.size // 4096)][:4096].copy()   # strided sample keeps fixtures small
    out["grad_norms"] = np.asarray([float(g.norm()) for g in grads.values()])
    return out


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    for preset in PRESETS:
        data = compute(preset)
        path = os.path.join(ROOT, "tests", "golden", preset + ".npz")
        np.savez_compressed(path, **data)
        print("wrote", path, {k: v.shape for k, v in data.items()})
