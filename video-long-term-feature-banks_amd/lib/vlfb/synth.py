"""Synthetic clips / proposals / bank features / weights for bench.py and smoke() (no datasets or
checkpoints are reachable).  Shapes and value ranges follow the reference's blobs (SURVEY.md 8a
I1-I4, 8d): data ~ N(0,1) clipped to the range of (x/255-0.45)/0.225, multi-hot int32 labels,
proposals [clip_idx, x1, y1, x2, y2] inside the crop, non-negative zero-padded bank rows.
Weights: the recorded fillers' distributions, except that the frozen affine layers get gains like
a trained, folded BatchNorm (small on the residual-branch exits) so activations stay O(1)
through the un-normalised residual stack in bf16."""
import math
from collections import OrderedDict

import numpy as np


def rois_per_clip_draw(n_clips, seed=2):
    """RoIs per clip ~ U{1..5} (so R ~ 3 per clip on average)"""
    gen = np.random.default_rng(seed + 7919)
    return [int(gen.integers(1, 6)) for _ in range(n_clips)]


def inputs(cfg, n_clips, rois_per_clip=3, seed=2, crop=None, frames=None, suffix="_train"):
    """rois_per_clip: an int (every clip) or a list with one count per clip"""
    gen = np.random.default_rng(seed)
    per_clip = list(rois_per_clip) if isinstance(rois_per_clip, (list, tuple)) else [int(rois_per_clip)] * n_clips
    assert len(per_clip) == n_clips
    crop = crop or cfg.TRAIN.CROP_SIZE
    frames = frames or cfg.TRAIN.VIDEO_LENGTH
    out = OrderedDict()
    out["data" + suffix] = np.clip(gen.standard_normal((n_clips, 3, frames, crop, crop), dtype=np.float32), -2.0, 2.45)
    ncls = cfg.MODEL.NUM_CLASSES
    if cfg.DATASET == "ava":
        rows = []
        for c in range(n_clips):
            for _ in range(per_clip[c]):
                x1, y1 = gen.uniform(0, crop - 9, 2)
                rows.append([c, x1, y1, gen.uniform(x1 + 8, crop - 1), gen.uniform(y1 + 8, crop - 1)])
        out["proposals" + suffix] = np.asarray(rows, dtype=np.float32)
        R = len(rows)
        out["labels" + suffix] = (gen.uniform(size=(R, ncls)) < 0.05).astype(np.int32)
        if cfg.LFB.ENABLED:
            K = cfg.LFB.WINDOW_SIZE * cfg.AVA.LFB_MAX_NUM_FEAT_PER_STEP
            lfb = np.maximum(gen.standard_normal((R, K, cfg.LFB.LFB_DIM), dtype=np.float32), 0) * 0.5
            lfb *= (gen.uniform(size=(R, K, 1)) < 0.5)          # about half of the slots are zero padding
            out["lfb" + suffix] = lfb.astype(np.float32)
    else:
        out["labels" + suffix] = (gen.uniform(size=(n_clips, ncls)) < 0.05).astype(np.int32)
        if cfg.LFB.ENABLED:
            K = cfg.LFB.WINDOW_SIZE
            out["lfb" + suffix] = (np.maximum(gen.standard_normal((n_clips, K, cfg.LFB.LFB_DIM), dtype=np.float32), 0) * 0.5)
    return out


def params(model, seed=2):
    """{name: array in the reference layout} for every parameter of `model`"""
    gen = np.random.default_rng(seed)
    out = OrderedDict()
    fills = model.param_init_net.fills
    for name in list(model.params) + list(model.computed_params):
        f = fills[name]
        shape = f.shape
        bn = name.rsplit("_", 1)[0] + "_riv" in fills        # scale / bias / running statistics of a SpatialBN
        if bn and name.endswith("_rm"):
            v = gen.standard_normal(shape) * 0.1
        elif bn and name.endswith("_riv"):
            v = gen.uniform(0.5, 1.5, shape)
        elif name in model.affine_params or bn:
            if name.endswith("_s"):
                small = "_branch2c_bn" in name or name.startswith("nonlocal")
                v = gen.uniform(0.15, 0.35, shape) if small else gen.uniform(0.5, 1.5, shape)
            else:
                v = gen.standard_normal(shape) * 0.1
        elif f.fill == "MSRAFill":
            fan_out = shape[0] * int(np.prod(shape[2:]))
            v = gen.standard_normal(shape) * math.sqrt(2.0 / fan_out)
        elif f.fill == "GaussianFill":
            v = gen.standard_normal(shape) * f.kwargs.get("std", 0.01)
        elif len(shape) >= 2:          # zero-initialised output convs: give them small weights
            v = gen.standard_normal(shape) * 0.01
        else:
            v = np.zeros(shape)
        out[name] = np.asarray(v, dtype=np.float32)
    return out
