"""ORACLE (test infrastructure only): the AVA multi-crop merge of the reference, restated on arrays.
  merge_ava_3shift_score_files   lib/utils/metrics.py:623-686   (per box: which crops overlap it; mean of sigmoids)
  merge_ava_score_files          lib/utils/metrics.py:689-711   (sum over scales and flips)
and the per-pass loop of tools/test_net.py:48-93 over oracle.preprocess + oracle.model.  Pinned: the two merge functions of the
reference are run on synthetic score files by oracle/make_ref_aux_golden.py and merge_three_shifts must reproduce their
output exactly (tests/test_ref_aux.py); loops are written per box exactly like the csv code, not vectorised."""
import numpy as np


def sigmoid(x):
    return float(1.0 / (1.0 + np.exp(-x)))


def merge_three_shifts(scores3, boxes_norm, flip, scale, height, width, max_crop=256):
    """scores3: [shift][box][class] logits; boxes_norm: [box] = (x1, y1, x2, y2) of the ORIGINAL frame"""
    out = np.zeros((len(boxes_norm), len(scores3[0][0])))
    for i, box in enumerate(boxes_norm):
        box = [float(v) for v in box]
        h, w = scale, float(width * scale) / height
        norm_crop_size = float(min(h, max_crop)) / w
        center_left = 0.5 - norm_crop_size / 2.0
        center_right = 0.5 + norm_crop_size / 2.0
        lcrop_right = norm_crop_size
        rcrop_left = 1.0 - norm_crop_size
        if flip:
            box[0], box[2] = 1.0 - box[2], 1.0 - box[0]
        for k in range(out.shape[1]):
            valid = []
            if box[2] > center_left and box[0] < center_right:
                valid.append(scores3[1][i][k])
            if box[0] < lcrop_right:
                valid.append(scores3[0][i][k])
            if box[2] > rcrop_left:
                valid.append(scores3[2][i][k])
            out[i, k] = float(np.mean([sigmoid(s) for s in valid])) if valid else float("nan")
    return out


def multi_crop_scores(cfg, params, clips, boxes, lfb, scales, max_crop=256):
    """the 2 x len(scales) x 3 passes through the oracle model; returns (merged (R, classes), per_pass logits)"""
    import torch
    from . import model as om
    from . import preprocess as op
    H, W = clips[0][0].shape[:2]
    all_boxes = np.concatenate([np.asarray(b, dtype=np.float64).reshape(-1, 4) for b in boxes])
    files, per_pass = [], {}
    saved = (cfg.TEST.SCALE, cfg.TEST.CROP_SIZE, cfg.AVA.FORCE_TEST_FLIP)
    try:
        for scale in scales:
            for flip in (False, True):
                cfg.TEST.SCALE, cfg.TEST.CROP_SIZE, cfg.AVA.FORCE_TEST_FLIP = scale, min(max_crop, scale), flip
                shifts = []
                for shift in range(3):
                    data, rows = [], []
                    for c, frames in enumerate(clips):
                        clip, b = op.images_and_boxes_preprocessing(list(frames), 0, cfg.TEST.CROP_SIZE, shift, cfg, boxes=np.asarray(boxes[c]))
                        data.append(clip)
                        rows.append(np.concatenate([np.full((len(b), 1), c, dtype=np.float64), b], axis=1))
                    inputs = {"data": np.stack(data), "proposals": np.concatenate(rows).astype(np.float32)}
                    if lfb is not None:
                        inputs["lfb"] = lfb
                    blobs, _ = om.run(cfg, params, inputs, "test", torch.float64, False, None)
                    lg = blobs["pred"].numpy().reshape(len(all_boxes), -1)
                    per_pass[(scale, flip, shift)] = lg
                    shifts.append(lg)
                files.append(merge_three_shifts(shifts, all_boxes, flip, scale, H, W, max_crop))
    finally:
        cfg.TEST.SCALE, cfg.TEST.CROP_SIZE, cfg.AVA.FORCE_TEST_FLIP = saved
    return np.sum(np.stack(files), axis=0), per_pass
