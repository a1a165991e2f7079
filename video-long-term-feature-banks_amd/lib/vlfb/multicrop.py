"""AVA multi-crop testing over the device engine.

Reference: tools/test_net.py:48-93 runs the test net once per (flip in {False, True}) x (scale in
AVA.TEST_MULTI_CROP_SCALES) x (spatial shift in {0, 1, 2}) with TEST.SCALE = scale and
TEST.CROP_SIZE = min(256, scale), writes the `pred` logits of every box to a csv per pass, and
lib/utils/metrics.py:599-716 merges them: for one (scale, flip) the sigmoid scores of the shifts whose
crop overlaps the box are averaged (a box may fall outside the left / right crop), then the (scale, flip)
results are summed.  Dataset I/O, csv files and the mAP evaluation are out of scope; this module is the
part that touches the model: device preprocessing (datasets.data_input_helper, one kernel per clip), one
planned Engine per crop size, and the merge arithmetic on arrays.
"""
import collections

import numpy as np

from core.config import config as cfg


def shift_validity(boxes_norm, flip, scale, height, width, max_crop=256):
    """(3, R) bool: which spatial shifts (0 = left/top, 1 = centre, 2 = right/bottom) see each box
    (metrics.py:652-676; the reference's geometry assumes landscape frames resized to height = scale).
    boxes_norm: (R, 4) [x1, y1, x2, y2] normalised to the ORIGINAL frame."""
    b = np.asarray(boxes_norm, dtype=np.float64).copy()
    w = float(width * scale) / height
    norm_crop = float(min(scale, max_crop)) / w
    center_left, center_right = 0.5 - norm_crop / 2.0, 0.5 + norm_crop / 2.0
    lcrop_right, rcrop_left = norm_crop, 1.0 - norm_crop
    if flip:
        b[:, 0], b[:, 2] = 1.0 - b[:, 2].copy(), 1.0 - b[:, 0].copy()
    return np.stack([b[:, 0] < lcrop_right, (b[:, 2] > center_left) & (b[:, 0] < center_right), b[:, 2] > rcrop_left])


def merge_shifts(logits, valid):
    """logits (3, R, C), valid (3, R) -> (R, C): mean over the valid shifts of sigmoid(logit) (metrics.py:677)"""
    s = 1.0 / (1.0 + np.exp(-np.asarray(logits, dtype=np.float64)))
    v = np.asarray(valid, dtype=np.float64)[:, :, None]
    n = v.sum(axis=0)
    with np.errstate(invalid="ignore", divide="ignore"):
        return (s * v).sum(axis=0) / n          # a box no crop sees gives nan, like np.mean([]) in the reference


def merge_scales_and_flips(per_pass):
    """sum over the (scale, flip) results (metrics.py:689-711)"""
    return np.sum(np.stack(per_pass), axis=0)


class AvaMultiCropTester(object):
    """Owns one forward-only Engine per crop size; `run` does the 2 x len(scales) x 3 passes for one batch
    of clips and returns the merged (R, classes) scores."""

    def __init__(self, params, dtype="bf16", device="cuda:0", scales=None, max_crop=256):
        self.params, self.dtype, self.device = params, dtype, device
        self.scales = list(scales if scales is not None else cfg.AVA.TEST_MULTI_CROP_SCALES)
        self.max_crop = int(max_crop)
        self._engines = {}

    def _engine(self, crop, n_clips, frames, n_rois, lfb_shape):
        from models.model_builder_video import ModelBuilder
        from vlfb.engine import Engine
        key = (crop, n_clips, frames, n_rois, lfb_shape)
        if key not in self._engines:
            cfg.TEST.CROP_SIZE = crop
            model = ModelBuilder(train=False, split=cfg.TEST.DATA_TYPE, name="final_test_%d" % crop)
            model.build_model(suffix="_final_test")
            eng = Engine(model, self.dtype, device=self.device)
            shapes = collections.OrderedDict()
            for n in model.input_blob_names:
                if n.startswith("data"):
                    shapes[n] = (n_clips, 3, frames, crop, crop)
                elif n.startswith("proposals"):
                    shapes[n] = (n_rois, 5)
                elif n.startswith("lfb"):
                    shapes[n] = lfb_shape
                elif n.startswith("labels"):
                    shapes[n] = (n_rois, cfg.MODEL.NUM_CLASSES)
            eng.plan(shapes)
            eng.feed_params({k: v for k, v in self.params.items() if k in eng.param_views})
            self._engines[key] = (model, eng)
        return self._engines[key]

    def run(self, clips, boxes, lfb=None):
        """clips: list of (T, H, W, 3) uint8 BGR frame stacks (one per clip); boxes: list of (r_i, 4) arrays
        normalised to the original frames; lfb: (R, K, D) bank rows per RoI (models with LFB.ENABLED).
        Returns (scores (R, classes) float64, per_pass {(scale, flip, shift): logits (R, classes)})."""
        import torch
        from datasets import data_input_helper as dih
        n_clips = len(clips)
        T, H, W = clips[0].shape[:3]
        n_rois = int(sum(len(b) for b in boxes))
        all_boxes = np.concatenate([np.asarray(b, dtype=np.float64).reshape(-1, 4) for b in boxes])
        per_pass, merged = {}, []
        saved = (cfg.TEST.SCALE, cfg.TEST.CROP_SIZE, cfg.AVA.FORCE_TEST_FLIP)
        try:
            for scale in self.scales:                        # metrics.combine_ava_multi_crops order: scale, then flip
                crop = min(self.max_crop, scale)
                model, eng = self._engine(crop, n_clips, T, n_rois, None if lfb is None else tuple(lfb.shape))
                names = {n.split("_")[0]: n for n in model.input_blob_names}
                if lfb is not None and "lfb" in names:
                    eng.feed(names["lfb"], lfb)
                data, (wpad, cpad) = eng.blob_padded(names["data"])
                for flip in (False, True):
                    cfg.TEST.SCALE, cfg.TEST.CROP_SIZE, cfg.AVA.FORCE_TEST_FLIP = scale, crop, flip
                    logits = []
                    for shift in range(3):
                        rows = []
                        for c in range(n_clips):
                            _, b = dih.images_and_boxes_preprocessing(clips[c], 0, crop, shift, boxes=boxes[c], out=data[c],
                                                                      out_dtype=data.dtype, w_pad=wpad, c_pad=cpad,
                                                                      device=self.device)
                            rows.append(np.concatenate([np.full((len(b), 1), c, dtype=np.float64), b], axis=1))
                        eng.feed(names["proposals"], np.concatenate(rows).astype(np.float32))
                        eng.forward()
                        torch.cuda.synchronize()
                        lg = eng.fetch("pred").reshape(n_rois, -1).astype(np.float64)
                        per_pass[(scale, flip, shift)] = lg
                        logits.append(lg)
                    merged.append(merge_shifts(np.stack(logits), shift_validity(all_boxes, flip, scale, H, W, self.max_crop)))
        finally:
            cfg.TEST.SCALE, cfg.TEST.CROP_SIZE, cfg.AVA.FORCE_TEST_FLIP = saved
        return merge_scales_and_flips(merged), per_pass
