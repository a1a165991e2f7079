"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference's clip preprocessing:
  images_and_boxes_preprocessing     lib/datasets/data_input_helper.py:70-139
  scale / scale_boxes / random_short_side_scale_jitter_list / random_crop_list /
  horizontal_flip_list / spatial_shift_crop_list / clip_boxes_to_image / color_normalization
                                     lib/datasets/image_processor.py:41-251
The bilinear resize itself is third-party: `cv2.resize(..., interpolation=cv2.INTER_LINEAR)`
(cfg.INTERPOLATION, lib/core/config.py:238) -- OpenCV is not installed here and the reference pins no
version (INSTALL.md: "pip install opencv-python", early 2019 => 3.4.x / 4.0.x).  Restated from the
published algorithm of modules/imgproc/src/resize.cpp for 8-bit INTER_LINEAR: per-axis source index
sx = floor((dx + 0.5) * scale - 0.5) with border clamps, weights quantised to 11 bits
(INTER_RESIZE_COEF_BITS) with cvRound + saturate_cast<short>, horizontal pass in int, vertical pass
((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.  The resize is parity-unpinned (no cv2 to run, no
fixtures in the reference); everything around it is pinned: oracle/make_ref_aux_golden.py runs the reference's own
images_and_boxes_preprocessing with `cv2.resize` replaced by resize_u8 below, and tests/test_ref_aux.py requires this
module to reproduce its clips and boxes bit for bit under the same np.random seed (27 cases).  TRAIN.USE_COLOR_AUGMENTATION (off in every shipped config) is not restated.
"""
import math

import numpy as np


def resize_tables(src, dst):
    """(ofs int32 [dst], coef int16 [dst][2]) of cv::resize INTER_LINEAR along one axis"""
    scale = 1.0 / (float(dst) / float(src))
    ofs = np.zeros(dst, np.int32)
    coef = np.zeros((dst, 2), np.int16)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(math.floor(float(f)))
        f = np.float32(f - np.float32(s))
        if s < 0:
            f, s = np.float32(0), 0
        if s >= src - 1:
            f, s = np.float32(0), src - 1
        c0 = np.float32(np.float32(1.0) - f) * np.float32(2048.0)
        c1 = np.float32(f * np.float32(2048.0))
        ofs[d] = s
        coef[d, 0] = int(np.clip(np.rint(c0), -32768, 32767))     # cvRound: nearest, ties to even
        coef[d, 1] = int(np.clip(np.rint(c1), -32768, 32767))
    return ofs, coef


def resize_u8(img, new_w, new_h):
    """img (H, W, 3) uint8 -> (new_h, new_w, 3) uint8"""
    h, w = img.shape[:2]
    xo, xc = resize_tables(w, new_w)
    yo, yc = resize_tables(h, new_h)
    src = img.astype(np.int64)
    x1 = np.minimum(xo + 1, w - 1)
    hor = src[:, xo, :] * xc[None, :, 0, None].astype(np.int64) + src[:, x1, :] * xc[None, :, 1, None].astype(np.int64)
    y1 = np.minimum(yo + 1, h - 1)
    b0 = yc[:, 0].astype(np.int64)[:, None, None]
    b1 = yc[:, 1].astype(np.int64)[:, None, None]
    v = (((b0 * (hor[yo] >> 4)) >> 16) + ((b1 * (hor[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def _scaled_size(height, width, size):
    """image_processor.py:189-205 / 226-251: short side -> size, long side floor()ed"""
    if (width <= height and width == size) or (height <= width and height == size):
        return height, width
    if width < height:
        return int(math.floor((float(height) / width) * size)), size
    return size, int(math.floor((float(width) / height) * size))


def images_and_boxes_preprocessing(imgs, split, crop_size, spatial_shift_pos, cfg, boxes=None, rng=np.random):
    """imgs: list of (H, W, 3) uint8 BGR frames; split 1 = train.  -> (clip (3, T, crop, crop) float32,
    boxes).  Random draws in the reference's order: jitter scale (uniform), crop y then x (randint),
    flip (uniform)."""
    height, width = imgs[0].shape[:2]
    if boxes is not None:
        boxes = boxes.astype(np.float64).copy()
        boxes[:, [0, 2]] *= width
        boxes[:, [1, 3]] *= height
        boxes[:, [0, 2]] = np.minimum(width - 1., np.maximum(0., boxes[:, [0, 2]]))
        boxes[:, [1, 3]] = np.minimum(height - 1., np.maximum(0., boxes[:, [1, 3]]))
    if split == 1:
        lo, hi = cfg.TRAIN.JITTER_SCALES
        size = int(round(1.0 / rng.uniform(1.0 / hi, 1.0 / lo)))
        nh, nw = _scaled_size(height, width, size)
        if (nh, nw) != (height, width):
            if boxes is not None:
                boxes = boxes * float(nh) / height if width < height else boxes * float(nw) / width
            imgs = [resize_u8(im, nw, nh) for im in imgs]
        y0 = int(rng.randint(0, nh - crop_size)) if nh > crop_size else 0
        x0 = int(rng.randint(0, nw - crop_size)) if nw > crop_size else 0
        if (nh, nw) == (crop_size, crop_size):
            y0 = x0 = 0
        imgs = [im[y0:y0 + crop_size, x0:x0 + crop_size] for im in imgs]
        if boxes is not None and (nh, nw) != (crop_size, crop_size):
            boxes[:, [0, 2]] -= x0
            boxes[:, [1, 3]] -= y0
        if rng.uniform() < 0.5:
            if boxes is not None:
                b = boxes.copy()
                b[:, 0] = crop_size - boxes[:, 2] - 1
                b[:, 2] = crop_size - boxes[:, 0] - 1
                boxes = b
            imgs = [im[:, ::-1] for im in imgs]
    else:
        nh, nw = _scaled_size(height, width, cfg.TEST.SCALE)
        if (nh, nw) != (height, width):
            imgs = [resize_u8(im, nw, nh) for im in imgs]
            if boxes is not None:
                boxes *= (float(nh) / height) if width < height else (float(nw) / width)
        if cfg.AVA.FORCE_TEST_FLIP and cfg.DATASET == 'ava':
            if boxes is not None:
                b = boxes.copy()
                b[:, 0] = nw - boxes[:, 2] - 1
                b[:, 2] = nw - boxes[:, 0] - 1
                boxes = b
            imgs = [im[:, ::-1] for im in imgs]
        y0 = int(math.ceil((nh - crop_size) / 2))
        x0 = int(math.ceil((nw - crop_size) / 2))
        if nh > nw:
            y0 = 0 if spatial_shift_pos == 0 else (nh - crop_size if spatial_shift_pos == 2 else y0)
        else:
            x0 = 0 if spatial_shift_pos == 0 else (nw - crop_size if spatial_shift_pos == 2 else x0)
        imgs = [im[y0:y0 + crop_size, x0:x0 + crop_size] for im in imgs]
        if boxes is not None:
            boxes[:, [0, 2]] -= x0
            boxes[:, [1, 3]] -= y0
    mean = np.array(cfg.DATA_MEAN, dtype=np.float32)
    std = np.array(cfg.DATA_STD, dtype=np.float32)
    out = []
    for im in imgs:
        chw = np.ascontiguousarray(im.transpose(2, 0, 1)).astype(np.float32)
        chw = (chw / np.float32(255.0)).astype(np.float32)
        for c in range(3):
            chw[c] = chw[c] - mean[c]
            chw[c] = chw[c] / std[c]
        out.append(chw)
    clip = np.stack(out, axis=1)                       # (3, T, H, W), BGR
    if not cfg.MODEL.USE_BGR:
        clip = clip[::-1]
    if boxes is not None:
        boxes[:, [0, 2]] = np.minimum(crop_size - 1., np.maximum(0., boxes[:, [0, 2]]))
        boxes[:, [1, 3]] = np.minimum(crop_size - 1., np.maximum(0., boxes[:, [1, 3]]))
    return np.ascontiguousarray(clip), boxes
