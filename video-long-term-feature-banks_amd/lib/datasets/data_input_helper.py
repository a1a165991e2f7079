"""Clip preprocessing with the pixel work on the GPU.

Mirror of the reference's lib/datasets/data_input_helper.py:70-139 `images_and_boxes_preprocessing`
(+ the geometry helpers of lib/datasets/image_processor.py:66-251): the HOST decides the geometry
exactly as the reference does (jitter scale, crop offsets, flip, box transforms -- a few scalars per
clip, drawn from the same `np.random` calls in the same order), ONE kernel (`vlfb_clip_preprocess`)
does resize + crop + flip + /255 + mean/std + BGR->RGB for all frames of the clip and writes the model's
`data` input in its device layout.  The reference runs cv2.resize / flip / NumPy per frame on
cfg.MODEL.SAMPLE_THREADS host threads and ships 19.3 MB of fp32 per clip through the blob queue; here
4.8 MB of uint8 cross PCIe (or nothing, if a GPU decoder produced the frames).

TRAIN.USE_COLOR_AUGMENTATION (off in every shipped config) is not implemented and raises.
"""
import ctypes as C
import math

import numpy as np
import torch

from core.config import config as cfg
from vlfb import hip

_COEF_BITS = 11


def resize_tables(src, dst):
    """cv::resize INTER_LINEAR (8-bit path) along one axis: left source index and the two 11-bit
    weights per destination index"""
    scale = 1.0 / (float(dst) / float(src))
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    low, high = s < 0, s >= src - 1
    f[low | high] = 0.0
    s[low] = 0
    s[high] = src - 1
    one = np.float32(1 << _COEF_BITS)
    coef = np.stack([np.rint((np.float32(1.0) - f) * one), np.rint(f * one)], axis=1)
    return s.astype(np.int32), np.clip(coef, -32768, 32767).astype(np.int16)


def _scaled_size(height, width, size):
    if (width <= height and width == size) or (height <= width and height == size):
        return height, width
    if width < height:
        return int(math.floor((float(height) / width) * size)), size
    return size, int(math.floor((float(width) / height) * size))


def _clip_boxes(boxes, height, width):
    boxes[:, [0, 2]] = np.minimum(width - 1., np.maximum(0., boxes[:, [0, 2]]))
    boxes[:, [1, 3]] = np.minimum(height - 1., np.maximum(0., boxes[:, [1, 3]]))
    return boxes


def plan_clip(height, width, split, crop_size, spatial_shift_pos, boxes=None, rng=np.random):
    """geometry of one clip: dict(resized_h, resized_w, y0, x0, flip) in the kernel's convention, and
    the transformed boxes"""
    if boxes is not None:
        boxes = np.asarray(boxes, dtype=np.float64).copy()
        boxes[:, [0, 2]] *= width
        boxes[:, [1, 3]] *= height
        boxes = _clip_boxes(boxes, height, width)
    if split == 1:
        if cfg.TRAIN.USE_COLOR_AUGMENTATION:
            raise NotImplementedError("TRAIN.USE_COLOR_AUGMENTATION is not available in the device pipeline")
        lo, hi = cfg.TRAIN.JITTER_SCALES
        size = int(round(1.0 / rng.uniform(1.0 / hi, 1.0 / lo)))
        nh, nw = _scaled_size(height, width, size)
        if (nh, nw) != (height, width) and boxes is not None:
            boxes = boxes * float(nh) / height if width < height else boxes * float(nw) / width
        y0 = x0 = 0
        if (nh, nw) != (crop_size, crop_size):
            if nh > crop_size:
                y0 = int(rng.randint(0, nh - crop_size))
            if nw > crop_size:
                x0 = int(rng.randint(0, nw - crop_size))
            if boxes is not None:
                boxes[:, [0, 2]] -= x0
                boxes[:, [1, 3]] -= y0
        flip = bool(rng.uniform() < 0.5)
        if flip:
            if boxes is not None:
                b = boxes.copy()
                b[:, 0] = crop_size - boxes[:, 2] - 1
                b[:, 2] = crop_size - boxes[:, 0] - 1
                boxes = b
            x0 = x0 + crop_size - 1            # the window is walked right to left
    else:
        nh, nw = _scaled_size(height, width, cfg.TEST.SCALE)
        if (nh, nw) != (height, width) and boxes is not None:
            boxes *= (float(nh) / height) if width < height else (float(nw) / width)
        flip = bool(cfg.AVA.FORCE_TEST_FLIP and cfg.DATASET == 'ava')
        if flip and boxes is not None:
            b = boxes.copy()
            b[:, 0] = nw - boxes[:, 2] - 1
            b[:, 2] = nw - boxes[:, 0] - 1
            boxes = b
        y0 = int(math.ceil((nh - crop_size) / 2))
        x0 = int(math.ceil((nw - crop_size) / 2))
        if nh > nw:
            y0 = 0 if spatial_shift_pos == 0 else (nh - crop_size if spatial_shift_pos == 2 else y0)
        else:
            x0 = 0 if spatial_shift_pos == 0 else (nw - crop_size if spatial_shift_pos == 2 else x0)
        if boxes is not None:
            boxes[:, [0, 2]] -= x0
            boxes[:, [1, 3]] -= y0
        if flip:
            x0 = nw - 1 - x0                   # flipped BEFORE the crop: column x0 of the mirror image
    if boxes is not None:
        boxes = _clip_boxes(boxes, crop_size, crop_size)
    return dict(resized_h=nh, resized_w=nw, y0=y0, x0=x0, flip=int(flip)), boxes


_table_cache = {}


def _tables(src, dst, device):
    key = (src, dst, str(device))
    if key not in _table_cache:
        ofs, coef = resize_tables(src, dst)
        _table_cache[key] = (torch.as_tensor(ofs).to(device), torch.as_tensor(coef).to(device))
    return _table_cache[key]


def images_and_boxes_preprocessing(imgs, split, crop_size, spatial_shift_pos, boxes=None, out=None,
                                   out_dtype=torch.float32, w_pad=0, c_pad=3, device="cuda:0", rng=np.random):
    """imgs: (T, H, W, 3) uint8 BGR frames (NumPy array, list of frames, or a device tensor).
    Returns (clip, boxes): `clip` is `out` if given (a device tensor viewing frame 0 of the destination
    clip in the layout [T][crop][w_pad + crop + w_pad][c_pad], e.g. a slice of the engine's data blob),
    else a new tensor of that layout.  split == 1 is train (reference convention)."""
    if not torch.is_tensor(imgs):
        imgs = torch.as_tensor(np.ascontiguousarray(np.stack(list(imgs)) if not isinstance(imgs, np.ndarray) else imgs))
    assert imgs.dtype == torch.uint8 and imgs.dim() == 4 and imgs.shape[3] == 3, "frames must be (T, H, W, 3) uint8"
    frames = imgs.to(device).contiguous()
    T, H, W = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
    plan, boxes = plan_clip(H, W, split, crop_size, spatial_shift_pos, boxes, rng)
    wtot = crop_size + 2 * w_pad
    if out is None:
        out = torch.zeros(T, crop_size, wtot, c_pad, device=device, dtype=out_dtype)
    assert out.is_contiguous() and out.numel() == T * crop_size * wtot * c_pad, "destination has the wrong size"
    d = hip.ClipDesc()
    d.frames, d.src_h, d.src_w = T, H, W
    d.resized_h, d.resized_w = plan["resized_h"], plan["resized_w"]
    d.crop_h = d.crop_w = crop_size
    d.y0, d.x0, d.flip = plan["y0"], plan["x0"], plan["flip"]
    d.to_rgb = 0 if cfg.MODEL.USE_BGR else 1
    d.w_left, d.w_total, d.c_pad = w_pad, wtot, c_pad
    for c in range(3):
        d.mean[c] = float(np.float32(cfg.DATA_MEAN[c]))
        d.std[c] = float(np.float32(cfg.DATA_STD[c]))
    xo = xc = yo = yc = None
    if (d.resized_h, d.resized_w) != (H, W):
        xo, xc = _tables(W, d.resized_w, frames.device)
        yo, yc = _tables(H, d.resized_h, frames.device)
    hip.call("vlfb_clip_preprocess", C.byref(d), hip.ptr(frames), hip.ptr(xo), hip.ptr(xc), hip.ptr(yo), hip.ptr(yc),
             hip.ptr(out), hip.dtype_code(out.dtype))
    torch.cuda.current_stream().synchronize()     # `frames` may be a temporary
    return out, boxes
