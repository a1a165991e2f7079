"""Clip preprocessing: the oracle's restatement of OpenCV's 8-bit bilinear resize,
and the host-side geometry of the device pipeline against the oracle (CPU); the kernel itself against the
oracle bit for bit (GPU)."""
import numpy as np
import pytest

from oracle import preprocess as op


def _cfg(dataset="ava", extra=()):
    from vlfb.presets import load_preset
    from core.config import config as cfg
    load_preset("ava_r50_lfb_nl" if dataset == "ava" else "charades_r50_baseline", ["NUM_GPUS", 1] + list(extra))
    return cfg


def _frames(rng, t, h, w):
    base = rng.integers(0, 256, (t, h // 8 + 2, w // 8 + 2, 3)).astype(np.uint8)
    big = np.repeat(np.repeat(base, 8, axis=1), 8, axis=2)[:, :h, :w]            # blocky content + noise
    return np.clip(big.astype(np.int32) + rng.integers(-20, 21, big.shape), 0, 255).astype(np.uint8)


def test_resize_restatement_properties():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3)).astype(np.uint8)
    assert np.array_equal(op.resize_u8(img, 53, 37), img)                       # same size: identity
    flat = np.full((20, 30, 3), 201, np.uint8)
    assert np.array_equal(op.resize_u8(flat, 77, 41), np.full((41, 77, 3), 201, np.uint8))
    # against float bilinear with half-pixel centres: at most 1 grey level apart
    big = op.resize_u8(img, 80, 56).astype(np.float64)
    ys = np.clip((np.arange(56) + 0.5) * 37 / 56 - 0.5, 0, 36)
    xs = np.clip((np.arange(80) + 0.5) * 53 / 80 - 0.5, 0, 52)
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    y1, x1 = np.minimum(y0 + 1, 36), np.minimum(x0 + 1, 52)
    wy, wx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
    f = img.astype(np.float64)
    ref = (f[y0][:, x0] * (1 - wx) + f[y0][:, x1] * wx) * (1 - wy) + (f[y1][:, x0] * (1 - wx) + f[y1][:, x1] * wx) * wy
    assert np.abs(big - ref).max() <= 1.0
    ofs, coef = op.resize_tables(340, 256)
    assert ofs.min() == 0 and ofs.max() <= 339 and np.all(np.diff(ofs) >= 0)
    assert np.all(coef.sum(axis=1) == 2048)


@pytest.mark.parametrize("split,seed", [(1, 0), (1, 1), (1, 2), (1, 3), (0, 0)])
def test_host_geometry_matches_oracle(split, seed):
    """same np.random stream -> same boxes, and the kernel plan reproduces the oracle's crop of the
    oracle's resized frames (checked on the host with the oracle's resize)"""
    from datasets import data_input_helper as dh
    cfg = _cfg()
    rng = np.random.default_rng(seed)
    h, w = (256, 340) if seed % 2 == 0 else (360, 270)
    frames = _frames(rng, 2, h, w)
    boxes = np.array([[0.1, 0.2, 0.5, 0.9], [0.0, 0.0, 1.0, 1.0], [0.55, 0.3, 0.8, 0.6]])
    crop = 224 if split == 1 else 256
    for shift in ((1,) if split == 1 else (0, 1, 2)):
        clip, b_ref = op.images_and_boxes_preprocessing(list(frames), split, crop, shift, cfg, boxes.copy(),
                                                        np.random.RandomState(seed))
        plan, b = dh.plan_clip(h, w, split, crop, shift, boxes.copy(), np.random.RandomState(seed))
        assert np.array_equal(b, b_ref)
        rs = [op.resize_u8(f, plan["resized_w"], plan["resized_h"]) if (plan["resized_h"], plan["resized_w"]) != (h, w) else f
              for f in frames]
        cols = plan["x0"] + (-1 if plan["flip"] else 1) * np.arange(crop)
        win = np.stack([r[plan["y0"]:plan["y0"] + crop][:, cols] for r in rs])          # (T, crop, crop, 3) BGR u8
        want = ((win.astype(np.float32) / np.float32(255.0)) - np.float32(0.45)) / np.float32(0.225)
        got = clip.transpose(1, 2, 3, 0)[..., ::-1]                                       # back to BGR, THWC
        assert np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["train0", "train1", "train5", "test_shift0", "test_shift2", "test_flip", "noresize"])
def test_kernel_matches_oracle_bit_for_bit(case):
    import torch
    from datasets import data_input_helper as dh
    extra = ["AVA.FORCE_TEST_FLIP", True] if case == "test_flip" else []
    cfg = _cfg(extra=extra)
    rng = np.random.default_rng(7)
    split = 1 if case.startswith("train") or case == "noresize" else 0
    seed = int(case[5:]) if case.startswith("train") else 11
    h, w = (240, 320) if case != "train1" else (330, 250)
    crop = 224 if split == 1 else 256
    if case == "noresize":
        cfg.TRAIN.JITTER_SCALES = [240, 240]
    frames = _frames(rng, 3, h, w)
    boxes = np.array([[0.2, 0.1, 0.7, 0.8], [0.4, 0.4, 0.95, 1.0]])
    shift = 0 if case == "test_shift0" else 2 if case == "test_shift2" else 1
    want, b_ref = op.images_and_boxes_preprocessing(list(frames), split, crop, shift, cfg, boxes.copy(),
                                                    np.random.RandomState(seed))
    out, b = dh.images_and_boxes_preprocessing(frames, split, crop, shift, boxes.copy(), w_pad=4, c_pad=4,
                                               rng=np.random.RandomState(seed))
    assert np.array_equal(b, b_ref)
    got = out.cpu().numpy()                                       # (T, crop, crop + 8, 4)
    assert np.all(got[:, :, :4] == 0) and np.all(got[:, :, 4 + crop:] == 0) and np.all(got[..., 3] == 0)
    assert np.array_equal(got[:, :, 4:4 + crop, :3].transpose(3, 0, 1, 2), want)
    # bf16 destination = rounding of the fp32 result; straight into a slice of a larger buffer
    big = torch.zeros(2, 3, crop, crop + 8, 4, device="cuda", dtype=torch.bfloat16)
    dh.images_and_boxes_preprocessing(frames, split, crop, shift, None, out=big[1], w_pad=4, c_pad=4,
                                      rng=np.random.RandomState(seed))
    ref16 = torch.as_tensor(want).to(torch.bfloat16).float().numpy()
    assert np.array_equal(big[1].float().cpu().numpy()[:, :, 4:4 + crop, :3].transpose(3, 0, 1, 2), ref16)
    assert float(big[0].abs().max()) == 0.0
