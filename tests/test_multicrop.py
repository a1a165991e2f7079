"""AVA multi-crop testing (tools/test_net.py:48-93, lib/utils/metrics.py:599-716): merge arithmetic on the CPU,
the whole 2 x scales x 3 loop over the device engine on the GPU."""
import numpy as np
import pytest


def test_merge_arithmetic_matches_the_reference_loops():
    from vlfb import multicrop as mc
    from oracle import multicrop as omc
    rng = np.random.default_rng(2)
    R, Cc = 9, 5
    boxes = np.sort(rng.uniform(0, 1, (R, 2, 2)), axis=1).reshape(R, 4)[:, [0, 1, 2, 3]]
    boxes = np.stack([boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]], axis=1)
    boxes[:, [0, 2]] = np.sort(boxes[:, [0, 2]], axis=1)
    boxes[0] = [0.0, 0.1, 0.08, 0.3]        # only the left crop sees it
    boxes[1] = [0.93, 0.1, 1.0, 0.3]        # only the right crop
    for scale, H, W in ((224, 360, 640), (320, 240, 426), (256, 256, 256)):
        for flip in (False, True):
            logits = rng.standard_normal((3, R, Cc))
            got = mc.merge_shifts(logits, mc.shift_validity(boxes, flip, scale, H, W))
            want = omc.merge_three_shifts(logits, boxes, flip, scale, H, W)
            assert np.allclose(got, want, rtol=1e-12, atol=0, equal_nan=True), (scale, flip)
    assert np.array_equal(mc.merge_scales_and_flips([np.ones((2, 3)), 2 * np.ones((2, 3))]), 3 * np.ones((2, 3)))


@pytest.mark.gpu
def test_ava_multi_crop_loop_over_the_engine_matches_the_oracle():
    import torch
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from vlfb.multicrop import AvaMultiCropTester
    from oracle import model as om, multicrop as omc
    load_preset("ava_r50_lfb_nl", ["NUM_GPUS", 1, "TEST.BATCH_SIZE", 2, "TEST.VIDEO_LENGTH", 8, "TRAIN.VIDEO_LENGTH", 8,
                                   "AVA.TEST_MULTI_CROP", True])
    rng = np.random.default_rng(4)
    clips = [rng.integers(0, 256, (8, 72, 128, 3), dtype=np.uint8) for _ in range(2)]
    boxes = [np.array([[0.02, 0.1, 0.2, 0.9], [0.4, 0.2, 0.7, 0.8]]), np.array([[0.75, 0.05, 0.98, 0.6]])]
    params = om.synth_params(cfg, seed=3)
    R = 3
    lfb = (np.maximum(rng.standard_normal((R, 60 * 5, 2048)), 0) * 0.5).astype(np.float32)
    scales, max_crop = [56, 64, 80], 64          # crops 56, 64, 64 (the reference: 224, 256, 256)
    tester = AvaMultiCropTester(params, dtype="fp32", scales=scales, max_crop=max_crop)
    got, passes = tester.run(clips, boxes, lfb)
    want, ref_passes = omc.multi_crop_scores(cfg, {k: v for k, v in params.items()}, clips, boxes, lfb, scales, max_crop)
    assert set(passes) == set(ref_passes) and len(passes) == 18
    for key in passes:
        d = np.linalg.norm(passes[key] - ref_passes[key]) / np.linalg.norm(ref_passes[key])
        assert d < 1e-3, (key, d)
    assert got.shape == (R, 80) and np.allclose(got, want, rtol=1e-4, atol=1e-6, equal_nan=True)
