"""The pieces either side of the hot path, chained the way tools/train_net.py + tools/lfb_loader.py chain
them in the reference: decoded frames -> device preprocessing -> baseline model in lfb_infer_only mode ->
device feature bank -> LFB model trains on sampled windows -> NaN guard -> checkpoint -> resume."""
import collections

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_infer_bank_train_checkpoint_resume(tmp_path):
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from models.model_builder_video import ModelBuilder
    from vlfb.engine import Engine
    from vlfb.lfb_bank import DeviceBank
    from vlfb import synth
    from datasets import data_input_helper as dh
    from utils import checkpoints as ck
    import utils.misc as misc

    T, CROP, N = 8, 64, 2
    ov = ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", N, "TEST.BATCH_SIZE", N, "TRAIN.VIDEO_LENGTH", T, "TEST.VIDEO_LENGTH", T,
          "TRAIN.CROP_SIZE", CROP, "TEST.CROP_SIZE", CROP, "TEST.SCALE", CROP, "TRAIN.JITTER_SCALES", [CROP, CROP + 16],
          "LFB.WINDOW_SIZE", 4]
    rng = np.random.default_rng(0)
    videos = [rng.integers(0, 256, (T, 72, 96, 3)).astype(np.uint8) for _ in range(N)]
    boxes01 = [np.array([[0.1, 0.1, 0.6, 0.9], [0.3, 0.2, 0.95, 0.8]]) for _ in range(N)]

    def clip_inputs(eng, sfx, split, seed):
        """preprocess every video straight into the engine's data blob; return proposals (R, 5)"""
        data, _ = eng.blob_padded("data" + sfx)
        rois = []
        for n in range(N):
            _, b = dh.images_and_boxes_preprocessing(videos[n], split, CROP, 1, boxes01[n].copy(), out=data[n],
                                                     w_pad=4, c_pad=4, rng=np.random.RandomState(seed + n))
            rois.append(np.concatenate([np.full((len(b), 1), n), b], axis=1))
        return np.concatenate(rois).astype(np.float32)

    # 1) LFB inference with the baseline model (tools/lfb_loader.py:155-236)
    load_preset("ava_r50_baseline", ov)
    m = ModelBuilder(train=False, split="test", name="infer")
    m.build_model(suffix="_infer_test", lfb_infer_only=True)
    eng = Engine(m, "bf16", device="cuda:0", base_seed=2)
    R = 2 * N
    eng.plan(collections.OrderedDict([("data_infer_test", (N, 3, T, CROP, CROP)), ("labels_infer_test", (R, cfg.MODEL.NUM_CLASSES)),
                                      ("proposals_infer_test", (R, 5))]))
    eng.feed_params(synth.params(m, seed=2))
    props = clip_inputs(eng, "_infer_test", 0, 5)
    eng.feed("proposals_infer_test", props)
    eng.forward()
    bank = DeviceBank(N, 8, 4, 2048, "bf16", step_base=902)
    feats, _ = eng.blob_tensor("box_pooled")
    bank.append(feats.view(R, 2048), props[:, 0].astype(np.int64), 904 + (np.arange(R) % 2))
    bank.check_no_drops()
    assert int(bank.counts().sum()) == R

    # 2) the LFB model trains on windows sampled from the bank (tools/train_net.py:100-170)
    load_preset("ava_r50_lfb_nl", ov)
    cfg.CHECKPOINT.DIR = str(tmp_path)
    cfg.TRAIN.PARAMS_FILE = ""
    m2 = ModelBuilder(train=True, split="train", name="train")
    m2.build_model(suffix="_train")
    eng2 = Engine(m2, "bf16", device="cuda:0", base_seed=2)
    K = cfg.LFB.WINDOW_SIZE * cfg.AVA.LFB_MAX_NUM_FEAT_PER_STEP
    shapes = collections.OrderedDict([("data_train", (N, 3, T, CROP, CROP)), ("labels_train", (R, cfg.MODEL.NUM_CLASSES)),
                                      ("proposals_train", (R, 5)), ("lfb_train", (R, K, 2048))])
    eng2.plan(collections.OrderedDict((k, shapes[k]) for k in m2.input_blob_names))
    eng2.feed_params(synth.params(m2, seed=2))
    lfb_in, _ = eng2.blob_tensor("lfb_train")
    labels = (rng.uniform(size=(R, cfg.MODEL.NUM_CLASSES)) < 0.05).astype(np.int32)
    m2.UpdateWorkspaceLr(0)
    losses = []
    for it in range(3):
        props = clip_inputs(eng2, "_train", 1, 100 * it)
        eng2.feed("proposals_train", props)
        eng2.feed("labels_train", labels)
        clip_of = props[:, 0].astype(np.int64)
        bank.sample_window(clip_of, np.full(R, 905), it * N + clip_of, cfg.LFB.WINDOW_SIZE,
                           cfg.AVA.LFB_MAX_NUM_FEAT_PER_STEP, seed=cfg.RNG_SEED, out=lfb_in)
        eng2.train_step(float(m2.current_lr))
        losses.append(float(eng2.fetch("loss").reshape(-1)[0]))
    assert float(lfb_in.float().abs().max()) > 0                       # the head really saw bank features
    assert np.allclose(misc.check_nan_losses(m2), losses) and all(np.isfinite(losses))

    # 3) checkpoint, resume into a fresh engine, identical parameters and momentum
    path = ck.create_and_get_checkpoint_directory() + "/c2_model_iter3.pkl"
    ck.save_model_params(m2, path, model_iter=2)
    want = {n: eng2.fetch_param(n) for n in ("pred_w", "lfb_nl0_theta_w", "conv1_w")}
    mom = eng2.fetch_momentum("pred_w")
    m3 = ModelBuilder(train=True, split="train", name="train")
    m3.build_model(suffix="_train")
    eng3 = Engine(m3, "bf16", device="cuda:0", base_seed=9)
    eng3.plan(collections.OrderedDict((k, shapes[k]) for k in m3.input_blob_names))
    eng3.init_params()
    assert ck.load_model_from_params_file(m3) == 3
    for n, v in want.items():
        assert np.array_equal(eng3.fetch_param(n), v), n
    assert np.array_equal(eng3.fetch_momentum("pred_w"), mom)


def test_momentum_correction_on_learning_rate_jumps():
    """ModelBuilder.UpdateWorkspaceLr / _SetNewLr / _CorrectMomentum (model_builder_video.py:258-314):
    when the LR changes by more than SOLVER.SCALE_MOMENTUM_THRESHOLD the update history V (= momentum
    blobs, V := mu V + lr g) of every trainable parameter is scaled by new_lr / old_lr; smaller changes
    (warm-up ramps) leave it alone; SCALE_MOMENTUM False disables it."""
    import collections
    import numpy as np
    import torch
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from models.model_builder_video import ModelBuilder
    from vlfb.engine import Engine
    from vlfb import synth
    load_preset("charades_r50_baseline", ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 1, "TRAIN.VIDEO_LENGTH", 8, "TRAIN.CROP_SIZE", 64,
                                          "SOLVER.BASE_LR", 0.02, "SOLVER.LR_POLICY", "steps_with_relative_lrs",
                                          "SOLVER.STEP_SIZES", [2, 2], "SOLVER.LRS", [1, 0.1], "SOLVER.MAX_ITER", 4,
                                          "SOLVER.SCALE_MOMENTUM", True, "SOLVER.SCALE_MOMENTUM_THRESHOLD", 1.1])
    m = ModelBuilder(train=True, split="train", name="train")
    m.build_model(suffix="_train")
    eng = Engine(m, "fp32", device="cuda:0", base_seed=2)
    batch = synth.inputs(cfg, 1, seed=3, crop=64, frames=8)
    eng.plan(collections.OrderedDict((k, v.shape) for k, v in batch.items() if k in m.input_blob_names))
    eng.init_params(seed=1)
    for k, v in batch.items():
        if k in m.input_blob_names:
            eng.feed(k, v)
    names = ["pred_w", "res3_1_branch2a_w", "conv1_w"]
    m.UpdateWorkspaceLr(0)
    assert abs(eng.lr - 0.02) < 1e-9
    eng.train_step()
    m.UpdateWorkspaceLr(1)                         # same LR: the history is untouched
    v1 = {n: eng.fetch_momentum(n).copy() for n in names}
    eng.train_step()
    v2 = {n: eng.fetch_momentum(n).copy() for n in names}
    m.UpdateWorkspaceLr(2)                         # 0.02 -> 0.002: ratio 10 > 1.1
    torch.cuda.synchronize()
    assert abs(eng.lr - 0.002) < 1e-9 and abs(m.current_lr - 0.002) < 1e-9
    for n in names:
        assert not np.array_equal(v1[n], v2[n])
        got = eng.fetch_momentum(n)
        assert np.allclose(got, v2[n] * np.float32(0.1), rtol=1e-6, atol=0), n
    # a change below the threshold only moves the LR
    v3 = {n: eng.fetch_momentum(n).copy() for n in names}
    m._SetNewLr(m.current_lr, m.current_lr * 1.05)
    torch.cuda.synchronize()
    for n in names:
        assert np.array_equal(eng.fetch_momentum(n), v3[n])
    # switched off
    cfg.SOLVER.SCALE_MOMENTUM = False
    m._SetNewLr(m.current_lr, m.current_lr * 10.0)
    torch.cuda.synchronize()
    for n in names:
        assert np.array_equal(eng.fetch_momentum(n), v3[n])
