"""Checkpoint import/export rules on the host: BN folding, 2-D -> 3-D
inflation, classifier rule, momentum policy, resume policy and the on-disk format.  A dict-backed
stand-in replaces the device Engine; tests/test_workspace_gpu.py round-trips through the real one."""
import os
import pickle

import numpy as np
import pytest


class DictEngine(object):
    """what utils.checkpoints needs from vlfb.engine.Engine, on host arrays"""

    def __init__(self, model, train=True):
        self.train = train
        self.lr = 0.0
        rng = np.random.default_rng(0)
        self.p = {n: rng.standard_normal(model.param_init_net.fills[n].shape).astype(np.float32)
                  for n in model.GetAllParams()}
        self.m = {n: rng.standard_normal(self.p[n].shape).astype(np.float32) for n in model.TrainableParams()}
        model.engine = self

    def feed_params(self, d):
        for k, v in d.items():
            assert v.shape == self.p[k].shape and v.dtype == np.float32
            self.p[k] = v.copy()

    def feed_momentum(self, d):
        for k, v in d.items():
            assert v.shape == self.m[k].shape
            self.m[k] = v.copy()

    def fetch_param(self, n):
        return self.p[n]

    def fetch_momentum(self, n):
        return self.m[n]

    def set_lr(self, lr):
        self.lr = float(lr)


def build(preset="charades_r50_baseline", train=True, extra=()):
    from vlfb.presets import load_preset
    from models.model_builder_video import ModelBuilder
    load_preset(preset, ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 1] + list(extra))
    split = "train" if train else "test"
    m = ModelBuilder(train=train, split=split, name=split)
    m.build_model(suffix="_" + split)
    return m


def test_fold_spatial_bn_matches_inference_bn():
    from utils import checkpoints as ck
    rng = np.random.default_rng(1)
    c = 16
    blobs = {"res2_0_branch2a_bn_s": rng.uniform(0.5, 1.5, c).astype(np.float32),
             "res2_0_branch2a_bn_b": rng.standard_normal(c).astype(np.float32),
             "res2_0_branch2a_bn_rm": rng.standard_normal(c).astype(np.float32),
             "res2_0_branch2a_bn_riv": rng.uniform(0.1, 2.0, c).astype(np.float32),
             "res2_0_branch2a_w": rng.standard_normal((c, 4, 1, 1)).astype(np.float32),
             "res_conv1_bn_s": np.ones(4, np.float32), "res_conv1_bn_b": np.zeros(4, np.float32)}
    ref = {k: v.copy() for k, v in blobs.items()}
    folded = ck.remove_spatial_bn_layers({"blobs": blobs})
    assert folded == ["res2_0_branch2a", "res_conv1"]
    assert "res2_0_branch2a_bn_rm" not in blobs and "res2_0_branch2a_bn_riv" not in blobs
    x = rng.standard_normal((5, c))
    bn = (x - ref["res2_0_branch2a_bn_rm"]) / np.sqrt(ref["res2_0_branch2a_bn_riv"] + 1e-5) \
        * ref["res2_0_branch2a_bn_s"] + ref["res2_0_branch2a_bn_b"]
    aff = x * blobs["res2_0_branch2a_bn_s"] + blobs["res2_0_branch2a_bn_b"]
    np.testing.assert_allclose(aff, bn, rtol=1e-5, atol=1e-6)
    # an affine pair without statistics is left alone
    assert np.array_equal(blobs["res_conv1_bn_s"], ref["res_conv1_bn_s"])


def test_fit_blob_inflation_and_classifier_rule():
    from utils import checkpoints as ck
    rng = np.random.default_rng(2)
    w2d = rng.standard_normal((8, 4, 3, 3)).astype(np.float32)
    w3d = ck.fit_blob("res2_0_branch2a_w", w2d, (8, 4, 3, 3, 3))
    assert w3d.shape == (8, 4, 3, 3, 3) and w3d.dtype == np.float32
    np.testing.assert_allclose(w3d.sum(axis=2), w2d, rtol=1e-6)          # response to a static clip is kept
    assert np.array_equal(w3d[:, :, 0], w3d[:, :, 2])
    one = ck.fit_blob("conv1_w", w2d, (8, 4, 1, 3, 3))
    np.testing.assert_array_equal(one[:, :, 0], w2d)
    with pytest.raises(AssertionError):
        ck.fit_blob("conv1_w", w2d, (8, 4, 3, 5, 5))
    with pytest.raises(AssertionError):
        ck.fit_blob("res_conv1_bn_s", np.zeros(4, np.float32), (8,))
    # classifier: element count decides, then reshape
    assert ck.fit_blob("pred_w", np.zeros((400, 2048), np.float32), (157, 2048)) is None
    assert ck.fit_blob("pred_w", np.zeros((157, 2048, 1, 1, 1), np.float32), (157, 2048)).shape == (157, 2048)
    assert ck.fit_blob("pred_b", np.float64(np.arange(80)), (80,)).dtype == np.float32


def test_read_blobs_accepts_reference_python2_pickles(tmp_path):
    from utils import checkpoints as ck
    # protocol-2 pickle with byte-string keys, as Python 2 writes them
    arr = np.arange(6, dtype=np.float32).reshape(2, 3)
    p = tmp_path / "py2.pkl"
    with open(p, "wb") as fh:
        pickle.dump({b"blobs": {b"pred_w": arr, b"lr": np.float32(0.01), b"model_iter": 7}}, fh, protocol=2)
    got = ck.read_blobs(str(p))
    assert set(got) == {"pred_w", "lr", "model_iter"} and np.array_equal(got["pred_w"], arr)
    # bare dict (convert_model output)
    q = tmp_path / "bare.pkl"
    ck.write_blobs(str(q), {"pred_w": arr, "lr": 0.00125}, wrap=False)
    assert set(ck.read_blobs(str(q))) == {"pred_w", "lr"}


def test_save_then_resume_round_trip_with_momentum(tmp_path):
    from core.config import config as cfg
    from utils import checkpoints as ck
    m = build()
    eng = DictEngine(m)
    eng.lr = 0.0375
    cfg.CHECKPOINT.DIR = str(tmp_path)
    path = os.path.join(ck.create_and_get_checkpoint_directory(), "c2_model_iter120.pkl")
    ck.save_model_params(m, path, model_iter=119)
    raw = pickle.load(open(path, "rb"))
    assert set(raw) == {"blobs"} and raw["blobs"]["model_iter"] == 120
    assert raw["blobs"]["conv1_w"].shape == (64, 3, 5, 7, 7)              # reference layout
    assert "conv1_w_momentum" in raw["blobs"] and "res_conv1_bn_s_momentum" not in raw["blobs"]
    n_train = len(m.TrainableParams())
    assert len(raw["blobs"]) == 2 + len(m.GetAllParams()) + n_train

    m2 = build()                                          # (resets the config)
    cfg.CHECKPOINT.DIR = str(tmp_path)
    eng2 = DictEngine(m2)
    for k in eng2.p:
        eng2.p[k] = np.zeros_like(eng2.p[k])
    older = os.path.join(ck.get_checkpoint_directory(), "c2_model_iter20.pkl")
    ck.save_model_params(m2, older, model_iter=19)
    assert ck.find_checkpoint() and ck.get_checkpoint_resume_file() == path     # newest wins
    cfg.CHECKPOINT.RESUME = True
    cfg.TRAIN.PARAMS_FILE = ""                            # (the YAML names a Kinetics file + CONVERT_MODEL)
    start = ck.load_model_from_params_file(m2)
    assert start == 120 and abs(m2.current_lr - 0.0375) < 1e-7 and abs(eng2.lr - 0.0375) < 1e-7
    for k in eng.p:
        assert np.array_equal(eng.p[k], eng2.p[k]), k
    for k in eng.m:
        assert np.array_equal(eng.m[k], eng2.m[k]), k


def test_pretrained_file_policy_no_momentum_missing_blobs_and_reset(tmp_path):
    from core.config import config as cfg
    from utils import checkpoints as ck
    m = build()
    eng = DictEngine(m)
    src = {k: v + 1.0 for k, v in eng.p.items()}
    del src["pred_w"]                                    # missing blob keeps its initial value
    src["pred_b"] = np.zeros(400, np.float32)            # Kinetics classifier: wrong size, skipped
    src["conv1_w_momentum"] = np.ones_like(eng.p["conv1_w"])
    src["res2_0_branch2b_w"] = src["res2_0_branch2b_w"][:, :, 0]        # 2-D weight, inflated over kT=1
    src["lr"] = np.float32(0.5)
    src["model_iter"] = 4000
    f = tmp_path / "pretrained.pkl"
    ck.write_blobs(str(f), src)
    cfg.CHECKPOINT.DIR = str(tmp_path / "run")
    cfg.CHECKPOINT.RESUME = True                          # no checkpoint there -> params file is used
    cfg.TRAIN.PARAMS_FILE = str(f)
    cfg.CHECKPOINT.CONVERT_MODEL = False
    cfg.TRAIN.RESET_START_ITER = False
    cfg.TRAIN.RESUME_FROM_BATCH_SIZE = 32                 # file was trained at batch 32, we run batch 1
    before_pred_w, before_pred_b = eng.p["pred_w"].copy(), eng.p["pred_b"].copy()
    mom_before = eng.m["conv1_w"].copy()
    start = ck.load_model_from_params_file(m)
    assert start == int(4000 * 32 / cfg.TRAIN.BATCH_SIZE) and m.current_lr == 0.5
    np.testing.assert_array_equal(eng.p["res4_0_branch2a_w"], src["res4_0_branch2a_w"])
    np.testing.assert_array_equal(eng.p["res2_0_branch2b_w"][:, :, 0], src["res2_0_branch2b_w"])
    np.testing.assert_array_equal(eng.p["pred_w"], before_pred_w)
    np.testing.assert_array_equal(eng.p["pred_b"], before_pred_b)
    np.testing.assert_array_equal(eng.m["conv1_w"], mom_before)           # pre-trained momentum is ignored
    cfg.TRAIN.RESET_START_ITER = True
    assert ck.load_model_from_params_file(m) == 0


def test_test_net_loads_weights_only_and_needs_lr_unless_reset(tmp_path):
    from core.config import config as cfg
    from utils import checkpoints as ck
    m = build(train=False)
    eng = DictEngine(m, train=False)
    src = {k: v * 2 for k, v in eng.p.items()}
    f = tmp_path / "w.pkl"
    ck.write_blobs(str(f), src)
    cfg.TRAIN.RESET_START_ITER = False
    with pytest.raises(Exception, match="No lr blob"):
        ck.load_model_from_params_file_for_test(m, str(f))
    cfg.TRAIN.RESET_START_ITER = True
    ck.load_model_from_params_file_for_test(m, str(f))
    np.testing.assert_array_equal(eng.p["res5_2_branch2c_w"], src["res5_2_branch2c_w"])


def test_convert_model_drops_classifier_and_folds_bn(tmp_path):
    from core.config import config as cfg
    from utils import checkpoints as ck
    cfg.CHECKPOINT.DIR = str(tmp_path)
    src = {"conv1_w": np.ones((64, 3, 7, 7), np.float32), "res_conv1_bn_s": np.full(64, 2.0, np.float32),
           "res_conv1_bn_b": np.zeros(64, np.float32), "res_conv1_bn_rm": np.ones(64, np.float32),
           "res_conv1_bn_riv": np.full(64, 4.0 - 1e-5, np.float32), "pred_w": np.zeros((1000, 2048), np.float32),
           "pred_b": np.zeros(1000, np.float32), "conv1_w_momentum": np.zeros((64, 3, 7, 7), np.float32),
           "model_iter": 9, "lr": 0.1, "epoch": 3}
    f = tmp_path / "imagenet.pkl"
    ck.write_blobs(str(f), src)
    out = ck.convert_model(str(f))
    got = ck.read_blobs(out)
    assert set(got) == {"conv1_w", "res_conv1_bn_s", "res_conv1_bn_b", "lr"} and got["lr"] == 0.00125
    np.testing.assert_allclose(got["res_conv1_bn_s"], 1.0, rtol=1e-6)
    np.testing.assert_allclose(got["res_conv1_bn_b"], -1.0, rtol=1e-6)
