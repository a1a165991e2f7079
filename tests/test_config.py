"""core.config keeps the reference's surface: strict YAML merge, KEY VAL overrides, derived keys;
the presets equal the reference's configs/*.yaml (checked where /root/reference is mounted)."""
import glob
import json
import os

import pytest

REF = "/root/reference/configs"


def _tree(cfg):
    return json.loads(json.dumps(cfg))


def test_defaults_and_inference():
    from core import config as C
    C.reset_cfg()
    cfg = C.config
    assert cfg.TRAIN.VIDEO_LENGTH == 32 and cfg.NUM_GPUS == 8 and cfg.RNG_SEED == 2
    assert cfg.FBO_NL.NL_DROPOUT_ON is True and cfg.RESNETS.STRIDE_1X1 is False  # dead-but-required keys
    C.assert_and_infer_cfg()
    assert cfg.SOLVER.STEPS == [0, 100000, 120000, 140000]
    assert cfg.LFB.NUM_LFB_FEAT == 500


def test_strict_merge_rules():
    from core import config as C
    C.reset_cfg()
    with pytest.raises(KeyError):
        C.merge_dicts({"NOT_A_KEY": 1}, C.config)
    with pytest.raises(ValueError):
        C.merge_dicts({"NUM_GPUS": "eight"}, C.config)
    C.cfg_from_list(["TRAIN.BATCH_SIZE", "16", "MODEL.USE_AFFINE", "True", "DATASET", "ava"])
    assert C.config.TRAIN.BATCH_SIZE == 16 and C.config.MODEL.USE_AFFINE is True and C.config.DATASET == "ava"
    with pytest.raises(AssertionError):
        C.cfg_from_list(["TRAIN.NOPE", "1"])
    C.reset_cfg()


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")
def test_every_reference_yaml_loads_unmodified():
    from core import config as C
    files = sorted(glob.glob(os.path.join(REF, "*.yaml")))
    assert len(files) == 26
    for f in files:
        C.reset_cfg()
        C.cfg_from_file(f)
        C.assert_and_infer_cfg()
        assert C.config.MODEL.USE_AFFINE is True  # every shipped config freezes BN
    C.reset_cfg()


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")
def test_presets_equal_the_reference_yamls():
    from core import config as C
    from vlfb.presets import PRESETS, load_preset
    for name in PRESETS:
        load_preset(name)
        mine = _tree(C.config)
        C.reset_cfg()
        C.cfg_from_file(os.path.join(REF, name + ".yaml"))
        C.assert_and_infer_cfg()
        assert mine == _tree(C.config), name
    C.reset_cfg()


def test_lr_policy_with_warmup():
    from vlfb.presets import load_preset
    from utils import lr_policy
    load_preset("ava_r50_lfb_nl")
    got = [round(float(lr_policy.get_lr_at_iter(i)), 6) for i in (0, 1999, 2000, 99999, 100000, 120000, 140000)]
    assert got == [0.01, 0.04, 0.04, 0.04, 0.004, 0.0004, 4e-05]
    load_preset("charades_r50_baseline")
    assert abs(float(lr_policy.get_lr_at_iter(0)) - 0.02) < 1e-9
    assert abs(float(lr_policy.get_lr_at_iter(20000)) - 0.002) < 1e-9
