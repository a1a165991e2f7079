"""The reference's 26 experiment definitions (configs/*.yaml, incl. the ones BASELINE.json names), as override trees over
core.config defaults.

The reference's configs/*.yaml load unmodified through core.config.cfg_from_file; these presets
exist because /root/reference (and its configs/) is not present on the GPU box.  Only keys that
differ from the defaults are listed; tests/test_config.py checks (where the reference tree is
mounted) that every preset equals the corresponding YAML on all keys the hot path reads.
"""
from core import config as _config

_COMMON = {
    "NUM_GPUS": 8, "LOG_PERIOD": 10,
    "MODEL": {"MODEL_NAME": "resnet_video", "BN_INIT_GAMMA": 0.0, "DEPTH": 50, "VIDEO_ARC_CHOICE": 2,
              "MULTI_LABEL": True, "USE_AFFINE": True},
    "RESNETS": {"NUM_GROUPS": 1, "WIDTH_PER_GROUP": 64, "TRANS_FUNC": "bottleneck_transformation_3d"},
    "TRAIN": {"DATA_TYPE": "train", "BATCH_SIZE": 16, "COMPUTE_PRECISE_BN": False, "CROP_SIZE": 224,
              "VIDEO_LENGTH": 32, "DROPOUT_RATE": 0.3, "RESET_START_ITER": True},
    "TEST": {"BATCH_SIZE": 16, "CROP_SIZE": 256, "SCALE": 256, "VIDEO_LENGTH": 32},
    "SOLVER": {"MOMENTUM": 0.9, "NESTEROV": True, "WEIGHT_DECAY_BN": 0.0, "SCALE_MOMENTUM": True},
    "NONLOCAL": {"USE_ZERO_INIT_CONV": True, "USE_BN": False, "USE_AFFINE": True,
                 "CONV3_NONLOCAL": True, "CONV4_NONLOCAL": True, "USE_SCALE": True},
}

_CHARADES = {
    "DATASET": "charades", "DATADIR": "data/charades/frames",
    "MODEL": {"NUM_CLASSES": 157},
    "TRAIN": {"EVAL_PERIOD": 4000, "JITTER_SCALES": [256, 320], "SAMPLE_RATE": 4, "DATASET_SIZE": 7811},
    "TEST": {"DATA_TYPE": "val", "SAMPLE_RATE": 4, "DATASET_SIZE": 1814},
    "SOLVER": {"BASE_LR": 0.02, "WEIGHT_DECAY": 0.0000125},
    "CHECKPOINT": {"CHECKPOINT_PERIOD": 4000},
}

_AVA = {
    "DATASET": "ava", "DATADIR": "data/ava/frames",
    "MODEL": {"NUM_CLASSES": 80},
    "TRAIN": {"EVAL_PERIOD": 8000, "JITTER_SCALES": [256, 320], "SAMPLE_RATE": 2, "DATASET_SIZE": 235,
              "PARAMS_FILE": "pretrained_weights/r50_k400_pretrained.pkl"},
    "TEST": {"DATA_TYPE": "val", "SAMPLE_RATE": 2, "DATASET_SIZE": 64},
    "SOLVER": {"BASE_LR": 0.04, "STEP_SIZES": [100000, 20000, 20000], "LRS": [1, 0.1, 0.01, 0.001],
               "MAX_ITER": 140000, "WEIGHT_DECAY": 0.000001,
               "WARMUP": {"WARMUP_ON": True, "WARMUP_START_LR": 0.01, "WARMUP_END_ITER": 2000}},
    "CHECKPOINT": {"CHECKPOINT_PERIOD": 4000, "CONVERT_MODEL": True},
}

_EPIC = {
    "DATASET": "epic", "DATADIR": "data/epic/frames",
    "MODEL": {"MULTI_LABEL": False, "DILATIONS_AFTER_CONV5": False},
    "TRAIN": {"JITTER_SCALES": [256, 320], "SAMPLE_RATE": 2, "DATASET_SIZE": 23191,
              "PARAMS_FILE": "pretrained_weights/r50_k400_pretrained.pkl"},
    "TEST": {"DATA_TYPE": "val", "SAMPLE_RATE": 2, "DATASET_SIZE": 5281},
    "CHECKPOINT": {"CONVERT_MODEL": True},
}
_EPIC_VERB = {
    "MODEL": {"NUM_CLASSES": 125},
    "TRAIN": {"EVAL_PERIOD": 4000},
    "SOLVER": {"BASE_LR": 0.001, "STEP_SIZES": [28000, 4000, 4000], "LRS": [1, 0.1, 0.01], "MAX_ITER": 36000,
               "WEIGHT_DECAY": 0.000001},
    "CHECKPOINT": {"CHECKPOINT_PERIOD": 4000},
    "EPIC": {"CLASS_TYPE": "verb"},
}
_EPIC_NOUN = {
    "MODEL": {"NUM_CLASSES": 352},
    "TRAIN": {"EVAL_PERIOD": 5000},
    "SOLVER": {"BASE_LR": 0.0003, "STEP_SIZES": [40000, 5000, 5000], "LRS": [1, 0.1, 0.01], "MAX_ITER": 50000,
               "WEIGHT_DECAY": 0.00001},
    "CHECKPOINT": {"CHECKPOINT_PERIOD": 4000},
    "EPIC": {"CLASS_TYPE": "noun", "MAX_NUM_FEATS_PER_NOUN_LFB_FRAME": 10, "NOUN_LFB_FRAMES_PER_SECOND": 1},
}

_R101 = {"MODEL": {"DEPTH": 101, "VIDEO_ARC_CHOICE": 4}}
_R101_K400 = {"TRAIN": {"PARAMS_FILE": "pretrained_weights/r101_k400_pretrained.pkl"}}
_CHARADES_BASE = {
    "TRAIN": {"PARAMS_FILE": "pretrained_weights/r50_k400_pretrained.pkl"},
    "SOLVER": {"STEP_SIZES": [20000, 4000], "LRS": [1, 0.1], "MAX_ITER": 24000},
    "CHECKPOINT": {"CONVERT_MODEL": True},
}
# Charades LFB models train the head on a frozen, already fine-tuned backbone (PARAMS_FILE is given on the command line)
_CHARADES_LFB = {
    "MODEL": {"FREEZE_BACKBONE": True},
    "TRAIN": {"PARAMS_FILE": ""},
    "SOLVER": {"STEP_SIZES": [10000, 2000], "LRS": [1, 0.1], "MAX_ITER": 12000},
    "FBO_NL": {"PRE_ACT": False},
}
_CHARADES_R101 = {"MODEL": {"DILATIONS_AFTER_CONV5": False}}     # (the R101 Charades configs keep res5 undilated)


def _lfb(fbo, window, **kw):
    """LFB.ENABLED with feature-bank operator `fbo` over `window` bank steps; banks are written by the run unless loaded"""
    d = {"ENABLED": True, "FBO_TYPE": fbo, "WINDOW_SIZE": window}
    d.update(kw or {"WRITE_LFB": True})
    return {"LFB": d}


_EPIC_NOUN_BANK = {"LOAD_LFB": True, "LOAD_LFB_PATH": "data/epic/noun_lfb"}

# one entry per configs/*.yaml of the reference (26), same names
PRESETS = {
    "ava_r50_baseline": [_COMMON, _AVA],
    "ava_r50_lfb_nl": [_COMMON, _AVA, _lfb("nl", 60)],
    "ava_r50_lfb_nl_3l": [_COMMON, _AVA, _lfb("nl", 60), {"FBO_NL": {"NUM_LAYERS": 3}}],
    "ava_r50_lfb_avg": [_COMMON, _AVA, _lfb("avg", 60)],
    "ava_r50_lfb_max": [_COMMON, _AVA, _lfb("max", 60)],
    "ava_r101_baseline": [_COMMON, _AVA, _R101, _R101_K400],
    "ava_r101_lfb_nl": [_COMMON, _AVA, _R101, _R101_K400, _lfb("nl", 60)],
    "ava_r101_lfb_nl_3l": [_COMMON, _AVA, _R101, _R101_K400, _lfb("nl", 60), {"FBO_NL": {"NUM_LAYERS": 3}}],
    "ava_r101_lfb_avg": [_COMMON, _AVA, _R101, _R101_K400, _lfb("avg", 60)],
    "ava_r101_lfb_max": [_COMMON, _AVA, _R101, _R101_K400, _lfb("max", 60)],
    "charades_r50_baseline": [_COMMON, _CHARADES, _CHARADES_BASE],
    "charades_r50_lfb_nl": [_COMMON, _CHARADES, _CHARADES_LFB, _lfb("nl", 20)],
    "charades_r50_lfb_avg": [_COMMON, _CHARADES, _CHARADES_LFB, _lfb("avg", 20)],
    "charades_r50_lfb_max": [_COMMON, _CHARADES, _CHARADES_LFB, _lfb("max", 20)],
    "charades_r101_baseline": [_COMMON, _CHARADES, _CHARADES_BASE, _R101, _R101_K400, _CHARADES_R101],
    "charades_r101_lfb_nl": [_COMMON, _CHARADES, _CHARADES_LFB, _R101, _CHARADES_R101, _lfb("nl", 20)],
    "charades_r101_lfb_avg": [_COMMON, _CHARADES, _CHARADES_LFB, _R101, _CHARADES_R101, _lfb("avg", 20)],
    "charades_r101_lfb_max": [_COMMON, _CHARADES, _CHARADES_LFB, _R101, _CHARADES_R101, _lfb("max", 20)],
    # EPIC-Kitchens: single-label heads (Softmax / SoftmaxWithLoss), clip-level pooling, no res5 dilation
    "epic_verb_r50_baseline": [_COMMON, _EPIC, _EPIC_VERB],
    "epic_verb_r50_lfb_nl": [_COMMON, _EPIC, _EPIC_VERB, _lfb("nl", 40)],
    "epic_verb_r50_lfb_avg": [_COMMON, _EPIC, _EPIC_VERB, _lfb("avg", 40)],
    "epic_verb_r50_lfb_max": [_COMMON, _EPIC, _EPIC_VERB, _lfb("max", 40)],
    "epic_noun_r50_baseline": [_COMMON, _EPIC, _EPIC_NOUN],
    "epic_noun_r50_lfb_nl": [_COMMON, _EPIC, _EPIC_NOUN, _lfb("nl", 120, **_EPIC_NOUN_BANK)],
    "epic_noun_r50_lfb_avg": [_COMMON, _EPIC, _EPIC_NOUN, _lfb("avg", 120, **_EPIC_NOUN_BANK)],
    "epic_noun_r50_lfb_max": [_COMMON, _EPIC, _EPIC_NOUN, _lfb("max", 120, **_EPIC_NOUN_BANK)],
}


def load_preset(name, overrides=None):
    """Reset the global cfg, apply preset `name`, then `KEY VAL` style overrides, and infer."""
    if name not in PRESETS:
        raise KeyError("unknown preset %r (have: %s)" % (name, ", ".join(sorted(PRESETS))))
    _config.reset_cfg()
    for layer in PRESETS[name]:
        _config.merge_dicts(layer, _config.config)
    if overrides:
        _config.cfg_from_list([str(x) for x in overrides])
    _config.assert_and_infer_cfg()
    return _config.config
