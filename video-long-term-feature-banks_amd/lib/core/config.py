"""Global experiment configuration (`from core.config import config as cfg`).

Same import path, key set, default values and merge rules as the reference's lib/core/config.py
(defaults :52-364, assert_and_infer_cfg :373-391, merge_dicts :394-420, cfg_from_file :423-428,
cfg_from_list :431-451) so the untouched configs/*.yaml load here.  Re-written for Python 3:
string defaults are `str` (the reference has bytes under unicode_literals), YAML goes through
`safe_load`, and the defaults are one nested literal instead of attribute assignments.
Keys that only the (out-of-scope) data pipeline reads are kept so the strict key check passes.
"""
import copy
import logging
from ast import literal_eval

from utils.collections import AttrDict

logger = logging.getLogger(__name__)

_DEFAULTS = {
    "DEBUG": False,
    "DATALOADER": {"MAX_BAD_IMAGES": 100},
    "DATA_MEAN": [0.45, 0.45, 0.45],
    "DATA_STD": [0.225, 0.225, 0.225],
    "TRAIN": {
        "PARAMS_FILE": "", "DATA_TYPE": "train", "BATCH_SIZE": 64,
        "RESUME_FROM_BATCH_SIZE": -1, "RESET_START_ITER": False,
        "JITTER_SCALES": [256, 480], "CROP_SIZE": 224, "USE_COLOR_AUGMENTATION": False,
        "PCA_EIGVAL": [0.225, 0.224, 0.229],
        "PCA_EIGVEC": [[-0.5675, 0.7192, 0.4009], [-0.5808, -0.0045, -0.8140],
                       [-0.5836, -0.6948, 0.4203]],
        "COMPUTE_PRECISE_BN": True, "ITER_COMPUTE_PRECISE_BN": 200,
        "EVAL_PERIOD": 4000, "DATASET_SIZE": 0,
        "VIDEO_LENGTH": 32, "SAMPLE_RATE": 2, "DROPOUT_RATE": 0.0, "TEST_AFTER_TRAIN": True,
    },
    "MODEL": {
        "NUM_CLASSES": -1, "MODEL_NAME": "", "VIDEO_ARC_CHOICE": 2, "DEPTH": 50,
        "BN_MOMENTUM": 0.9, "BN_EPSILON": 1.0000001e-5, "BN_INIT_GAMMA": 1.0,
        "FC_INIT_STD": 0.01, "MEAN": 114.75, "STD": 57.375,
        "ALLOW_INPLACE_SUM": True, "ALLOW_INPLACE_RELU": True, "ALLOW_INPLACE_RESHAPE": True,
        "MEMONGER": True, "USE_BGR": False, "USE_AFFINE": False, "SAMPLE_THREADS": 8,
        "MULTI_LABEL": True, "DILATIONS_AFTER_CONV5": True, "FREEZE_BACKBONE": False,
    },
    "RESNETS": {"NUM_GROUPS": 1, "WIDTH_PER_GROUP": 64, "STRIDE_1X1": False,
                "TRANS_FUNC": "bottleneck_transformation"},
    "TEST": {
        "PARAMS_FILE": "", "DATA_TYPE": "", "BATCH_SIZE": 64, "SCALE": 256, "CROP_SIZE": 256,
        "DATASET_SIZE": 0, "VIDEO_LENGTH": 32, "SAMPLE_RATE": 2, "CROP_SHIFT": 1,
    },
    "SOLVER": {
        "NESTEROV": True, "WEIGHT_DECAY": 0.0001, "WEIGHT_DECAY_BN": 0.0001, "MOMENTUM": 0.9,
        "LR_POLICY": "steps_with_relative_lrs", "BASE_LR": 0.1,
        "STEP_SIZES": [100000, 20000, 20000], "LRS": [1, 0.1, 0.01], "MAX_ITER": 140000,
        "STEPS": None, "GAMMA": 0.1, "SCALE_MOMENTUM": False, "SCALE_MOMENTUM_THRESHOLD": 1.1,
        "WARMUP": {"WARMUP_ON": False, "WARMUP_START_LR": 0.1, "WARMUP_END_ITER": 5000},
    },
    "CHECKPOINT": {"CHECKPOINT_MODEL": True, "CHECKPOINT_PERIOD": -1, "RESUME": True, "DIR": ".",
                   "CONVERT_MODEL": False},
    "NONLOCAL": {
        "CONV_INIT_STD": 0.01, "NO_BIAS": 0, "USE_MAXPOOL": True, "USE_SOFTMAX": True,
        "USE_ZERO_INIT_CONV": False, "USE_BN": True, "USE_SCALE": True, "USE_AFFINE": False,
        "BN_MOMENTUM": 0.9, "BN_EPSILON": 1.0000001e-5, "BN_INIT_GAMMA": 0.0,
        "LAYER_MOD": 2, "CONV3_NONLOCAL": True, "CONV4_NONLOCAL": True,
    },
    "DATADIR": "", "DATASET": "", "ROOT_GPU_ID": 0, "NUM_GPUS": 8, "CUDNN_WORKSPACE_LIMIT": 256,
    "RNG_SEED": 2, "USE_CYTHON": False, "LOG_PERIOD": 10, "PROF_DAG": False,
    "INTERPOLATION": "INTER_LINEAR", "MINIBATCH_QUEUE_SIZE": 64,
    "AVA": {
        "FRAME_LIST_DIR": "data/ava/frame_lists", "ANNOTATION_DIR": "data/ava/annotations",
        "FPS": 30, "FULL_EVAL_DURING_TRAINING": False, "DETECTION_SCORE_THRESH_TRAIN": 0.9,
        "DETECTION_SCORE_THRESH_EVAL": [0.85], "LFB_DETECTION_SCORE_THRESH": 0.9,
        "TRAIN_ON_TRAIN_VAL": False, "TEST_ON_TEST_SET": False,
        "TRAIN_LISTS": ["train.csv"], "TEST_LISTS": ["val.csv"],
        "TRAIN_BOX_LISTS": ["ava_train_v2.1.csv", "ava_train_predicted_boxes.csv"],
        "TEST_BOX_LISTS": ["ava_val_predicted_boxes.csv"],
        "TRAIN_LFB_BOX_LISTS": ["ava_train_predicted_boxes.csv"],
        "TEST_LFB_BOX_LISTS": ["ava_val_predicted_boxes.csv"],
        "TEST_MULTI_CROP": False, "TEST_MULTI_CROP_SCALES": [224, 256, 320],
        "FORCE_TEST_FLIP": False, "LFB_MAX_NUM_FEAT_PER_STEP": 5,
    },
    "EPIC": {
        "FRAME_LIST_DIR": "data/epic/frame_lists", "ANNOTATION_DIR": "data/epic/annotations",
        "TRAIN_LISTS": ["train.csv"], "TEST_LISTS": ["val.csv"],
        "ANNOTATIONS": "EPIC_train_action_labels.csv", "FPS": 30, "CLASS_TYPE": "",
        "VERB_LFB_CLIPS_PER_SECOND": 1, "NOUN_LFB_FRAMES_PER_SECOND": 1,
        "MAX_NUM_FEATS_PER_NOUN_LFB_FRAME": 10,
    },
    "CHARADES": {
        "FRAME_LIST_DIR": "data/charades/frame_lists", "TRAIN_LISTS": ["train.csv"],
        "TEST_LISTS": ["val.csv"], "FPS": 24, "NUM_TEST_CLIPS_DURING_TRAINING": 9,
        "NUM_TEST_CLIPS_FINAL_EVAL": 30, "LFB_CLIPS_PER_SECOND": 2,
    },
    "ROI": {"SCALE_FACTOR": 16, "XFORM_RESOLUTION": 7},
    "LFB": {"ENABLED": False, "MODEL_PARAMS_FILE": "", "WRITE_LFB": False, "LOAD_LFB": False,
            "LOAD_LFB_PATH": "", "LFB_DIM": 2048, "WINDOW_SIZE": 100, "FBO_TYPE": "nl"},
    "FBO_NL": {"NUM_LAYERS": 2, "PRE_ACT": True, "PRE_ACT_LN": True, "SCALE": True,
               "LATENT_DIM": 512, "INPUT_REDUCE_DIM": True, "DROPOUT_RATE": 0.2,
               "INPUT_DROPOUT_ON": True, "LFB_DROPOUT_ON": True, "NL_DROPOUT_ON": True},
    "IMG_LOAD_RETRY": 10,
    "GET_TRAIN_LFB": False,
}


def _wrap(node):
    if isinstance(node, dict):
        return AttrDict((k, _wrap(v)) for k, v in node.items())
    return copy.deepcopy(node)


config = _wrap(_DEFAULTS)
__C = config


def reset_cfg():
    """Restore every default in place (tests build several configs in one process)."""
    for k in list(config.keys()):
        del config[k]
    config.update(_wrap(_DEFAULTS))


def print_cfg():
    import pprint
    logger.info("Config:")
    logger.info(pprint.pformat(config))


def assert_and_infer_cfg():
    """Derived keys and sanity checks (reference: config.py:373-391)."""
    sol = config.SOLVER
    if sol.STEPS is None:
        edges = [0]
        for size in sol.STEP_SIZES:
            edges.append(edges[-1] + size)
        sol.STEPS = edges
    assert config.TRAIN.BATCH_SIZE % config.NUM_GPUS == 0, \
        "Train batch size should be multiple of num_gpus."
    assert config.TEST.BATCH_SIZE % config.NUM_GPUS == 0, \
        "Test batch size should be multiple of num_gpus."
    config.LFB.NUM_LFB_FEAT = config.AVA.LFB_MAX_NUM_FEAT_PER_STEP * config.LFB.WINDOW_SIZE


def _coerce(value):
    if isinstance(value, str):
        try:
            return literal_eval(value)
        except Exception:
            return value
    return value


def merge_dicts(src, dst, path=""):
    """Strict merge of `src` into `dst` with the reference's rules (config.py:394-421): an unknown key is a KeyError, a
    value whose type is not exactly the default's a ValueError (so a key whose default is None -- SOLVER.STEPS -- can only
    be set through cfg_from_list), string values are evaluated as literals first, and whatever goes wrong below the top
    level surfaces as `Exception('Error under config key: ...')`."""
    for key, raw in src.items():
        where = path + key
        if key not in dst:
            raise KeyError("Invalid key in config file: {}".format(where))
        value = _coerce(raw)
        if isinstance(value, dict):
            if not isinstance(dst[key], dict):
                raise ValueError("Type mismatch (dict vs. {}) for config key: {}".format(
                    type(dst[key]), where))
            try:
                merge_dicts(value, dst[key], where + ".")
            except BaseException as e:
                raise Exception("Error under config key: {} ({})".format(where, e))
            continue
        old = dst[key]
        if value is not None and type(old) is not type(value):
            raise ValueError("Type mismatch ({} vs. {}) for config key: {}".format(
                type(old), type(value), where))
        dst[key] = value


def cfg_from_file(filename):
    """Load a YAML experiment file and merge it into the defaults."""
    import yaml
    with open(filename, "r") as f:
        loaded = yaml.safe_load(f) or {}
    merge_dicts(loaded, config)


def cfg_from_list(args_list):
    """`KEY VAL KEY VAL ...` overrides (command line)."""
    assert len(args_list) % 2 == 0, "Specify values or keys for args"
    for dotted, raw in zip(args_list[0::2], args_list[1::2]):
        node = config
        parts = dotted.split(".")
        for part in parts[:-1]:
            assert part in node, "Config key {} not found".format(part)
            node = node[part]
        leaf = parts[-1]
        assert leaf in node, "Config key {} not found".format(leaf)
        value = _coerce(raw)
        assert node[leaf] is None or isinstance(value, type(node[leaf])), \
            "type {} does not match original type {}".format(type(value), type(node[leaf]))
        node[leaf] = value
