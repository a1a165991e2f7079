"""Helpers the builders need from the reference's lib/utils/misc.py (batch/crop per GPU and
blob-name scoping); the Caffe2-specific debugging helpers there are out of scope."""
from core.config import config as cfg

_SCOPE_SEP = "/"


def get_batch_size(split):
    """per-GPU batch (reference: misc.py:68-72)"""
    if split in ("test", "val"):
        return int(cfg.TEST.BATCH_SIZE / cfg.NUM_GPUS)
    if split == "train":
        return int(cfg.TRAIN.BATCH_SIZE / cfg.NUM_GPUS)
    raise ValueError("unknown split %r" % (split,))


def get_crop_size(split):
    return cfg.TEST.CROP_SIZE if split in ("test", "val") else cfg.TRAIN.CROP_SIZE


def unscope_name(blob_name):
    blob_name = str(blob_name)
    return blob_name[blob_name.rfind(_SCOPE_SEP) + 1:]
