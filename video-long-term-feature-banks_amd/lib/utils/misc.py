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


def check_nan_losses(model=None):
    """NaN guard of the training loop (reference misc.py:50-58, called every iteration from
    tools/train_net.py:160 -- one FetchBlob, i.e. one host sync, per GPU per step).

    Here the loss step keeps its last 64 values in a device ring, so the loop may call this every
    N <= 64 iterations and still see every step's loss with a single sync.  The reference logs and
    `os._exit(0)`s; a library raises instead (FloatingPointError).  Returns the losses it inspected.

    CONTRACT in a data-parallel job (a process group exists): the verdict is COLLECTIVE -- one all-reduce + one host sync --
    so every rank must call this, and at the same iterations; a call on rank 0 only, or at rank-dependent iterations,
    deadlocks where the reference's per-process check was harmless."""
    import math
    if model is not None:
        engines = [model.engine]
    else:
        from vlfb import workspace
        engines = list(workspace._engines.values())
    seen = []
    bad = None
    for eng in engines:
        if eng is None:
            continue
        losses = eng.recent_losses()
        seen.extend(losses)
        if bad is None and any(math.isnan(v) for v in losses):
            bad = getattr(eng.model, "scope", "") or "gpu_0/"
    # One process per GPU: the rank whose loss went NaN first must not leave alone -- its NaN gradient reaches the others
    # only through the next all-reduce, which would then wait for a rank that is gone.  Every rank calls this at the same
    # iterations, so the verdict is taken together (one int32 all-reduce) and every rank raises.
    from vlfb import dist
    if dist.initialized():
        if not dist.all_ok(bad is None) and bad is None:
            bad = "another rank"
    if bad is not None:
        raise FloatingPointError("NaN losses on %s" % bad)
    return seen
