"""Checkpoint import / export in the reference's on-disk format.

Format (reference lib/utils/checkpoints.py:421-459): a pickle of `{'blobs': {unscoped_name: ndarray}}`
holding every parameter in Caffe2 layout ((Cout,Cin,kT,kH,kW) / (out,in) / (C,)), `<param>_momentum`
for the trainable ones, plus the scalars `model_iter` (next iteration) and `lr`.  Files written here
use pickle protocol 2 so the Python-2 reference can read them back; files written by the reference
(Python-2 pickles) are read with latin-1 decoding.

Import rules restated from the reference (file:line in each function):
  * SpatialBN statistics are folded into the affine pair            (checkpoints.py:88-116)
  * `epoch/model_iter/lr` and `*_momentum` are dropped on conversion (checkpoints.py:119-146)
  * `pred*` blobs load only when the element count matches, then are reshaped (checkpoints.py:316-331)
  * 4-D (2-D conv) weights are inflated to 5-D by repeating over kT and dividing by kT (:333-360)
  * blobs missing from the file keep their initial values           (checkpoints.py:311-313)

Parameters live in the model's Engine (flat device buckets); nothing here touches Caffe2.
"""
import logging
import os
import pickle
import re
from collections import OrderedDict

import numpy as np

from core.config import config as cfg
import utils.misc as misc

logger = logging.getLogger(__name__)

_CKPT_RE = re.compile(r"^c2_model_iter(\d+)\.pkl$")
_NON_PARAM_FIELDS = ("epoch", "model_iter", "lr")
BN_EPS = 1e-5


# ---- files ------------------------------------------------------------------------------------
def get_checkpoint_directory():
    """<CHECKPOINT.DIR>/checkpoints (checkpoints.py:247-252)"""
    if not cfg.CHECKPOINT.DIR:
        raise Exception("No cfg.CHECKPOINT.DIR specified.")
    return os.path.abspath(os.path.join(cfg.CHECKPOINT.DIR, "checkpoints"))


def create_and_get_checkpoint_directory():
    path = get_checkpoint_directory()
    os.makedirs(path, exist_ok=True)
    return path


def _checkpoint_iters(path):
    if not os.path.isdir(path):
        return []
    found = []
    for f in os.listdir(path):
        m = _CKPT_RE.match(f)
        if m:
            found.append(int(m.group(1)))
    return sorted(found)


def get_checkpoint_resume_file():
    """the newest c2_model_iter<N>.pkl, or None (checkpoints.py:51-69)"""
    path = get_checkpoint_directory()
    iters = _checkpoint_iters(path)
    return os.path.join(path, "c2_model_iter%d.pkl" % iters[-1]) if iters else None


def find_checkpoint():
    return bool(_checkpoint_iters(get_checkpoint_directory()))


def read_blobs(path):
    """-> OrderedDict name -> value.  Accepts both `{'blobs': {...}}` and a bare dict (what
    convert_model writes), Python-2 or Python-3 pickles, bytes or str keys."""
    with open(path, "rb") as fh:
        try:
            obj = pickle.load(fh, encoding="latin1")
        except TypeError:      # not a py2 pickle issue; re-raise the real error
            fh.seek(0)
            obj = pickle.load(fh)
    if isinstance(obj, dict) and ("blobs" in obj or b"blobs" in obj):
        obj = obj.get("blobs", obj.get(b"blobs"))
    out = OrderedDict()
    for k, v in obj.items():
        out[k.decode("latin1") if isinstance(k, bytes) else str(k)] = v
    return out


def write_blobs(path, blobs, wrap=True):
    """atomic: a uniquely named temporary in the destination directory, then rename (several processes --
    one per GPU -- may run the same start-up code; they must never share a half-written file)"""
    import tempfile
    fd, tmp = tempfile.mkstemp(prefix=os.path.basename(path) + ".", suffix=".tmp", dir=os.path.dirname(path) or ".")
    try:
        with os.fdopen(fd, "wb") as fh:
            pickle.dump(dict(blobs=dict(blobs)) if wrap else dict(blobs), fh, protocol=2)
        os.replace(tmp, path)
    except Exception:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise


# ---- conversions ------------------------------------------------------------------------------
def remove_spatial_bn_layers(c2cls_weights):
    """Fold every `<layer>_bn_{s,b,rm,riv}` quadruple into the affine pair the frozen-BN graph uses:
    s' = s / sqrt(var + 1e-5), b' = b - mean * s'   (checkpoints.py:88-116).  In place."""
    blobs = c2cls_weights["blobs"] if "blobs" in c2cls_weights else c2cls_weights
    layers = sorted({k[:k.find("_bn_")] for k in blobs if "_bn_" in k})
    for layer in layers:
        rm, riv = layer + "_bn_rm", layer + "_bn_riv"
        if rm not in blobs or riv not in blobs:
            continue                       # already an affine pair
        # the reference's arithmetic, in the dtype of the file (float32) and in its order of operations, so that a
        # converted file is the same file bit for bit (tests/test_ref_aux.py)
        gamma, beta = np.asarray(blobs[layer + "_bn_s"]), np.asarray(blobs[layer + "_bn_b"])
        std = np.sqrt(np.asarray(blobs.pop(riv)) + BN_EPS)
        mean = np.asarray(blobs.pop(rm))
        blobs[layer + "_bn_s"] = gamma / std
        blobs[layer + "_bn_b"] = beta - mean * gamma / std
    return layers


def remove_non_param_fields(c2cls_weights):
    blobs = c2cls_weights["blobs"] if "blobs" in c2cls_weights else c2cls_weights
    for f in _NON_PARAM_FIELDS:
        blobs.pop(f, None)


def remove_momentum(c2cls_weights):
    blobs = c2cls_weights["blobs"] if "blobs" in c2cls_weights else c2cls_weights
    for k in [k for k in blobs if k.endswith("_momentum")]:
        del blobs[k]


def load_and_convert_caffe2_cls_model(model_file_name):
    """an image-classification Caffe2 checkpoint -> frozen-affine blobs (checkpoints.py:135-149)"""
    weights = {"blobs": read_blobs(model_file_name)}
    remove_non_param_fields(weights)
    remove_momentum(weights)
    remove_spatial_bn_layers(weights)
    return weights


def convert_model(model_path):
    """CHECKPOINT.CONVERT_MODEL: drop the classifier (`pred*`) and any momentum, pin lr, write
    <checkpoint dir>/converted_model.pkl as a bare dict (checkpoints.py:152-183)"""
    from vlfb import dist
    out_path = os.path.join(create_and_get_checkpoint_directory(), "converted_model.pkl")
    error = None
    if dist.rank() == 0:          # one process per GPU: rank 0 converts, the others wait for the file
        try:
            blobs = load_and_convert_caffe2_cls_model(model_path)["blobs"]
            for k in [k for k in blobs if "pred" in k or "momentum" in k]:
                del blobs[k]
            blobs["lr"] = 0.00125
            write_blobs(out_path, blobs, wrap=False)
        except Exception as e:    # (a missing / corrupt file must not leave the other ranks in the barrier forever)
            error = e
    # every rank learns whether the conversion worked BEFORE anyone returns: a failure raises everywhere
    ok = dist.all_ok(error is None)
    if error is not None:
        raise error
    if not ok:
        raise RuntimeError("convert_model: rank 0 failed to convert %r" % (model_path,))
    # ... and whether EVERY rank sees the file, again before anyone moves on: a rank that raised alone would leave the others
    # in their next collective (broadcast_parameters) forever
    seen = os.path.exists(out_path)
    if not dist.all_ok(seen):
        raise RuntimeError("convert_model: %s was written by rank 0 but is not visible on every rank (here, rank %d: %s); "
                           "the checkpoint directory must be shared by all ranks"
                           % (out_path, dist.rank(), "visible" if seen else "NOT visible"))
    return out_path


def fit_blob(name, value, want_shape):
    """Adapt one file blob to the shape the graph wants; None = leave the initial value.
    (checkpoints.py:316-366: classifier rule, 2-D -> 3-D inflation, final shape assert)"""
    value = np.asarray(value)
    want_shape = tuple(int(d) for d in want_shape)
    if "pred" in name:
        if int(np.prod(want_shape)) != int(value.size):
            logger.info("%s (classifier) found but unmatching (not loaded): %s ---> %s",
                        name, value.shape, want_shape)
            return None
        value = value.reshape(want_shape)
    if value.ndim != len(want_shape):
        if not (value.ndim == 4 and len(want_shape) == 5 and value.shape[:2] == want_shape[:2]
                and value.shape[-2:] == want_shape[-2:]):
            raise AssertionError("Workspace blob %s with shape %s does not match weights file shape %s"
                                 % (name, want_shape, value.shape))
        kt = want_shape[2]
        value = np.repeat(value[:, :, None, :, :], kt, axis=2) / float(kt)
        logger.info("%s inflated %s ---> %s", name, value.shape[:2] + value.shape[3:], want_shape)
    if tuple(value.shape) != want_shape:
        raise AssertionError("Workspace blob %s with shape %s does not match weights file shape %s"
                             % (name, want_shape, value.shape))
    return value.astype(np.float32, copy=False)


# ---- model <-> file ---------------------------------------------------------------------------
def _engine(model):
    eng = getattr(model, "engine", None)
    if eng is None or not getattr(eng, "allocated", True):
        raise RuntimeError("checkpoint I/O needs an allocated engine (workspace.CreateNet(model.net) first)")
    return eng


def _param_shape(model, name):
    return tuple(model.param_init_net.fills[name].shape)


def initialize_master_gpu_model_params(model, weights_file, load_momentum=True):
    """Feed parameters (and, when resuming a training net, momentum) from `weights_file`.
    Returns (model_iter, prev_lr)   (checkpoints.py:271-376)"""
    eng = _engine(model)
    blobs = read_blobs(weights_file)
    model_iter = int(np.asarray(blobs["model_iter"]).reshape(-1)[0]) if "model_iter" in blobs else 0
    if "lr" in blobs:
        prev_lr = float(np.asarray(blobs["lr"]).reshape(-1)[0])
    elif cfg.TRAIN.RESET_START_ITER:
        prev_lr = 1.0
    else:
        raise Exception("No lr blob found.")

    params, momenta = {}, {}
    trainable = set(model.TrainableParams())
    for p in model.GetAllParams():
        name = misc.unscope_name(p)
        if name not in blobs:
            logger.info("%s not found", name)
            continue
        v = fit_blob(name, blobs[name], _param_shape(model, name))
        if v is not None:
            params[name] = v
    if model.train and load_momentum:
        for p in model.params:
            name = misc.unscope_name(p)
            key = name + "_momentum"
            if p not in trainable:
                continue
            if key not in blobs:
                logger.info("%s not found", key)
                continue
            v = fit_blob(key, blobs[key], _param_shape(model, name))
            if v is not None:
                momenta[name] = v
    eng.feed_params(params)
    if momenta:
        eng.feed_momentum(momenta)
    eng.set_lr(prev_lr)
    return model_iter, prev_lr


def broadcast_parameters(model):
    """rank 0's parameters and momentum to every replica.  The reference copies through host memory
    GPU by GPU (checkpoints.py:386-407); here each GPU is a process and it is one RCCL broadcast of
    the flat buckets."""
    from vlfb import dist
    if dist.world_size() == 1:
        return
    import torch.distributed as td
    eng = _engine(model)
    td.broadcast(eng.flat_param, 0)
    td.broadcast(eng.flat_frozen, 0)
    if eng.train:
        td.broadcast(eng.flat_mom, 0)
    eng.refresh_operands(all_params=True)


def initialize_params_from_file(model, weights_file, load_momentum=True):
    model_iter, prev_lr = initialize_master_gpu_model_params(model, weights_file, load_momentum)
    broadcast_parameters(model)
    return model_iter, prev_lr


def load_model_from_params_file_for_test(model, weights_file):
    initialize_params_from_file(model=model, weights_file=weights_file)


def resume_from(start_model_iter):
    """rescale the iteration when pre-training used another batch size (misc.py / checkpoints.py:239-244)"""
    assert cfg.TRAIN.RESUME_FROM_BATCH_SIZE > 0
    return int(start_model_iter * cfg.TRAIN.RESUME_FROM_BATCH_SIZE / cfg.TRAIN.BATCH_SIZE)


def load_model_from_params_file(model):
    """Start-of-training policy (checkpoints.py:186-236): convert if asked; an existing checkpoint
    wins when CHECKPOINT.RESUME, else TRAIN.PARAMS_FILE (without momentum), else from scratch.
    Returns the iteration to start from."""
    use_checkpoint = bool(cfg.CHECKPOINT.RESUME and find_checkpoint())
    if cfg.TRAIN.PARAMS_FILE and cfg.CHECKPOINT.CONVERT_MODEL:
        assert cfg.MODEL.USE_AFFINE
        cfg.TRAIN.PARAMS_FILE = convert_model(cfg.TRAIN.PARAMS_FILE)
    if cfg.TRAIN.PARAMS_FILE and not use_checkpoint:
        start_iter, prev_lr = initialize_params_from_file(model=model, weights_file=cfg.TRAIN.PARAMS_FILE,
                                                          load_momentum=False)
        model.current_lr = prev_lr
        if cfg.TRAIN.RESUME_FROM_BATCH_SIZE > 0:
            start_iter = resume_from(start_iter)
        if cfg.TRAIN.RESET_START_ITER:
            start_iter = 0
    elif use_checkpoint:
        start_iter, prev_lr = initialize_params_from_file(model=model, weights_file=get_checkpoint_resume_file())
        model.current_lr = prev_lr
    else:
        start_iter = 0
        logger.info("No checkpoint found; training from scratch...")
    # the iteration also seeds the dropout masks: a resumed run continues the mask sequence
    eng = getattr(model, "engine", None)
    if eng is not None:
        eng.iteration = int(start_iter)
    return start_iter


def save_model_params(model, params_file, model_iter):
    """momentum of the trainable parameters, then all (computed) parameters, `model_iter + 1`, `lr`
    (checkpoints.py:421-459)"""
    from vlfb import dist
    if dist.rank() != 0:           # replicas hold identical parameters: rank 0 writes
        return
    eng = _engine(model)
    out = OrderedDict()
    out["model_iter"] = model_iter + 1
    out["lr"] = np.float32(eng.lr)
    trainable = set(model.TrainableParams())
    for p in model.GetParams():
        if p in trainable:
            out[misc.unscope_name(p) + "_momentum"] = eng.fetch_momentum(misc.unscope_name(p))
    for p in model.GetParams() + model.GetComputedParams():
        name = misc.unscope_name(p)
        if name not in out:
            out[name] = eng.fetch_param(name)
    try:
        write_blobs(params_file, out, wrap=True)
    except Exception:
        logger.warning("save_model_params: dump parameters failed.")
