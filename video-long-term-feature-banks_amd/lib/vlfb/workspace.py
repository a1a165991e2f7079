"""A `caffe2.python.workspace`-shaped facade over vlfb engines.

tools/train_net.py, tools/test_net.py and tools/lfb_loader.py in the reference talk to the model
through workspace.FeedBlob / FetchBlob / RunNetOnce / CreateNet / RunNet with scoped blob names
(`gpu_0/data_train`, `gpu_0/pred`, ...).  The same calls work here: nets are ModelBuilder.net
objects, blobs are device tensors owned by the model's Engine.
"""
import numpy as np

_engines = {}      # id(net) -> engine (insertion order = CreateNet order)
_models = {}       # id(net) -> model
_dtype = {"value": "bf16"}
_clock = {"t": 0}  # RunNet counter: a blob that several nets of one scope produce belongs to the last one run


def set_compute_dtype(name):
    """'bf16' / 'fp16' (throughput paths), 'mix' / 'split' (parity-grade: fp32 storage, split-bf16 products; 'mix' with an
    fp16 backward) or 'fp32' (exact-fp32 MFMA) for engines created afterwards"""
    if name not in ("bf16", "fp16", "fp32", "split", "mix"):
        raise ValueError("set_compute_dtype: %r is not one of bf16, fp16, fp32, split, mix" % (name,))
    _dtype["value"] = name


def register(model):
    _models[id(model.net)] = model
    return model


def _unscoped(name):
    name = str(name)
    return name[name.rfind("/") + 1:]


def _scope_engines(name):
    scope = str(name).split("/")[0] + "/" if "/" in str(name) else None
    return [e for e in _engines.values() if scope is None or e.model.scope == scope]


def _has(eng, base):
    if base in eng.param_views or base == "lr":
        return True
    if base.endswith("_momentum") and eng.train and base[:-9] in eng.train_layout:
        return True
    if base.endswith("_grad") and eng.train and base[:-5] in eng.grad_views:
        return True
    b = eng.env.get(base) or (eng.env.get(base[:-5]) if base.endswith("_grad") else None)
    return b is not None and not getattr(b.root, "dead", False)


def _engine_for_blob(name):
    """Caffe2 semantics: the nets of one workspace share a blob namespace per device scope, so
    `gpu_0/prob` is whatever the LAST net that ran wrote there.  Among the engines of the scope that
    have the blob, the one that ran most recently answers (parameters are shared storage anyway)."""
    base = _unscoped(name)
    cands = [e for e in _scope_engines(name) if _has(e, base)]
    if not cands:
        raise KeyError("no engine owns blob %r (CreateNet the model first)" % name)
    return max(cands, key=lambda e: getattr(e, "_last_run", -1))


def CreateNet(net, input_shapes=None, dtype=None):
    """Plan + allocate the engine behind `net` (a ModelBuilder.net).  `input_shapes` maps the
    unscoped input blob names to their reference-layout shapes; if omitted they must have been
    fed with FeedBlob before (their shapes are taken from the fed arrays)."""
    from vlfb.engine import Engine
    model = _models[id(net)]
    # a second net of the same scope (test_model next to train_model, tools/train_net.py:60-77) shares the
    # parameter blobs of the first one that trains
    owner = next((e for e in _engines.values() if e.model.scope == model.scope and e.train), None)
    eng = Engine(model, dtype or _dtype["value"], share_params_with=owner if not (model.train and not model.force_fw_only) else None)
    shapes = dict(input_shapes or {})
    for k, v in _pending.items():
        if k in model.input_blob_names and k not in shapes:
            shapes[k] = v.shape
    missing = [b for b in model.input_blob_names if b not in shapes]
    if missing:
        raise KeyError("CreateNet: shapes of input blobs %r are unknown (FeedBlob them first)" % missing)
    eng.plan({k: shapes[k] for k in model.input_blob_names})
    eng.init_params()
    _engines[id(net)] = eng
    for k in list(_pending):
        if k in model.input_blob_names:
            eng.feed(k, _pending.pop(k))
    return eng


_pending = {}


def RunNetOnce(net):
    """param_init_net equivalent: (re)run the recorded fillers"""
    eng = _engines.get(id(net))
    if eng is not None:
        eng.init_params()
    return True


def RunNet(net, num_iter=1):
    """one (or more) forward[+backward+update] passes, like workspace.RunNet(model.net)"""
    eng = _engines[id(net) if not isinstance(net, str) else next(k for k, e in _engines.items() if e.model.net.name == net)]
    _clock["t"] += 1
    eng._last_run = _clock["t"]
    for _ in range(num_iter):
        eng.forward()
        if eng.train:
            eng.backward()
            if eng.comm is not None:
                eng.comm.wait()
            if eng.model.with_update:
                eng.sgd_step()
    return True


def FeedBlob(name, arr):
    arr = np.asarray(arr)
    base = _unscoped(name)
    fed = False
    done = set()           # storages already written (engines of one scope alias their parameters)
    for eng in _scope_engines(name):
        if base in eng.model.input_blob_names:
            eng.feed(base, arr)
            fed = True
        elif base in eng.param_views:
            key = eng.param_views[base].data_ptr()
            if key not in done:
                eng.feed_params({base: arr})
                done.add(key)
            fed = True
        elif base.endswith("_momentum") and eng.train and base[:-9] in eng.train_layout:
            eng.feed_momentum({base[:-9]: arr})
            fed = True
        elif base == "lr":
            eng.set_lr(float(arr))
            fed = True
    if not fed:
        _pending[base] = arr
    return True


def FetchBlob(name):
    base = _unscoped(name)
    eng = _engine_for_blob(name)
    if base in eng.param_views:
        return eng.fetch_param(base)
    if base.endswith("_momentum") and base[:-9] in eng.train_layout:
        return eng.fetch_momentum(base[:-9])
    if base.endswith("_grad") and base[:-5] in eng.grad_views:
        return eng.fetch_grad(base[:-5])
    if base == "lr":
        return np.float32(eng.lr)
    out = eng.fetch(base)
    return out.reshape(()) if base == "loss" else out


def GetEngine(net):
    """the Engine behind a created net (no Caffe2 equivalent; used by utils.checkpoints)"""
    return _engines[id(net)]


def HasBlob(name):
    try:
        FetchBlob(name)
        return True
    except KeyError:
        return False


def ResetWorkspace():
    _engines.clear()
    _models.clear()
    _pending.clear()
    return True
