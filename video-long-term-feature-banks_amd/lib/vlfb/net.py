"""A small static op recorder standing in for Caffe2's NetDef / core.Net.

The reference's builders (lib/models/*.py) talk to `model.net` / `model.param_init_net` and to
CNNModelHelper methods; each call appends one operator to a NetDef.  Here each call appends an
`Op` record; vlfb.lowering later turns the recorded list into fused HIP kernel launches.
Blob references are plain strings, exactly as they are observable through the reference's
FetchBlob / checkpoint names.
"""
from collections import OrderedDict


class Op(object):
    __slots__ = ("type", "inputs", "outputs", "args")

    def __init__(self, type_, inputs, outputs, args=None):
        self.type = type_
        self.inputs = [str(b) for b in inputs]
        self.outputs = [str(b) for b in outputs]
        self.args = dict(args or {})

    def __repr__(self):
        return "%s(%s -> %s, %s)" % (self.type, ", ".join(self.inputs), ", ".join(self.outputs), self.args)


def _as_list(x):
    if x is None:
        return []
    if isinstance(x, (list, tuple)):
        return list(x)
    return [x]


class Net(object):
    """Ordered list of ops.  Any attribute access yields an op emitter, like core.Net:
    `net.Sum([a, b], out)`, `net.BatchMatMul([a, b], out, trans_a=1)`."""

    def __init__(self, name):
        self._name = name
        self.ops = []
        self._auto = 0

    def Proto(self):
        return self

    # NetDef-ish fields some callers poke at
    @property
    def name(self):
        return self._name

    @property
    def op(self):
        return self.ops

    @property
    def external_input(self):
        return _ListSink()

    def NextName(self):
        self._auto += 1
        return "%s_blob_%d" % (self._name, self._auto)

    def add(self, type_, inputs, outputs, **args):
        op = Op(type_, _as_list(inputs), _as_list(outputs), args)
        self.ops.append(op)
        outs = op.outputs
        return outs[0] if len(outs) == 1 else tuple(outs)

    def __getattr__(self, type_):
        if type_.startswith("_"):
            raise AttributeError(type_)

        def emit(inputs, outputs=None, **args):
            if outputs is None:
                outputs = [self.NextName()]
            return self.add(type_, inputs, outputs, **args)
        return emit


class _ListSink(object):
    def extend(self, items):
        pass

    def append(self, item):
        pass


class ParamInit(object):
    """One recorded filler of param_init_net."""
    __slots__ = ("name", "fill", "shape", "kwargs")

    def __init__(self, name, fill, shape, kwargs):
        self.name, self.fill, self.shape, self.kwargs = name, fill, tuple(shape), dict(kwargs)


class ParamInitNet(object):
    """param_init_net: records `<Fill>([], name, shape=..., **kw)` calls in order."""

    def __init__(self):
        self.fills = OrderedDict()

    def _record(self, fill, inputs, name, shape=None, **kw):
        name = str(name)
        if shape is None:
            # re-fill of an existing blob (e.g. ConstantFill([p], p + '_momentum', value=0))
            src = _as_list(inputs)
            shape = self.fills[str(src[0])].shape if src and str(src[0]) in self.fills else ()
            if src and str(src[0]) == name and name in self.fills:
                self.fills[name] = ParamInit(name, fill, self.fills[name].shape, kw)
                return name
        self.fills[name] = ParamInit(name, fill, shape, kw)
        return name

    def ConstantFill(self, inputs, name, shape=None, **kw):
        return self._record("ConstantFill", inputs, name, shape, **kw)

    def GaussianFill(self, inputs, name, shape=None, **kw):
        return self._record("GaussianFill", inputs, name, shape, **kw)

    def MSRAFill(self, inputs, name, shape=None, **kw):
        return self._record("MSRAFill", inputs, name, shape, **kw)

    def __getattr__(self, fill):
        if fill.startswith("_"):
            raise AttributeError(fill)

        def emit(inputs, name, shape=None, **kw):
            return self._record(fill, inputs, name, shape, **kw)
        return emit


def ssa_form(ops):
    """Version every blob name so in-place ops (Relu_, Sum, StopGradient, Reshape, ...) become
    distinct values.  Returns [(op, [input (name, version)], [output (name, version)])]."""
    version = {}
    out = []
    for op in ops:
        ins = [(b, version.get(b, 0)) for b in op.inputs]
        outs = []
        for b in op.outputs:
            version[b] = version.get(b, 0) + 1
            outs.append((b, version[b]))
        out.append((op, ins, outs))
    return out
