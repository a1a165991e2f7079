"""A stand-in for the Caffe2 model helper that RECORDS what a model-definition module asks of it.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference's model definition -- lib/models/resnet_video.py, resnet_helper.py, nonlocal_helper.py, lfb_helper.py,
head_helper.py -- never touches Caffe2 directly: every operator goes through the `model` object it is handed
(`model.ConvNd(...)`, `model.net.Sum(...)`, `model.param_init_net.ConstantFill(...)`), and the only attribute it reads
is `model.split` (nonlocal_helper.py:167,182).  So the reference's OWN files can be executed in this container, without
Caffe2, against an object that writes down every call: operator name, positional arguments, keyword arguments.  That
transcript is the reference-held description of the graph -- layer order, blob names, kernel / stride / pad / dilation
of every convolution, channel counts, initialisers, which blobs are summed, transposed, reshaped and how.

oracle/make_ref_graph_golden.py runs the reference through this recorder and commits the transcripts
(tests/golden/ref_graphs.json.gz); tests/test_ref_graph.py runs THIS repo's lib/models through the same recorder and
requires the identical transcript, call for call.

The reference's ModelBuilder methods the definition calls (ConvNd, AffineNd, Relu_, ... model_builder_video.py:159-250)
are recorded AS CALLS, not expanded: both sides of the comparison stop at the same interface.

Return values follow Caffe2's convention for `net.Op(inputs, outputs, **kw)`: one output -> that blob, several -> a tuple
(the reference unpacks `blob, shape = model.Reshape(...)`, lfb_helper.py:49).  Blobs are plain strings, which support the
one thing the reference does to a BlobReference besides passing it on: `blob + '_4d'` (head_helper.py:101).
"""
import json


def _plain(v):
    """JSON-able, order-preserving, type-normalised copy of an argument"""
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    if isinstance(v, dict):
        return {str(k): _plain(x) for k, x in sorted(v.items())}
    if isinstance(v, bool) or v is None or isinstance(v, str):
        return v
    if isinstance(v, int):
        return int(v)
    if isinstance(v, float):
        return float(v)
    if hasattr(v, "item"):                       # numpy scalar
        return _plain(v.item())
    return repr(v)


class _Net(object):
    def __init__(self, owner, prefix):
        self._owner = owner
        self._prefix = prefix

    def __getattr__(self, op):
        if op.startswith("__"):
            raise AttributeError(op)
        return self._owner._op(self._prefix + op)


class RecordingModel(object):
    def __init__(self, split, train, inplace_relu):
        self.split = split
        self.train = train
        self.inplace_relu = inplace_relu        # cfg.MODEL.ALLOW_INPLACE_RELU of whoever drives the recorder
        self.calls = []
        self.net = _Net(self, "net.")
        self.param_init_net = _Net(self, "param_init_net.")

    def _op(self, name):
        def call(*args, **kwargs):
            self.calls.append([name, _plain(args), _plain(kwargs)])
            outs = args[1] if len(args) > 1 else kwargs.get("blob_out")
            if isinstance(outs, (list, tuple)):
                return tuple(outs) if len(outs) > 1 else outs[0]
            return outs
        return call

    def __getattr__(self, op):
        if op.startswith("__"):
            raise AttributeError(op)
        return self._op(op)

    def Relu_(self, blob_in):
        """the one helper-facing method whose output name is decided inside the model helper
        (reference model_builder_video.py:169-174: in place, or `<blob>_relu`)"""
        self.calls.append(["Relu_", [_plain(blob_in)], {}])
        return blob_in if self.inplace_relu else blob_in + "_relu"

    # -- what the reference's AffineNd composite touches besides operators (model_builder_video.py:236-240) ----------
    class _Proto(object):
        def __init__(self):
            self.external_input = []

    def bookkeeping(self):
        """enable params / weights / biases / net.Proto().external_input (plain lists) for composites that register
        parameters themselves"""
        self.params, self.weights, self.biases = [], [], []
        proto = RecordingModel._Proto()
        self.net.Proto = lambda: proto
        self._proto = proto
        return self

    def registry(self):
        return {"params": [str(x) for x in self.params], "weights": [str(x) for x in self.weights],
                "biases": [str(x) for x in self.biases], "external_input": list(self._proto.external_input)}

    def transcript(self):
        return json.loads(json.dumps(self.calls))   # (what a reader of the committed fixture sees)
