"""The graph this repo's lib/models emits == the graph the REFERENCE's lib/models emits, call for call.

tests/golden/ref_graphs.json.gz holds transcripts of the reference's own `create_model` (lib/models/resnet_video.py:133
and the helper modules it drives), executed unmodified in the build container against oracle.graph_recorder.RecordingModel
by oracle/make_ref_graph_golden.py: every operator the reference asks of its model helper, with positional and keyword
arguments -- all 26 shipped configs as train / test / LFB-extraction graphs plus the option switches the configs leave
at their defaults (118 graphs).  Here this repo's builder modules run against the same recorder under the same
configuration and must produce the identical list: layer order, blob names, kernel / stride / pad / dilation / group of
every convolution, channel counts, initialisers, pool windows, RoIAlign arguments, transposes, reshapes, dropout ratios
and is_test flags, loss scale.  This pins rows M4-M11, H1-H7, L1 of SURVEY.md 8(a) (the STRUCTURE of the hot path) to the
reference's code rather than to a restatement of it.
"""
import gzip
import json
import os

import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ref_graphs.json.gz")


def _fixture():
    with gzip.open(GOLDEN, "rb") as fh:
        return json.loads(fh.read().decode())


FIXTURE = _fixture()
GRAPHS = FIXTURE["graphs"]


def _id(g):
    o = "".join(str(x).split(".")[-1][:10] + "," for x in g["overrides"])
    return "%s-%s%s%s" % (g["config"], g["split"], "-infer" if g["lfb_infer_only"] else "", "-" + o if o else "")


def _first_difference(a, b):
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            return "call %d:\n  reference: %r\n  this repo: %r" % (i, x, y)
    return "lengths differ: reference %d calls, this repo %d (first extra: %r)" % (
        len(a), len(b), (a + b)[min(len(a), len(b))])


@pytest.mark.parametrize("g", GRAPHS, ids=_id)
def test_builder_emits_the_reference_graph(g):
    from core import config as C
    from core.config import config as cfg
    from models import resnet_video
    from oracle.graph_recorder import RecordingModel
    from vlfb.presets import load_preset
    # this repo's preset of the same name + the same overrides == the reference's effective configuration for this graph
    # (its YAML through its own cfg_from_file / cfg_from_list / assert_and_infer_cfg), on every key incl. the derived ones
    load_preset(g["config"], g["overrides"])
    mine, want = _flat(json.loads(json.dumps(cfg))), _flat(g["cfg"])
    assert sorted(mine) == sorted(want) and not {k: (want[k], mine[k]) for k in want if want[k] != mine[k]}
    model = RecordingModel(split=g["split"], train=(g["split"] == "train"), inplace_relu=cfg.MODEL.ALLOW_INPLACE_RELU)
    suffix = "_{}".format(g["split"])
    resnet_video.create_model(model=model, data="data" + suffix, labels="labels" + suffix, split=g["split"],
                              lfb_infer_only=g["lfb_infer_only"], suffix=suffix)
    mine = model.transcript()
    C.reset_cfg()
    assert mine == g["calls"], _first_difference(g["calls"], mine)


def _flat(tree, prefix=""):
    out = {}
    for k, v in tree.items():
        if isinstance(v, dict):
            out.update(_flat(v, prefix + k + "."))
        else:
            out[prefix + k] = v
    return out


def test_config_defaults_equal_the_reference_defaults():
    """core/config.py:51-371 of the reference, as imported by the generator, key by key"""
    from core import config as C
    C.reset_cfg()
    mine = _flat(json.loads(json.dumps(C.config)))
    ref = _flat(FIXTURE["defaults"])
    assert sorted(mine) == sorted(ref), set(mine) ^ set(ref)
    diff = {k: (ref[k], mine[k]) for k in ref if ref[k] != mine[k]}
    assert not diff, diff


def test_presets_equal_the_reference_configurations():
    """vlfb.presets (what bench.py and the GPU tests load where the YAMLs are not mounted) == the reference's
    cfg_from_file + assert_and_infer_cfg on the YAML of the same name, on every key"""
    from core import config as C
    from vlfb.presets import PRESETS, load_preset
    ref = {g["config"]: g["cfg"] for g in GRAPHS if not g["overrides"]}
    for name in PRESETS:
        load_preset(name)
        mine = _flat(json.loads(json.dumps(C.config)))
        want = _flat(ref[name])
        diff = {k: (want[k], mine.get(k)) for k in want if want[k] != mine.get(k)}
        assert not diff, (name, diff)
    C.reset_cfg()


def test_the_fixture_covers_every_shipped_config_and_graph_kind():
    names = {g["config"] for g in GRAPHS if not g["overrides"]}
    assert len(names) == 26
    kinds = {(g["split"] == "train", g["lfb_infer_only"]) for g in GRAPHS}
    assert kinds == {(True, False), (False, False), (False, True)}
    ops = {c[0] for g in GRAPHS for c in g["calls"]}
    for op in ("ConvNd", "AffineNd", "SpatialBN", "Relu_", "MaxPool", "AveragePool", "net.Sum", "net.BatchMatMul", "Softmax",
               "net.Div", "RoIAlign", "LayerNorm", "Dropout", "FC", "SigmoidCrossEntropyLoss", "SoftmaxWithLoss",
               "StopGradient", "Transpose", "Reshape", "net.Concat"):
        assert op in ops, op


# ---- one level down: the reference ModelBuilder's own composites ---------------------------------------------------------
def _expanded_id(g):
    return _id(dict(g, lfb_infer_only=False))


@pytest.mark.parametrize("g", FIXTURE["expanded"], ids=_expanded_id)
def test_composites_expand_like_the_reference_composites(g):
    """Relu_, Conv3dBN, Conv3dAffine (reference model_builder_video.py:169-221), taken from the reference's class and
    run on the recorder by the generator, against the same three methods of THIS repo's ModelBuilder run on the
    recorder here: the primitives they ask for (ConvNd with its initialisers / no_bias / dilations / group, AffineNd
    with its name and in-place flag, SpatialBN with epsilon / momentum / is_test, the gamma re-fill) must be the same"""
    from core import config as C
    from core.config import config as cfg
    from models import resnet_video
    from models.model_builder_video import ModelBuilder
    from oracle.graph_recorder import RecordingModel

    class Expanded(RecordingModel):
        pass
    for meth in ("Relu_", "Conv3dBN", "Conv3dAffine"):
        setattr(Expanded, meth, ModelBuilder.__dict__[meth])
    from vlfb.presets import load_preset
    load_preset(g["config"], g["overrides"])
    model = Expanded(split=g["split"], train=(g["split"] == "train"), inplace_relu=cfg.MODEL.ALLOW_INPLACE_RELU)
    suffix = "_{}".format(g["split"])
    resnet_video.create_model(model=model, data="data" + suffix, labels="labels" + suffix, split=g["split"],
                              lfb_infer_only=False, suffix=suffix)
    mine = model.transcript()
    C.reset_cfg()
    assert mine == g["calls"], _first_difference(g["calls"], mine)


@pytest.mark.parametrize("case", FIXTURE["affine_nd"], ids=lambda c: ",".join(sorted(c["kwargs"])))
def test_affine_nd_registers_what_the_reference_registers(case):
    """reference model_builder_video.py:223-244 (recorded with its params / weights / biases / external_input
    bookkeeping) against the real ModelBuilder.AffineNd of this repo: parameter names, shapes, fill values, which list
    each lands in, the operator's inputs and its output (in place or not), and that shared parameters register nothing"""
    from models.model_builder_video import ModelBuilder
    from core import config as C
    C.reset_cfg()
    C.assert_and_infer_cfg()
    m = ModelBuilder(train=True, split="train", name="t")
    kw = case["kwargs"]
    if kw.get("share_with"):
        m.AffineNd("x", kw["share_with"], kw["dim_in"])          # the owner of the shared parameters
    n_ops, n_params = len(m.net.ops), len(m.params)
    ret = m.AffineNd(**kw)
    assert ret == case["returns"]
    op = m.net.ops[-1]
    name, (ins, out), _ = case["calls"][-1]
    assert len(m.net.ops) == n_ops + 1 and name == "net." + op.type
    assert op.inputs == ins and op.outputs == [out]
    reg = case["registry"]
    assert m.params[n_params:] == reg["params"]
    assert [p for p in m.weights if p in reg["params"]] == reg["weights"] or not reg["params"]
    assert [p for p in m.biases if p in reg["params"]] == reg["biases"] or not reg["params"]
    if reg["params"]:
        for fill, (_, pname), fkw in case["calls"][:-1]:
            mine = m.param_init_net.fills[pname]
            assert "param_init_net." + mine.fill == fill and list(mine.shape) == fkw["shape"]
            assert mine.kwargs["value"] == fkw["value"]
        # no gradient for scale / bias (caffe2_customized_ops/video/affine_nd_op.cc:45-53)
        assert set(reg["params"]) <= m.affine_params
