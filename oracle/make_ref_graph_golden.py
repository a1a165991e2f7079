"""Run the REFERENCE's own model definition and commit what it asks of the model helper.

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference; the GPU box has none):

    python oracle/make_ref_graph_golden.py            # writes tests/golden/ref_graphs.json.gz

What runs is the reference's code, unmodified, imported from where it lies: lib/models/resnet_video.py:133
`create_model` and everything it calls (resnet_helper.py, nonlocal_helper.py, lfb_helper.py, head_helper.py),
lib/core/config.py (defaults, cfg_from_file, cfg_from_list, assert_and_infer_cfg) on the reference's configs/*.yaml, and
lib/utils/misc.py:68 `get_batch_size`.  Two things are substituted, neither on the path under test: an empty
`caffe2.python.{workspace,scope}` (utils/misc.py:43 imports them for functions that are not called here) and
`yaml.load` with the safe loader (core/config.py:427 calls it without one, which PyYAML 6 refuses); the byte-string
defaults of core/config.py (Python 2) are decoded to str so that its own type check accepts its own YAMLs under
Python 3.  The `model` object is oracle.graph_recorder.RecordingModel.

Cases: every shipped config in the train graph (split 'train') and the test graph (TEST.DATA_TYPE), the LFB configs also
as the feature-extraction graph of tools/lfb_loader.py:175-180 (lfb_infer_only), plus the option switches the configs do
not exercise (dot-product and batch-norm non-local blocks, grouped bottlenecks, FBO-NL head options, 64-frame clips).
"""
import glob
import gzip
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
OUT = os.path.join(HERE, "..", "tests", "golden", "ref_graphs.json.gz")

# option switches beyond the shipped YAMLs: (config, overrides as cfg_from_list takes them)
EXTRA = [
    ("charades_r50_baseline", ["NONLOCAL.USE_SOFTMAX", False]),
    ("ava_r50_lfb_nl", ["NONLOCAL.USE_SOFTMAX", False, "NONLOCAL.USE_SCALE", False]),
    ("charades_r50_baseline", ["NONLOCAL.USE_BN", True, "NONLOCAL.USE_AFFINE", False]),
    ("charades_r50_baseline", ["MODEL.USE_AFFINE", False]),
    ("charades_r50_baseline", ["RESNETS.NUM_GROUPS", 32, "RESNETS.WIDTH_PER_GROUP", 4]),
    ("charades_r50_baseline", ["NONLOCAL.CONV3_NONLOCAL", False]),
    ("charades_r50_baseline", ["NONLOCAL.CONV4_NONLOCAL", False, "NONLOCAL.USE_MAXPOOL", False]),
    ("charades_r50_baseline", ["NONLOCAL.USE_ZERO_INIT_CONV", False, "NONLOCAL.CONV_INIT_STD", 0.02]),
    ("charades_r50_baseline", ["MODEL.VIDEO_ARC_CHOICE", 1]),
    ("charades_r50_baseline", ["MODEL.VIDEO_ARC_CHOICE", 3, "TRAIN.VIDEO_LENGTH", 8, "TEST.VIDEO_LENGTH", 8]),
    ("charades_r50_baseline", ["MODEL.VIDEO_ARC_CHOICE", 4, "TRAIN.VIDEO_LENGTH", 64, "TEST.VIDEO_LENGTH", 64]),
    ("ava_r50_lfb_nl", ["FBO_NL.PRE_ACT", False]),
    ("ava_r50_lfb_nl", ["FBO_NL.PRE_ACT_LN", False]),
    ("ava_r50_lfb_nl", ["FBO_NL.SCALE", False, "FBO_NL.NUM_LAYERS", 1]),
    ("ava_r50_lfb_nl", ["FBO_NL.INPUT_REDUCE_DIM", False]),
    ("ava_r50_lfb_nl", ["FBO_NL.INPUT_DROPOUT_ON", False, "FBO_NL.LFB_DROPOUT_ON", False, "FBO_NL.NL_DROPOUT_ON", False]),
    ("ava_r50_lfb_nl", ["FBO_NL.LATENT_DIM", 256, "FBO_NL.DROPOUT_RATE", 0.5]),
    ("ava_r50_lfb_nl", ["ROI.XFORM_RESOLUTION", 5, "ROI.SCALE_FACTOR", 8]),
    ("ava_r50_lfb_nl", ["MODEL.DILATIONS_AFTER_CONV5", False]),
    ("ava_r50_lfb_nl", ["MODEL.ALLOW_INPLACE_RELU", False, "MODEL.ALLOW_INPLACE_SUM", False]),
    ("ava_r50_lfb_nl", ["LFB.LFB_DIM", 1024, "LFB.WINDOW_SIZE", 20]),
    ("charades_r50_lfb_nl", ["MODEL.FREEZE_BACKBONE", False]),
    ("epic_verb_r50_lfb_nl", ["TRAIN.DROPOUT_RATE", 0.0]),
]


def _import_reference():
    caffe2 = types.ModuleType("caffe2")
    python = types.ModuleType("caffe2.python")
    python.workspace = types.ModuleType("caffe2.python.workspace")
    python.scope = types.ModuleType("caffe2.python.scope")
    caffe2.python = python
    sys.modules.update({"caffe2": caffe2, "caffe2.python": python, "caffe2.python.workspace": python.workspace,
                        "caffe2.python.scope": python.scope})
    import yaml
    _load = yaml.load
    yaml.load = lambda stream, Loader=yaml.SafeLoader: _load(stream, Loader)
    sys.path.insert(0, os.path.join(REF, "lib"))
    import core.config as config
    import models.resnet_video as resnet_video
    assert config.__file__.startswith(REF) and resnet_video.__file__.startswith(REF)

    def decode(d):                     # Python-2 byte-string defaults (core/config.py: b'ava', b'' ...) -> str
        for k, v in d.items():
            if isinstance(v, dict):
                decode(v)
            elif isinstance(v, bytes):
                d[k] = v.decode()
            elif isinstance(v, (list, tuple)):
                d[k] = type(v)(x.decode() if isinstance(x, bytes) else x for x in v)
    decode(config.config)
    return config, resnet_video


def _import_reference_builder():
    """lib/models/model_builder_video.py for its composite methods (Relu_ :169, Conv3dBN :176, Conv3dAffine :200,
    AffineNd :223): pure compositions over the model-helper interface, taken from the class and bound to the recorder.
    Its Caffe2 / data-loader imports (:43-53) are satisfied with empty modules; CNNModelHelper becomes `object`."""
    python = sys.modules["caffe2.python"]
    for name in ("cnn", "core", "data_parallel_model", "dyndep"):
        mod = types.ModuleType("caffe2.python." + name)
        setattr(python, name, mod)
        sys.modules["caffe2.python." + name] = mod
    python.cnn.CNNModelHelper = object
    proto = types.ModuleType("caffe2.proto")
    proto.caffe2_pb2 = types.ModuleType("caffe2.proto.caffe2_pb2")
    sys.modules["caffe2.proto"] = proto
    sys.modules["caffe2.proto.caffe2_pb2"] = proto.caffe2_pb2
    dl = types.ModuleType("datasets.dataloader")
    dl.DataLoader = dl.get_input_db = None
    sys.modules["datasets.dataloader"] = dl
    import models.model_builder_video as mb
    assert mb.__file__.startswith(REF)
    return mb


COMPOSITES = ("Relu_", "Conv3dBN", "Conv3dAffine")

# graphs recorded a second time with the reference's composites expanded into their primitives
EXPANDED = [("ava_r50_lfb_nl", []), ("charades_r50_baseline", []),
            ("charades_r50_baseline", ["MODEL.USE_AFFINE", False]),
            ("charades_r50_baseline", ["NONLOCAL.USE_BN", True, "NONLOCAL.USE_AFFINE", False]),
            ("charades_r50_baseline", ["RESNETS.NUM_GROUPS", 32, "RESNETS.WIDTH_PER_GROUP", 4]),
            ("ava_r50_lfb_nl", ["MODEL.ALLOW_INPLACE_RELU", False, "MODEL.ALLOW_INPLACE_SUM", False])]

# argument sets for the AffineNd composite (it registers parameters itself: recorded with its bookkeeping)
AFFINE_CASES = [
    dict(blob_in="x", blob_out="x_bn", dim_in=64),
    dict(blob_in="x", blob_out="x_bn", dim_in=256, inplace=True),
    dict(blob_in="y", blob_out="y_bn", dim_in=128, share_with="x_bn"),
    dict(blob_in="y", blob_out="y_bn", dim_in=128, share_with="x_bn", inplace=True),
]


def main():
    sys.path.insert(0, os.path.join(HERE, ".."))
    from oracle.graph_recorder import RecordingModel
    config, resnet_video = _import_reference()
    import copy

    def snapshot(d):
        return {k: snapshot(v) if isinstance(v, dict) else copy.deepcopy(v) for k, v in d.items()}
    defaults = snapshot(config.config)

    def reset():
        # (the reference has no reset: restore its module-level defaults key by key)
        def rec(dst, src):
            for k in list(dst.keys()):
                if k not in src:
                    del dst[k]
            for k, v in src.items():
                if isinstance(v, dict):
                    rec(dst[k], v)
                else:
                    dst[k] = copy.deepcopy(v)
        rec(config.config, defaults)

    cases = []
    names = sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(REF, "configs", "*.yaml")))
    for name in names:
        cases.append((name, []))
    cases += EXTRA
    out = []
    for name, overrides in cases:
        reset()
        config.cfg_from_file(os.path.join(REF, "configs", name + ".yaml"))
        if overrides:
            config.cfg_from_list([str(o) if not isinstance(o, str) else o for o in overrides])
        config.assert_and_infer_cfg()
        cfg = config.config
        tree = snapshot(cfg)             # the effective configuration (create_model adds cfg.DILATIONS while it runs)
        graphs = [("train", False), (cfg.TEST.DATA_TYPE, False)]
        if cfg.LFB.ENABLED and not overrides:
            graphs.append((cfg.TEST.DATA_TYPE, True))
        for split, infer in graphs:
            model = RecordingModel(split=split, train=(split == "train"), inplace_relu=cfg.MODEL.ALLOW_INPLACE_RELU)
            suffix = "_{}".format(split)
            resnet_video.create_model(model=model, data="data" + suffix, labels="labels" + suffix, split=split,
                                      lfb_infer_only=infer, suffix=suffix)
            out.append({"config": name, "overrides": overrides, "split": split, "lfb_infer_only": infer,
                        "cfg": tree, "calls": model.transcript()})
            print("%-28s %-40s %-5s infer=%d  %4d calls" % (name, overrides and str(overrides)[:40], split, infer,
                                                          len(model.calls)))
    mb = _import_reference_builder()

    class Expanded(RecordingModel):
        pass
    for meth in COMPOSITES:
        setattr(Expanded, meth, mb.ModelBuilder.__dict__[meth])
    expanded = []
    for name, overrides in EXPANDED:
        reset()
        config.cfg_from_file(os.path.join(REF, "configs", name + ".yaml"))
        if overrides:
            config.cfg_from_list([str(o) if not isinstance(o, str) else o for o in overrides])
        config.assert_and_infer_cfg()
        cfg = config.config
        tree = snapshot(cfg)
        for split in ("train", cfg.TEST.DATA_TYPE):
            model = Expanded(split=split, train=(split == "train"), inplace_relu=cfg.MODEL.ALLOW_INPLACE_RELU)
            suffix = "_{}".format(split)
            resnet_video.create_model(model=model, data="data" + suffix, labels="labels" + suffix, split=split,
                                      lfb_infer_only=False, suffix=suffix)
            expanded.append({"config": name, "overrides": overrides, "split": split, "cfg": tree,
                             "calls": model.transcript()})
            print("expanded %-28s %-5s %4d calls" % (name, split, len(model.calls)))

    class WithAffine(RecordingModel):
        AffineNd = mb.ModelBuilder.__dict__["AffineNd"]
    affine = []
    for kw in AFFINE_CASES:
        model = WithAffine(split="train", train=True, inplace_relu=True).bookkeeping()
        ret = model.AffineNd(**kw)
        affine.append({"kwargs": kw, "returns": ret, "calls": model.transcript(), "registry": model.registry()})

    payload = {"generator": "oracle/make_ref_graph_golden.py",
               "reference": "facebookresearch/video-long-term-feature-banks lib/models/*.py, lib/core/config.py, configs/*.yaml",
               "defaults": defaults,      # core/config.py:51-371 as imported (byte strings decoded)
               "graphs": out,
               "expanded": expanded,      # the same walk with Relu_ / Conv3dBN / Conv3dAffine of model_builder_video.py run
               "affine_nd": affine}       # model_builder_video.py:223-244 on its own, with its parameter bookkeeping
    with gzip.GzipFile(OUT, "wb", mtime=0) as fh:
        fh.write(json.dumps(payload, sort_keys=True, separators=(",", ":")).encode())
    print("wrote %s: %d graphs, %d bytes" % (os.path.normpath(OUT), len(out), os.path.getsize(OUT)))


if __name__ == "__main__":
    main()
