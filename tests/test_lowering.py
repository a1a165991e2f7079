"""Host logic of the engine, checked on the 'meta' device (no GPU): the lowering pass fuses the
recorded reference ops as designed, gradients are routed to the right tensors, and the flat
parameter buckets follow backward-completion order."""
import collections

import pytest


def plan(preset, overrides=("NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 2), rois=5, split="train", dtype="bf16", **kw):
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from models.model_builder_video import ModelBuilder
    from vlfb.engine import Engine
    load_preset(preset, list(overrides))
    m = ModelBuilder(train=(split == "train"), split=split, name=split)
    m.build_model(suffix="_" + split, **kw)
    n = 2
    T, S = cfg.TRAIN.VIDEO_LENGTH, (cfg.TRAIN.CROP_SIZE if split == "train" else cfg.TEST.CROP_SIZE)
    sh = collections.OrderedDict()
    sfx = "_" + split
    sh["data" + sfx] = (n, 3, T, S, S)
    if cfg.DATASET == "ava":
        sh["labels" + sfx] = (rois, cfg.MODEL.NUM_CLASSES)
        sh["proposals" + sfx] = (rois, 5)
        if "lfb" + sfx in m.input_blob_names:
            sh["lfb" + sfx] = (rois, cfg.LFB.WINDOW_SIZE * cfg.AVA.LFB_MAX_NUM_FEAT_PER_STEP, 2048)
    else:
        # multi-hot rows (Charades) or one class index per clip (EPIC-Kitchens, resnet_video.py:339-350)
        sh["labels" + sfx] = (n, cfg.MODEL.NUM_CLASSES) if cfg.MODEL.MULTI_LABEL else (n,)
        if "lfb" + sfx in m.input_blob_names:
            sh["lfb" + sfx] = (n, cfg.LFB.WINDOW_SIZE, 2048)
    eng = Engine(m, dtype, dry_run=True)
    eng.plan(sh)
    return cfg, m, eng


def kinds(eng):
    return collections.Counter(type(s).__name__ for s in eng.steps)


def test_baseline_graph_fuses_to_conv_pool_attention_steps():
    cfg, m, eng = plan("charades_r50_baseline")
    k = kinds(eng)
    # 267 recorded ops -> 89 launches-with-epilogues; no standalone Relu / Sum / AffineNd survives
    assert len(m.net.ops) == 267 and len(eng.steps) == 89
    assert k == {"ConvStep": 73, "PoolStep": 8, "AttentionStep": 5, "DropoutStep": 1, "FCStep": 1, "LossStep": 1}
    from vlfb.engine import ConvStep
    convs = {s.out.name: s for s in eng.steps if isinstance(s, ConvStep)}
    blk = convs["res3_1_branch2c_bn"]
    assert blk.relu and blk.residual is not None and blk.sname == "res3_1_branch2c_bn_s"
    nl = convs["nonlocal_conv4_1_sum"]       # out conv + affine + residual add, no ReLU
    assert not nl.relu and nl.residual is not None and nl.cbname == "nonlocal_conv4_1_out_b"
    stem = convs["res_conv1_bn"]
    assert stem.stem and stem.relu and eng.kernel_shape("conv1_w") == (64, 5, 7, 8, 4)
    # the grouped non-local block's Transpose/Reshape/Transpose are views: batch of 8 groups of 4 frames
    att = [s for s in eng.steps if type(s).__name__ == "AttentionStep"][0]
    assert att.theta.shape == (2 * 4, 256, 4 * 28 * 28) and att.phi.shape[2] == 4 * 14 * 14


def test_ava_lfb_graph_and_gradient_routing():
    cfg, m, eng = plan("ava_r50_lfb_nl")
    k = kinds(eng)
    assert k["RoiAlignMaxStep"] == 1 and k["ConcatStep"] == 1 and k["LayerNormStep"] == 2 and k["DropoutStep"] == 5
    single = [s for s in eng.steps if type(s).__name__ == "AttentionStep" and s.theta.shape[2] == 1]
    assert len(single) == 2 and single[0].phi.shape == (5, 512, 300)
    assert len(eng.train_order) == 115 and eng.train_order[0] in ("pred_w", "pred_b")
    assert eng.train_order[-1] == "conv1_w"       # backward completes at the stem
    # every root blob that needs a gradient has a contributor count and a buffer
    for b in eng.all_blobs:
        if b.root is b and b.slot.expected:
            assert b.slot.buf is not None
    x = eng.env["res2_0_branch2c_bn"]             # block output: read by next 2a conv + identity shortcut
    assert x.root.slot.expected == 2 and x.root.relu
    assert eng.env["data_train"].detached and not eng.env["lfb_train"].needs_grad
    # solver: one weight-decay range over the whole flat bucket (no trainable '_bn' parameter: the
    # affine pairs are frozen), i.e. a single fused launch
    assert len(eng.wd_ranges) == 1 and eng.wd_ranges[0][0] == 0 and eng.wd_ranges[0][1] == eng.flat_param.numel()
    assert eng.wd_ranges[0][2] == cfg.SOLVER.WEIGHT_DECAY


def test_frozen_backbone_runs_backward_only_through_the_head():
    cfg, m, eng = plan("charades_r50_lfb_nl")
    assert cfg.MODEL.FREEZE_BACKBONE and len(eng.trainable) == 22
    assert len(eng.bwd_steps) < 30
    assert not any(s.out.name.startswith(("res2", "res3", "res4", "nonlocal_conv", "res_conv1")) for s in eng.bwd_steps
                   if hasattr(s, "out"))
    assert not eng.env["res5_2_branch2c_bn"].root.slot.expected


def test_r101_three_layer_plan():
    cfg, m, eng = plan("ava_r101_lfb_nl_3l")
    assert kinds(eng)["ConvStep"] == 138 and len(eng.trainable) == 174
    assert sum(1 for s in eng.steps if type(s).__name__ == "AttentionStep") == 8   # 2 + 3 backbone, 3 FBO layers


def test_test_split_and_lfb_inference_plans():
    ov = ("NUM_GPUS", 1, "TEST.BATCH_SIZE", 2)
    cfg, m, eng = plan("ava_r50_lfb_nl", ov, split="test")
    assert not eng.train and kinds(eng)["DropoutStep"] == 0 and "prob" in eng.env
    cfg, m, eng = plan("ava_r50_lfb_nl", ov, split="test", lfb_infer_only=True)
    assert "box_pooled" in eng.env and "pred" not in eng.env and "lfb_test" not in m.input_blob_names


def test_grouped_convolution_is_a_batch_of_channel_slice_launches():
    """RESNETS.NUM_GROUPS > 1 (resnet_helper.py:56-63 forwards `group` to the 1x3x3 conv of every bottleneck; 1 in every
    shipped config): the weight is (Cout, Cin / G, 1, 3, 3) as in Caffe2, the step launches the ordinary kernels once per
    group on channel slices (leading dimensions = the tensors' row strides), one weight-operand block per group"""
    from vlfb.engine import ConvStep
    ov = ("NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 2, "RESNETS.NUM_GROUPS", 2, "RESNETS.WIDTH_PER_GROUP", 32)
    cfg, m, eng = plan("charades_r50_baseline", overrides=ov, dtype="mix")
    assert m.param_init_net.fills["res3_1_branch2b_w"].shape == (128, 64, 1, 3, 3)
    grouped = [s for s in eng.steps if isinstance(s, ConvStep) and s.group > 1]
    assert len(grouped) == 16 and all(s.out.name.endswith("_branch2b_bn") for s in grouped)
    c = [s for s in grouped if s.out.name == "res3_1_branch2b_bn"][0]
    assert (c.d_f.Cs, c.d_f.Cn, c.d_f.lda, c.d_f.ldo) == (64, 64, 128, 128)
    assert (c.d_d.Cs, c.d_d.Cn, c.d_d.lda, c.d_d.ldo, c.d_d.ldr) == (64, 64, 128, 128, 128)
    assert (c.d_w.Cs, c.d_w.Cn, c.d_w.lda, c.d_w.ldp) == (64, 64, 128, 128)
    assert c.wblk == 64 * 9 * 64 and c.w_f.numel() == 3 * 2 * c.wblk and c.w_d.numel() == 2 * 2 * c.wblk
    assert len(eng.trainable) == 95          # same parameter catalogue, half the 3x3 weights


def test_unsupported_graphs_fail_loudly():
    """what the kernels cannot run is refused when the engine plans, with the library's own message -- never a fallback:
    ResNeXt-32x4d has 4 channels per group in res2, and a 16-byte operand chunk is 8 (16-bit) channels"""
    from vlfb import hip
    with pytest.raises(hip.VlfbError, match="Cs=4"):
        plan("charades_r50_baseline", overrides=("NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 2, "RESNETS.NUM_GROUPS", 32,
                                                 "RESNETS.WIDTH_PER_GROUP", 4))
    with pytest.raises(ValueError):       # a group count that does not divide the channels (not reachable from the yaml keys)
        from models.model_builder_video import ModelBuilder
        ModelBuilder(train=True, split="train", name="t").ConvNd("x", "y", 64, 64, [1, 3, 3], group=3, no_bias=True)


def test_spatial_bn_graph_lowers_to_bn_steps():
    """MODEL.USE_AFFINE False / NONLOCAL.USE_BN True (Conv3dBN, model_builder_video.py:176-197): one BNStep per conv of the
    backbone and per non-local block, scale / bias trainable, running statistics computed parameters outside the solver's
    bucket; res3 non-local blocks are not grouped in this graph (resnet_video.py:248-272)"""
    ov = ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 2, "TRAIN.VIDEO_LENGTH", 8, "TRAIN.CROP_SIZE", 64,
          "MODEL.USE_AFFINE", False, "NONLOCAL.USE_BN", True, "NONLOCAL.USE_AFFINE", False]
    # the reference's Conv3dBN swallows `dilations=` (model_builder_video.py:176-183) while resnet_helper.py:57 still pads
    # for the dilated kernel: with cfg.DILATIONS = 2 res5_0_branch2b comes out 2 pixels larger per side and the residual
    # Sum cannot be formed -- Caffe2 fails when the net runs, the engine when it plans
    with pytest.raises(ValueError, match="Sum.res5_0_branch2c_bn, res5_0_branch1_bn.: operand shapes differ"):
        plan("ava_r50_lfb_nl", ov)
    ov += ["MODEL.DILATIONS_AFTER_CONV5", False]
    cfg, m, eng = plan("ava_r50_lfb_nl", ov)
    k = kinds(eng)
    assert k["BNStep"] == 1 + 16 * 3 + 4 + 5
    assert len(m.computed_params) == 2 * k["BNStep"] and not set(m.computed_params) & set(m.params)
    assert all(n in eng.trainable for n in ("res_conv1_bn_s", "res_conv1_bn_b", "nonlocal_conv4_1_bn_s"))
    assert "res_conv1_bn_rm" in eng.frozen_layout and "res_conv1_bn_riv" not in eng.train_layout
    assert m.param_init_net.fills["nonlocal_conv3_1_bn_s"].kwargs["value"] == cfg.NONLOCAL.BN_INIT_GAMMA
    cfg, m, eng = plan("ava_r50_lfb_nl", ov, split="test")
    assert all(st.is_test for st in eng.steps if type(st).__name__ == "BNStep")


def test_every_shipped_config_plans_in_train_and_test_mode():
    """all 26 presets (= the reference's configs/*.yaml, tests/test_ref_graph.py), train and test graph: the lowering
    knows every operator sequence they emit, the backbone fuses to one launch per convolution (R50: 73 + head convs,
    R101: 124 + head convs), FBO-NL heads get one attention step per layer, avg / max heads one extra pool + a concat"""
    from vlfb.presets import PRESETS
    small = ("NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 2, "TEST.BATCH_SIZE", 2, "TRAIN.VIDEO_LENGTH", 8, "TEST.VIDEO_LENGTH", 8,
             "TRAIN.CROP_SIZE", 64, "TEST.CROP_SIZE", 64)
    assert len(PRESETS) == 26
    for name in sorted(PRESETS):
        for split in ("train", "test"):
            cfg, m, eng = plan(name, small, split=split)
            k = kinds(eng)
            backbone = 73 if cfg.MODEL.DEPTH == 50 else 124
            layers = cfg.FBO_NL.NUM_LAYERS if (cfg.LFB.ENABLED and cfg.LFB.FBO_TYPE == "nl") else 0
            head_convs = (2 + 4 * layers) if layers else 0          # input reduction + lfb 1x1, theta / phi / g / out per layer
            assert k["ConvStep"] == backbone + head_convs, (name, split, dict(k))
            assert k["AttentionStep"] == 5 + layers, (name, split)
            assert k.get("ConcatStep", 0) == int(bool(cfg.LFB.ENABLED)), (name, split)
            assert k["PoolStep"] == 8 + int(cfg.LFB.ENABLED and cfg.LFB.FBO_TYPE in ("avg", "max")), (name, split)
            assert k.get("RoiAlignMaxStep", 0) == int(cfg.DATASET == "ava") and k["LossStep"] == 1
            assert (k.get("DropoutStep", 0) > 0) == (split == "train")
            if split == "train":
                frozen = bool(cfg.MODEL.FREEZE_BACKBONE)
                assert ("conv1_w" in eng.trainable) == (not frozen), (name, frozen)


def test_mix_keeps_fp32_gradients_on_the_direct_path_of_the_head():
    """Engine._plan_head_f32 (validated on the GPU at full size in round 5 and made part of the `mix` dtype: no switch): the
    gradient slots of the head's direct path -- classifier input, dropout input, the RoI features, the pooled res5 map --
    are fp32, and so are the 22 blobs of the FBO branch (Engine._plan_fbo_f32: its convs run the split-bf16 backward between
    fp32 slots); the backbone is untouched"""
    from vlfb.engine import Engine
    small = ("NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 2, "TRAIN.VIDEO_LENGTH", 8, "TRAIN.CROP_SIZE", 64)
    assert not hasattr(Engine, "MIX_HEAD_F32") and not hasattr(Engine, "MIX_W2_SKIP")     # no default-off parity switches
    import torch
    cfg, m, eng = plan("ava_r50_lfb_nl", small, dtype="mix")
    assert eng.head_f32 == ["pool5_dropout", "pool5", "roi_feat_1d", "blob_pooled"]
    f32 = sorted(b.name for b in eng.all_blobs if b.root is b and b.grad_f32)
    nl_f32 = [n for n in f32 if n.startswith("nonlocal_") and n.rsplit("_", 1)[-1] in ("theta", "phi", "g", "y")]
    fbo = eng.head_f32_fbo                              # the FBO branch between the concat and box_pooled: all of it, in fp32
    assert len(fbo) == 22 and all(n.startswith(("lfb_", "box_pooled_fbonl_")) for n in fbo)
    assert f32 == sorted(nl_f32 + eng.head_f32 + fbo)
    from vlfb.engine import ConvStep
    from vlfb import hip
    convs = {s.wname: s for s in eng.steps if isinstance(s, ConvStep)}
    th = convs["lfb_nl0_theta_w"]                       # a conv between fp32 slots: the `split` backward, bf16 term planes of W
    assert th.bwd_split and not th.w2 and th.wcode == hip.SPLIT and th.w_d.dtype == torch.bfloat16
    assert (th.d_d.dtype, th.d_d.out_dtype, th.d_d.math) == (hip.F32, hip.F32, hip.MATH_BF16X3)
    assert (th.d_w.dtype, th.d_w.math) == (hip.F32, hip.MATH_BF16X3)
    bank = convs["lfb_1x1_w"]                           # reads the fed bank: no DGRAD, the split WGRAD on fp32 operands
    assert bank.d_d is None and bank.bwd_f32 and not bank.bwd_split
    assert not convs["res5_2_branch2c_w"].bwd_split and convs["res5_2_branch2c_w"].w2
    for b in eng.all_blobs:
        if b.root is b and b.name in eng.head_f32:
            assert b.slot.buf.dtype == torch.float32 and not b.slot.two_term
    cfg, m, eng = plan("charades_r50_baseline", small, dtype="mix")
    assert eng.head_f32 == ["pool5_dropout", "res5_2_branch2c_bn_pooled"]
    for dtype in ("fp16", "split"):                       # part of `mix`: nothing happens on the other paths
        cfg, m, eng = plan("ava_r50_lfb_nl", small, dtype=dtype)
        assert eng.head_f32 == []


def test_mix_falls_back_to_plain_dgrad_weights_where_the_doubled_tap_form_does_not_plan(monkeypatch):
    """ConvStep._w2_geometry: a conv whose two-term DGRAD launch (doubled outermost tap dimension) the library refuses keeps a
    plain fp16 DGRAD copy instead of failing the graph; every conv of the shipped widths is two-term.  (The refusal is
    injected: the shipped widths all plan.)"""
    from vlfb import hip
    from vlfb.engine import ConvStep
    small = ("NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 2, "TRAIN.VIDEO_LENGTH", 8, "TRAIN.CROP_SIZE", 64)
    cfg, m, eng = plan("charades_r50_baseline", small, dtype="mix")
    convs = [s for s in eng.steps if isinstance(s, ConvStep) and s.d_d is not None]
    # (hip.MIXH*: the same DGRAD copies next to a two-plane fp16 FPROP copy -- the convs whose input is stored as two planes)
    def code_of(s):    # (w2i: the two terms interleaved per 64-channel k-tile, hip.MATH_F16W2 -- ConvStep._w2_interleaved)
        return {(False, False): hip.MIX_W2, (True, False): hip.MIXH_W2, (False, True): hip.MIX_W2I, (True, True): hip.MIXH_W2I}[(s.x_pair, s.w2i)]
    assert convs and all(s.w2 and s.wcode == code_of(s) for s in convs)
    # the unit-stride convs that the 128-row kernel runs share one gradient tile between the two weight terms; the strided
    # convs keep the doubled-tap form (class walk / in-place class-0 accumulate)
    assert all((s.d_d.math == hip.MATH_F16W2 and s.d_d.kt == s.k[0] and s.d_d.dt == s.d[0]) if s.w2i else
               (s.d_d.math == hip.MATH_NATIVE and s.d_d.kt == 2 and s.d_d.dt == 0) for s in convs)
    assert all(not s.w2i for s in convs if tuple(s.s) != (1, 1, 1)) and sum(s.w2i for s in convs) > len(convs) // 2
    real = hip.conv_workspace_bytes

    def refuse_some(d):
        if d.mode == hip.DGRAD and d.kt == 2 and d.dt == 0 and d.Cn == 64:
            raise hip.VlfbError("injected: the doubled-tap form of this conv does not plan")
        return real(d)
    monkeypatch.setattr(hip, "conv_workspace_bytes", refuse_some)
    cfg, m, eng = plan("charades_r50_baseline", small, dtype="mix")
    convs = [s for s in eng.steps if isinstance(s, ConvStep) and s.d_d is not None]
    plain = [s for s in convs if not s.w2]
    assert plain and all(s.d_d.Cn == 64 and s.wcode == (hip.MIXH if s.x_pair else hip.MIX) and s.d_d.kt == s.k[0] and s.wd_npl == 1 for s in plain)
    assert all(s.wcode == code_of(s) and s.wd_npl == 2 for s in convs if s.w2) and any(s.w2 for s in convs)
    assert {s.wcode for s in convs} <= {hip.MIX, hip.MIX_W2, hip.MIXH, hip.MIXH_W2, hip.MIX_W2I, hip.MIXH_W2I} and len({s.wcode for s in convs}) >= 2    # (one weight-prep batch per format: Engine._wprep_table)


def test_bank_side_of_the_fbo_head_is_issued_first_on_the_second_stream():
    """Engine._plan_forward_branches rule (c): the forward steps that do not depend on the clip -- lfb_1x1, its dropout, the
    phi / g convs of the FBO blocks -- are issued at the start of forward() on the second stream; their consumers on the main
    stream (the FBO attention steps) wait for an event; nothing that reads the clip is among them"""
    from vlfb.engine import ConvStep, DropoutStep, AttentionStep
    cfg, m, eng = plan("ava_r50_lfb_nl", dtype="mix")
    early = [eng.steps[i] for i in eng._fwd_early]
    assert [st.outputs[0].name for st in early] == ["lfb_1x1", "lfb_1x1_drop", "lfb_nl0_phi", "lfb_nl0_g", "lfb_nl1_phi", "lfb_nl1_g"]
    assert all(isinstance(st, (ConvStep, DropoutStep)) for st in early) and set(eng._fwd_early) <= eng._fwd_side
    for i, st in enumerate(eng.steps):
        if isinstance(st, AttentionStep) and st.outputs[0].name.startswith("lfb_nl"):
            assert set(eng._fwd_wait[i]) & set(eng._fwd_early), st.outputs[0].name      # (phi / g arrive from the other stream)
    cfg, m, eng = plan("charades_r50_baseline", dtype="mix")                         # no bank: nothing to hoist
    assert eng._fwd_early == []


def test_strided_projection_shortcuts_run_their_dgrad_as_an_in_place_accumulate():
    """Engine._plan_sparse_shortcut_dgrads (16-bit backward: fp16, bf16, mix): the DGRADs of res3_0 / res4_0 branch1 -- 1x1x1,
    stride (1, 2, 2) -- are planned as hip.ALGO_CLASS0 (only the rows the conv reads) and their backward step comes BEHIND
    branch2a's, the other contributor to the block-input gradient, so that they accumulate in place; the stride-1 shortcuts of
    res2_0 / res5_0 and the fp32-storage backward of `split` are untouched"""
    from vlfb import hip
    from vlfb.engine import ConvStep
    for dtype in ("mix", "fp16"):
        cfg, m, eng = plan("ava_r50_lfb_nl", dtype=dtype)
        convs = {s.wname: s for s in eng.steps if isinstance(s, ConvStep)}
        order = [s.wname for s in eng.bwd_steps if isinstance(s, ConvStep)]
        for blk in ("res3_0", "res4_0"):
            sc = convs[blk + "_branch1_w"]
            assert sc.sparse_dgrad and sc.d_d.algo == hip.ALGO_CLASS0 and sc.d_d_full.algo == hip.ALGO_AUTO, blk
            assert "class0" in hip.conv_plan(sc.d_d) and hip.conv_flops(sc.d_d) == hip.conv_flops(sc.d_d_full)
            assert order.index(blk + "_branch1_w") == order.index(blk + "_branch2a_w") + 1, order
            assert order.index(blk + "_branch2c_w") < order.index(blk + "_branch2a_w")
        for blk in ("res2_0", "res5_0"):
            assert not getattr(convs[blk + "_branch1_w"], "sparse_dgrad", False) and convs[blk + "_branch1_w"].d_d_full is None
    cfg, m, eng = plan("ava_r50_lfb_nl", dtype="split")
    assert not any(getattr(s, "sparse_dgrad", False) for s in eng.steps)


def test_product_code_never_imports_the_oracle():
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "video-long-term-feature-banks_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, re.M), os.path.join(dirpath, f)


def test_mix_plan_joins_a_split_forward_to_an_fp16_backward():
    """Engine dtype "mix" (DESIGN.md 3.5): the trunk's activations are two fp16 planes read by two-plane forward convs (three
    fp16 MFMAs per product), the rest of the forward is split-bf16 on fp32 storage; every forward value the backward reads
    has an fp16 copy with exactly one writer (the hi plane of a two-plane blob, the producing conv's epilogue, a copy pass
    behind a non-conv step, or the start of forward() for fed blobs); the backward descriptors are fp16 with two-term weights;
    theta / phi / g of the non-local blocks keep fp32 gradients; the residual-stream slots are two-term"""
    import torch
    from vlfb import hip
    from vlfb.engine import ConvStep, AttentionStep, Engine
    cfg, m, eng = plan("ava_r50_lfb_nl", dtype="mix")
    assert eng.mix and eng.split and eng.tdtype == torch.float32 and eng.btdtype == torch.float16
    assert eng.math_fwd == hip.MATH_BF16X3 and eng.math_bwd == hip.MATH_NATIVE and eng.wcode == hip.MIX_W2
    halves = [b for b in eng.all_blobs if b.root is b and b.half is not None]
    by_conv = {id(o.root) for st in eng.steps if isinstance(st, ConvStep) for o in st.outputs}
    posted = [id(b) for st in eng.steps for b in st._half_post]
    fed = [id(b) for b in eng._half_inputs]
    assert len(posted) == len(set(posted)) and not (set(posted) & by_conv) and not (set(fed) & by_conv)
    for b in halves:
        # (a two-plane blob: the hi plane IS the copy -- written by its producer, a conv epilogue or the max pool over planes)
        assert (id(b) in by_conv or b.pair) + (id(b) in posted) + (id(b) in fed) == 1, b.name
        assert b.half.dtype == torch.float16 and b.half.numel() * (2 if b.pair else 1) == b.tensor.numel()
        assert not b.pair or (b.tensor.dtype == torch.float16 and b.half.data_ptr() == b.tensor.data_ptr())
    # the two-plane blobs: every bottleneck / stem / pool output of the trunk and the attention output of every non-local block
    # (its P . g product writes two planes, so the `out` conv is a two-plane conv); theta / phi / g, the probabilities and the
    # head stay fp32
    pairs = {b.name for b in eng.all_blobs if b.root is b and b.pair}
    assert len(pairs) == 80 and {"res_conv1_bn", "pool1", "res2_0_branch2a_bn", "res3_1_branch2c_bn", "nonlocal_conv4_1_pool",
                                 "nonlocal_conv4_1_theta", "nonlocal_conv4_1_phi", "nonlocal_conv4_1_y", "nonlocal_conv4_1_sum",
                                 "res5_2_branch2c_bn"} <= pairs        # (theta / phi: the scores are a batched two-plane product)
    assert not any(n.endswith(("_g", "_prob")) or n.startswith(("lfb", "box_pooled", "blob_pooled", "pool5")) for n in pairs)
    assert sorted(b.name for b in eng._half_inputs) == ["data_train"]     # (the bank is read by fp32 steps only: the FBO head)
    convs = {s.out.name: s for s in eng.steps if isinstance(s, ConvStep)}
    c = convs["res4_1_branch2b_bn"]                       # 1x3x3, unit stride: the two weight terms interleaved per k-tile (MATH_F16W2)
    assert (c.d_f.dtype, c.d_f.out_dtype, c.d_f.math, c.d_f.kt) == (hip.F16, hip.F16, hip.MATH_F16X3, 1) and c.x_pair and c.o_pair
    assert c.d_f.a_pstride == c.x.numel and abs(c.d_f.alpha * hip.MIX_W2_SCALE - 1.0) < 1e-6 and c.wcode == hip.MIXH_W2I and c.w2 and c.w2i
    assert c.w_f.dtype == torch.float16 and c.w_f.shape[0] == 2 and hip.conv_plan(c.d_f).startswith("nt8_pair f16x3")         # (K = 2304: the 256-row pipelined form)
    assert (c.d_d.dtype, c.d_d.math, c.d_d.kt, c.d_d.dt, c.d_d.kh, c.d_d.kw) == (hip.F16, hip.MATH_F16W2, 1, 1, 3, 3)
    assert hip.conv_plan(c.d_d) == "nt f16 128x128 ut w2" and hip.conv_family(c.d_d) == ("nt_16", 2)
    assert abs(c.d_d.alpha * hip.MIX_W2_SCALE - 1.0) < 1e-6 and c.w_d.numel() == c.w_f.numel()
    assert hip.conv_flops(c.d_d) == hip.conv_flops(c.d_f)          # the second term is not algorithmic work
    assert (c.d_w.dtype, c.d_w.out_dtype, c.d_w.math) == (hip.F16, hip.F32, hip.MATH_NATIVE)
    sc = convs["res4_0_branch2b_bn"]                      # 1x3x3, stride (1, 2, 2): the class walk on a doubled kt of dilation 0
    assert (sc.d_d.dtype, sc.d_d.math, sc.d_d.kt, sc.d_d.dt, sc.d_d.kh, sc.d_d.kw) == (hip.F16, hip.MATH_NATIVE, 2, 0, 3, 3)
    assert sc.w2 and not sc.w2i and sc.wcode == hip.MIXH_W2 and hip.conv_flops(sc.d_d) == hip.conv_flops(sc.d_f)
    a = convs["res4_0_branch2a_bn"]                       # 3x1x1 that the library gives to the 256-row kernel: the doubled-tap form,
    N, _, T, H, W = a.x.shape                             # T plays H, H x W one pointwise axis, T' = 1 carries the terms
    assert not a.w2i and hip.conv_plan(a.d_d).startswith("nt8 ")
    assert (a.d_d.kt, a.d_d.dt, a.d_d.kh, a.d_d.kw, a.d_d.ph) == (2, 0, 3, 1, 1)
    assert (a.d_d.Tr, a.d_d.Hr, a.d_d.Wr, a.d_d.Ts, a.d_d.Hs, a.d_d.Ws) == (1, T, H * W, 1, T, H * W)
    a2 = convs["res4_2_branch2a_bn"]                      # the same conv shape on the 128-row kernel: its own 3-D geometry
    assert a2.w2i and (a2.d_d.math, a2.d_d.kt, a2.d_d.dt, a2.d_d.kh, a2.d_d.pt) == (hip.MATH_F16W2, 3, 1, 1, 1)
    # fp32 gradients and split products around the non-local softmax
    f32 = sorted(b.name for b in eng.all_blobs if b.root is b and b.grad_f32 and b.name not in eng.head_f32 + eng.head_f32_fbo)
    assert len(f32) == 20 and all(n.rsplit("_", 1)[1] in ("theta", "phi", "g", "y") for n in f32), f32
    oc = [c for c in convs.values() if c.wname == "nonlocal_conv4_1_out_w"][0]       # (fused with its AffineNd + Sum: named by the sum)
    assert oc.dx_f32 and oc.d_d.out_dtype == hip.F32 and oc.x.root.slot.buf.dtype == torch.float32
    # (its forward: the two-plane attention output in -- the P . g product of the block is a split-bf16 launch with a two-plane
    # output --, the two-plane block input added, a two-plane output)
    assert (oc.d_f.dtype, oc.d_f.out_dtype, oc.d_f.math) == (hip.F16, hip.F16, hip.MATH_F16X3) and oc.o_pair and oc.x_pair
    att4 = [s for s in eng.steps if isinstance(s, AttentionStep) and s.out.name == "nonlocal_conv4_1_y"][0]
    assert att4.o_pair and (att4.d_y.dtype, att4.d_y.out_dtype, att4.d_y.math) == (hip.F32, hip.F16, hip.MATH_BF16X3)
    th = convs["nonlocal_conv4_1_theta"]
    assert th.x_pair and th.o_pair and (th.d_f.dtype, th.d_f.out_dtype, th.d_f.math) == (hip.F16, hip.F16, hip.MATH_F16X3)
    gg = convs["nonlocal_conv4_1_g"]                      # g stays fp32 (the backward's dP product splits its fp32 values)
    assert gg.x_pair and not gg.o_pair and (gg.d_f.dtype, gg.d_f.out_dtype, gg.d_f.math) == (hip.F16, hip.F32, hip.MATH_F16X3)
    assert att4.s_pair and (att4.d_s.dtype, att4.d_s.out_dtype, att4.d_s.math, att4.d_s.batch) == (hip.F16, hip.F32, hip.MATH_F16X3, att4.B)
    assert th.bwd_f32 and (th.d_w.dtype, th.d_w.math, th.d_w.wgrad_bias) == (hip.F32, hip.MATH_BF16X3, 1)
    # the two fp32 gradients of a block that a 16-bit launch reads next -- the attention output (into the dP / dg products) and
    # theta (into the theta conv's DGRAD) -- get their fp16 rounding from the launch that produces them (GradSlot.half_buf)
    halves_g = sorted(b.name for b in eng.all_blobs if b.root is b and b.slot.half_buf is not None)
    assert halves_g == sorted("nonlocal_conv%s_%s" % (blk, n) for blk in ("3_1", "3_3", "4_1", "4_3", "4_5") for n in ("theta", "y"))
    assert all(b.slot.half_buf.dtype == torch.float16 and b.slot.half_buf.numel() == b.numel for b in eng.all_blobs if b.root is b and b.slot.half_buf is not None)
    assert th.out.root.slot.buf.dtype == torch.float32
    att = [s for s in eng.steps if isinstance(s, AttentionStep) and not s.single][0]
    assert att.precise and not att.fused_bwd and (att.d_dp.dtype, att.d_dp.math) == (hip.F32, hip.MATH_BF16X3)
    # the residual stream: the identity operands of the residual Sums (17), and every other 16-bit slot that is the sum of
    # several conv DGRADs -- the inputs of the four projection blocks (pool1, pool2, the last blobs of res3 / res4), the pooled
    # maps in front of phi / g (5); the FBO head keeps fp32 slots
    two = [b.name for b in eng.all_blobs if b.root is b and b.slot.two_term]
    assert len(two) == 27 and "res2_0_branch2c_bn" in two and "nonlocal_conv4_3_sum" in two
    # ... and the last blob of res5: the fp32 pooled gradient of the head re-enters the 16-bit backward as hi + lo
    assert "res5_2_branch2c_bn" in two and [s for s in eng.steps if s.name() == "avgpool:blob_pooled"][0].two_term_dx
    assert all(n in two for n in ("pool1", "pool2", "nonlocal_conv3_3_sum", "nonlocal_conv4_5_sum", "nonlocal_conv4_1_pool"))
    assert all(eng.env[n].root.slot.buf_lo is not None and eng.env[n].root.slot.buf.dtype == torch.float16 for n in two)
    # every launch has a plan; the table is a pure function of the descriptors (what bench.py and the plan test compare)
    table = eng.plan_table()
    assert len(table) > 250 and all(r[3] for r in table)
    assert Engine(m, "mix", dry_run=True).plan(collections.OrderedDict((b.name, b.shape) for b in eng.all_blobs
                                                if getattr(b, "is_input", False) and b.root is b)).plan_table() == table


def test_other_dtypes_plan_without_half_copies():
    for dtype in ("bf16", "fp16", "split", "fp32"):
        cfg, m, eng = plan("charades_r50_baseline", dtype=dtype)
        assert not eng.mix and all(b.half is None for b in eng.all_blobs)
        assert not eng._half_inputs and not any(st._half_post for st in eng.steps)
        assert eng.bcode == eng.code and not any(b.slot.two_term for b in eng.all_blobs if b.root is b)


def test_dot_product_nonlocal_variant_lowers_to_the_same_attention_step():
    """NONLOCAL.USE_SOFTMAX False (reference nonlocal_helper.py:107-119): BatchMatMul -> ConstantFill(1) -> ReduceBackSum ->
    ConstantFill(0) -> Add(broadcast) -> StopGradient -> Div -> BatchMatMul is recorded operator by operator and lowers
    to ONE attention step whose scores product carries 1 / L2 (no Scale, no Softmax)"""
    from vlfb.engine import AttentionStep
    cfg, m, eng = plan("charades_r50_baseline", overrides=("NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 2, "NONLOCAL.USE_SOFTMAX", False))
    types = [o.type for o in m.net.ops]
    assert types.count("ReduceBackSum") == 5 and types.count("Div") == 5 and "Softmax" not in types and "Scale" not in types
    att = [s for s in eng.steps if isinstance(s, AttentionStep)]
    assert len(att) == 5 and len(eng.steps) == 89 and all(a.dot and not a.fused_fwd and not a.fused_bwd for a in att)
    a = att[-1]
    assert a.prob.name == "nonlocal_conv4_5_affinity_sc" and abs(a.d_s.alpha * a.L2 - 1.0) < 1e-6
    assert abs(a.d_dp.alpha * a.L2 / a.ds_scale - 1.0) < 1e-6


def test_two_plane_storage_stops_at_the_fp32_head_with_and_without_dropout():
    """Engine._plan_pairs: the FBO head's convs run the split-bf16 backward on fp32 VALUES (ConvStep.bwd_split), so their
    blobs keep fp32 storage -- also when the dropouts between them are configured away and `lfb_1x1` feeds the phi / g convs
    directly (the benchmark-plan test runs that graph; a conv output with an fp32 gradient slot may only become two-plane as
    theta / phi of a space-time non-local block)."""
    from vlfb.engine import ConvStep
    for ov in ((), ("TRAIN.DROPOUT_RATE", 0.0, "FBO_NL.INPUT_DROPOUT_ON", False, "FBO_NL.LFB_DROPOUT_ON", False)):
        cfg, m, eng = plan("ava_r50_lfb_nl", ov, dtype="mix")
        assert len(eng.pair_blobs) == 80 and not [n for n in eng.pair_blobs if n.startswith(("lfb", "box_pooled", "pool5"))]
        for st in eng.steps:
            if isinstance(st, ConvStep) and st.bwd_split:
                assert not st.x_pair and not st.o_pair, st.out.name
