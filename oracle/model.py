"""CPU restatement of the reference graph: R50/R101-I3D-NL backbone, RoI / basic head, FBO heads,
classifier and loss -- forward in torch-CPU functional ops (NCTHW, the reference's layout),
backward by autograd.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py; parity unpinned by the
reference).

Each function cites the reference builder it follows.  Operator arithmetic (Conv, MaxPool,
BatchMatMul, Softmax, LayerNorm, RoIAlign, SigmoidCrossEntropyLoss, ...) is Caffe2's, restated
per SURVEY.md Appendix B.  Independent of the product's builder/engine on purpose: it is written
straight from the reference files, takes a plain `cfg` object and a `{blob name: tensor}` dict.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import rng as orng
from .roi_align import roi_align_torch

BLOCK_CONFIG = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}  # lib/models/resnet_video.py:33-36


def temporal_arc(cfg):
    """use_temp_convs per stage (lib/models/resnet_video.py:39-130); temp strides are all 1."""
    choice, depth = cfg.MODEL.VIDEO_ARC_CHOICE, cfg.MODEL.DEPTH
    n3 = BLOCK_CONFIG[depth][2]
    if choice in (1, 3):  # C2D
        return [[0], [0, 0, 0], [0, 0, 0, 0], [0] * n3, [0, 0, 0]]
    if choice in (2, 4):  # I3D
        return [[2], [1, 1, 1], [1, 0, 1, 0], [1 if i % 2 == 0 else 0 for i in range(n3)], [0, 1, 0]]
    raise ValueError("unknown VIDEO_ARC_CHOICE")


def batch_per_gpu(cfg, split):
    """lib/utils/misc.py:68-72"""
    total = cfg.TEST.BATCH_SIZE if split in ("test", "val") else cfg.TRAIN.BATCH_SIZE
    return int(total / cfg.NUM_GPUS)


# --------------------------------------------------------------------------------------------
# parameter catalogue (names/shapes as Caffe2's ConvNd / FC / ModelBuilder.AffineNd create them,
# SURVEY.md Appendix C) in the order the builders emit them
# --------------------------------------------------------------------------------------------
def param_spec(cfg, lfb_infer_only=False):
    """OrderedDict name -> dict(shape, kind, fan_out, std, trainable)."""
    spec = OrderedDict()

    def conv(name, cout, cin, k, bias, kind, std=None, groups=1):
        spec[name + "_w"] = dict(shape=(cout, cin // groups) + tuple(k), kind=kind, std=std, trainable=True)
        if bias:
            spec[name + "_b"] = dict(shape=(cout,), kind="zero_bias", trainable=True)

    def affine(name, c):
        spec[name + "_s"] = dict(shape=(c,), kind="affine_s", trainable=False)
        spec[name + "_b"] = dict(shape=(c,), kind="affine_b", trainable=False)

    def bn(name, c):
        """CNNModelHelper.SpatialBN: scale / bias trained, running statistics computed (not parameters of the solver)"""
        spec[name + "_s"] = dict(shape=(c,), kind="bn_s", trainable=True)
        spec[name + "_b"] = dict(shape=(c,), kind="bn_b", trainable=True)
        spec[name + "_rm"] = dict(shape=(c,), kind="bn_rm", trainable=False)
        spec[name + "_riv"] = dict(shape=(c,), kind="bn_riv", trainable=False)

    def conv_affine(prefix, cout, cin, k, groups=1):
        """Conv3dAffine, or Conv3dBN when MODEL.USE_AFFINE is off (resnet_helper.py:28-32)"""
        conv(prefix, cout, cin, k, False, "msra", groups=groups)
        (affine if cfg.MODEL.USE_AFFINE else bn)(prefix + "_bn", cout)

    def nonlocal_block(prefix, c, ci):
        std = cfg.NONLOCAL.CONV_INIT_STD
        has_b = not cfg.NONLOCAL.NO_BIAS
        conv(prefix + "_theta", ci, c, (1, 1, 1), has_b, "gauss", std)
        conv(prefix + "_phi", ci, c, (1, 1, 1), has_b, "gauss", std)
        conv(prefix + "_g", ci, c, (1, 1, 1), has_b, "gauss", std)
        conv(prefix + "_out", c, ci, (1, 1, 1), has_b, "nl_out", std)
        assert bool(cfg.NONLOCAL.USE_AFFINE) != bool(cfg.NONLOCAL.USE_BN), "one of NONLOCAL.USE_BN / USE_AFFINE"
        (bn if cfg.NONLOCAL.USE_BN else affine)(prefix + "_bn", c)

    arc = temporal_arc(cfg)
    n1, n2, n3, n4 = BLOCK_CONFIG[cfg.MODEL.DEPTH]
    conv("conv1", 64, 3, (1 + 2 * arc[0][0], 7, 7), False, "msra")
    (affine if cfg.MODEL.USE_AFFINE else bn)("res_conv1_bn", 64)
    w = cfg.RESNETS.NUM_GROUPS * cfg.RESNETS.WIDTH_PER_GROUP
    mod3 = cfg.NONLOCAL.LAYER_MOD
    if cfg.MODEL.DEPTH == 101:
        mod3 = 2
    if not cfg.NONLOCAL.CONV3_NONLOCAL:
        mod3 = 1000
    mod4 = cfg.NONLOCAL.LAYER_MOD
    if cfg.MODEL.DEPTH == 101:
        mod4 = mod4 * 4 - 1
    if not cfg.NONLOCAL.CONV4_NONLOCAL:
        mod4 = 1000
    stages = [("res2", n1, 64, 256, w, 1000, None), ("res3", n2, 256, 512, w * 2, mod3, "nonlocal_conv3"),
              ("res4", n3, 512, 1024, w * 4, mod4, "nonlocal_conv4"), ("res5", n4, 1024, 2048, w * 8, 1000, None)]
    for si, (prefix, nblocks, din, dout, dinner, mod, nlname) in enumerate(stages):
        for i in range(nblocks):
            p = "%s_%d" % (prefix, i)
            utc = arc[si + 1][i]
            conv_affine(p + "_branch2a", dinner, din, (1 + 2 * utc, 1, 1))
            conv_affine(p + "_branch2b", dinner, dinner, (1, 3, 3), groups=cfg.RESNETS.NUM_GROUPS)   # resnet_helper.py:56-63
            conv_affine(p + "_branch2c", dout, dinner, (1, 1, 1))
            if i == 0:
                conv_affine(p + "_branch1", dout, din, (1, 1, 1))
            din = dout
            if i % mod == mod - 1:
                nonlocal_block("%s_%d" % (nlname, i), dout, dout // 2)
    head_dim = 2048
    if cfg.LFB.ENABLED and not lfb_infer_only:
        if cfg.LFB.FBO_TYPE == "nl":
            x_name = "box_pooled" if cfg.DATASET == "ava" else "res5_%d_branch2c_bn_pooled" % (n4 - 1)
            lat = cfg.FBO_NL.LATENT_DIM
            # lfb_helper.init_params1/2 are evaluated at import with DEFAULT cfg (lfb_helper.py:31-40):
            # bias always present, std 0.01, out conv zero-init.
            if cfg.FBO_NL.INPUT_REDUCE_DIM:
                conv(x_name + "_fbonl_reduc", lat, 2048, (1, 1, 1), not cfg.NONLOCAL.NO_BIAS, "gauss", cfg.MODEL.FC_INIT_STD)
                a_dim = lat
            else:
                a_dim = 2048
            conv("lfb_1x1", lat, cfg.LFB.LFB_DIM, (1, 1, 1), not cfg.NONLOCAL.NO_BIAS, "gauss", cfg.MODEL.FC_INIT_STD)
            for l in range(cfg.FBO_NL.NUM_LAYERS):
                pre = "lfb_nl%d" % l
                conv(pre + "_theta", lat, a_dim, (1, 1, 1), True, "gauss", 0.01)
                conv(pre + "_phi", lat, lat, (1, 1, 1), True, "gauss", 0.01)
                conv(pre + "_g", lat, lat, (1, 1, 1), True, "gauss", 0.01)
                conv(pre + "_out", a_dim, lat, (1, 1, 1), True, "fbo_out", 0.01)
            head_dim += a_dim
        else:
            head_dim += cfg.LFB.LFB_DIM
    if not lfb_infer_only:
        spec["pred_w"] = dict(shape=(cfg.MODEL.NUM_CLASSES, head_dim), kind="gauss", std=cfg.MODEL.FC_INIT_STD, trainable=True)
        spec["pred_b"] = dict(shape=(cfg.MODEL.NUM_CLASSES,), kind="zero_bias", trainable=True)
    return spec


def synth_params(cfg, seed=2, lfb_infer_only=False):
    """Synthetic weights per SURVEY.md 8(d): MSRA for backbone convs (fan_out), N(0,std) elsewhere,
    NON-trivial affine (s~U(0.5,1.5), b~N(0,0.1)), non-zero biases and non-zero NL/FBO output convs
    so zero-init paths cannot hide bugs.  Returns {name: float32 ndarray}."""
    gen = np.random.default_rng(seed)
    out = OrderedDict()
    for name, s in param_spec(cfg, lfb_infer_only).items():
        shape, kind = s["shape"], s["kind"]
        if kind == "msra":
            fan_out = shape[0] * int(np.prod(shape[2:]))
            v = gen.standard_normal(shape) * math.sqrt(2.0 / fan_out)
        elif kind in ("gauss", "nl_out", "fbo_out"):
            v = gen.standard_normal(shape) * (s["std"] if kind == "gauss" else 0.01)
            if name.startswith(("nonlocal", "lfb_nl")) or "_fbonl_" in name or name == "lfb_1x1_w":
                v = v * 3.0
            if name.endswith(("_theta_w", "_phi_w")):
                # attention logits with O(1) spread: neither flat nor saturated
                v = v * (2.5 if name.startswith("lfb_nl") else 6.0)
        elif kind == "zero_bias":
            v = gen.standard_normal(shape) * 0.05
        elif kind == "affine_s":
            # the residual-branch exits get a small gain (like BN_INIT_GAMMA ~ 0 after some
            # training) so activations stay O(1) through 16-33 un-normalised residual blocks and
            # the non-local softmaxes do not saturate into hard arg-maxes
            if "_branch2c_bn" in name or name.startswith("nonlocal"):
                v = gen.uniform(0.15, 0.35, shape)
            else:
                v = gen.uniform(0.5, 1.5, shape)
        elif kind in ("affine_b", "bn_b"):
            v = gen.standard_normal(shape) * 0.1
        elif kind == "bn_s":
            # normalised activations: O(1) gains, small ones at the residual-branch exits (BN_INIT_GAMMA = 0 grown a little)
            small = "_branch2c_bn" in name or name.startswith("nonlocal")
            v = gen.uniform(0.15, 0.35, shape) if small else gen.uniform(0.5, 1.5, shape)
        elif kind == "bn_rm":
            v = gen.standard_normal(shape) * 0.1
        elif kind == "bn_riv":
            v = gen.uniform(0.5, 1.5, shape)
        else:
            raise ValueError(kind)
        out[name] = v.astype(np.float32)
    return out


def synth_inputs(cfg, n_clips, split="train", seed=2, rois_per_clip=None, crop=None, frames=None):
    """Synthetic input blobs per SURVEY.md 8(d) C1-C5 (seed = cfg.RNG_SEED).
    Returns dict with data / labels / [proposals] / [lfb] as numpy arrays (reference layouts)."""
    gen = np.random.default_rng(seed)
    crop = crop or (cfg.TRAIN.CROP_SIZE if split == "train" else cfg.TEST.CROP_SIZE)
    frames = frames or cfg.TRAIN.VIDEO_LENGTH
    blobs = OrderedDict()
    data = np.clip(gen.standard_normal((n_clips, 3, frames, crop, crop)), -2.0, 2.45)
    blobs["data"] = data.astype(np.float32)
    ncls = cfg.MODEL.NUM_CLASSES
    if cfg.DATASET == "ava":
        if rois_per_clip is None:
            rois_per_clip = [int(gen.integers(1, 6)) for _ in range(n_clips)]
        rows = []
        for c, k in enumerate(rois_per_clip):
            for _ in range(k):
                x1 = gen.uniform(0, crop - 9); y1 = gen.uniform(0, crop - 9)
                x2 = gen.uniform(x1 + 8, crop - 1); y2 = gen.uniform(y1 + 8, crop - 1)
                rows.append([c, x1, y1, x2, y2])
        blobs["proposals"] = np.asarray(rows, dtype=np.float32)
        R = len(rows)
        blobs["labels"] = (gen.uniform(size=(R, ncls)) < 0.05).astype(np.int32)
        if cfg.LFB.ENABLED:
            steps, per = cfg.LFB.WINDOW_SIZE, cfg.AVA.LFB_MAX_NUM_FEAT_PER_STEP
            lfb = np.zeros((R, steps * per, cfg.LFB.LFB_DIM), dtype=np.float32)
            r0 = 0
            for c, k in enumerate(rois_per_clip):  # one bank per clip, repeated per RoI (ava_data_input.py:191-192)
                bank = np.zeros((steps, per, cfg.LFB.LFB_DIM), dtype=np.float32)
                for s in range(steps):
                    cnt = int(gen.integers(0, per + 1))
                    bank[s, :cnt] = np.maximum(gen.standard_normal((cnt, cfg.LFB.LFB_DIM)), 0) * 0.5
                lfb[r0:r0 + k] = bank.reshape(1, steps * per, -1)
                r0 += k
            blobs["lfb"] = lfb
    else:
        if cfg.MODEL.MULTI_LABEL:
            blobs["labels"] = (gen.uniform(size=(n_clips, ncls)) < 0.05).astype(np.int32)
        else:   # EPIC-Kitchens: one class index per clip (epic.py:135, epic_data_input.py)
            blobs["labels"] = gen.integers(0, ncls, size=(n_clips,)).astype(np.int32)
        if cfg.LFB.ENABLED:
            K = cfg.LFB.WINDOW_SIZE
            lfb = (np.maximum(gen.standard_normal((n_clips, K, cfg.LFB.LFB_DIM)), 0) * 0.5).astype(np.float32)
            for c in range(n_clips):
                pad = int(gen.integers(0, min(10, K - 1) + 1))
                if pad:
                    lfb[c, K - pad:] = 0  # zero padding of short windows (charades.py:269-272)
            blobs["lfb"] = lfb
    return blobs


# --------------------------------------------------------------------------------------------
# graph
# --------------------------------------------------------------------------------------------
class _Ctx(object):
    def __init__(self, cfg, P, split, dropout_seed_fn, blobs_out, decisions=None):
        self.cfg, self.P, self.split = cfg, P, split
        self.test = split in ("test", "val")
        self.seed_fn = dropout_seed_fn
        self.B = blobs_out
        # Discrete decisions handed in by the caller (see `run`): the sign pattern of every ReLU output and the selected
        # window element of every max pool.  The oracle then evaluates THOSE branches of the piecewise-linear network
        # instead of taking its own -- a parity check of the arithmetic that a tie at zero cannot disturb.
        self.dec = decisions
        self.dec_used, self.dec_missing = set(), set()
        # ... and the other direction: when the caller passes decisions["_record"] = True, the decisions THIS evaluation
        # takes are written into decisions["relu"] / ["pool"] / ["roi_bin"] in the same format (oracle/fp32_control.py feeds
        # the fp32 oracle's decisions to the fp64 oracle: the reference-width witness of the parity protocol)
        self.record = bool(decisions and decisions.get("_record"))
        if self.record:
            self.dec = None
            decisions.setdefault("relu", {})
            decisions.setdefault("pool", {})
            self.rec = decisions


def _relu(cx, x, name):
    """ReLU; with caller-supplied decisions: x * [the caller's output was > 0]"""
    m = cx.dec["relu"].get(name) if cx.dec else None
    if m is None:
        if cx.dec:
            cx.dec_missing.add(name)
        y = torch.relu(x)
        if cx.record:
            cx.rec["relu"][name] = (y.detach() > 0).numpy()
            if name in cx.rec.get("_want_pre", ()):          # (diagnostics: the pre-activations of the named sites)
                cx.rec.setdefault("pre", {})[name] = x.detach().numpy().copy()
        return y
    cx.dec_used.add(name)
    return x * torch.from_numpy(np.ascontiguousarray(m)).reshape(x.shape).to(x.dtype)


def _max_pool(cx, x, name, k, s, p=(0, 0, 0)):
    """MaxPool over (T, H, W) windows; with caller-supplied decisions (`tap` = index of the selected element inside its
    window in t, h, w scan order, shape of the output) the output GATHERS those elements"""
    tap = cx.dec["pool"].get(name) if cx.dec else None
    if tap is None:
        if cx.dec:
            cx.dec_missing.add(name)
        if cx.record:
            y, idx = F.max_pool3d(x, k, s, p, return_indices=True)      # idx: flat (t, h, w) position of the selected element
            H, W = x.shape[3], x.shape[4]
            To, Ho, Wo = y.shape[2:]
            ti, hi, wi = idx // (H * W), (idx // W) % H, idx % W
            a = ti - (torch.arange(To).view(1, 1, To, 1, 1) * s[0] - p[0])
            b = hi - (torch.arange(Ho).view(1, 1, 1, Ho, 1) * s[1] - p[1])
            c = wi - (torch.arange(Wo).view(1, 1, 1, 1, Wo) * s[2] - p[2])
            cx.rec["pool"][name] = ((a * k[1] + b) * k[2] + c).numpy()
            return y
        return F.max_pool3d(x, k, s, p)
    cx.dec_used.add(name)
    tap = torch.from_numpy(np.ascontiguousarray(tap)).to(torch.int64)
    N, C, T, H, W = x.shape
    To, Ho, Wo = tap.shape[2:]
    a, rem = tap // (k[1] * k[2]), tap % (k[1] * k[2])
    b, c = rem // k[2], rem % k[2]
    ti = torch.arange(To).view(1, 1, To, 1, 1) * s[0] - p[0] + a
    hi = torch.arange(Ho).view(1, 1, 1, Ho, 1) * s[1] - p[1] + b
    wi = torch.arange(Wo).view(1, 1, 1, 1, Wo) * s[2] - p[2] + c
    assert int(ti.min()) >= 0 and int(ti.max()) < T and int(hi.min()) >= 0 and int(hi.max()) < H and \
        int(wi.min()) >= 0 and int(wi.max()) < W, "max-pool decision of %s selects a padding element" % name
    lin = (ti * H + hi) * W + wi
    return x.flatten(2).gather(2, lin.flatten(2)).reshape(N, C, To, Ho, Wo)


def _affine(x, P, prefix):
    """AffineNd (caffe2_customized_ops/video/affine_nd_op.cu:31-44)"""
    shp = (1, -1) + (1,) * (x.dim() - 2)
    return x * P[prefix + "_s"].view(shp) + P[prefix + "_b"].view(shp)


def _bn(cx, x, prefix, eps, momentum):
    """SpatialBN (Caffe2 spatial_batch_norm_op, a dependency outside the reference tree; call sites
    model_builder_video.py:186-190, resnet_video.py:185-188, nonlocal_helper.py:147-151).  Train nets normalise by the
    biased moments of the batch THIS GPU holds and move the running statistics (unbiased variance), test / val nets use
    the running statistics.  The new running statistics go to cx.B as `<prefix>_rm` / `<prefix>_riv`."""
    P = cx.P
    shp = (1, -1) + (1,) * (x.dim() - 2)
    s, b, rm, rv = (P[prefix + k] for k in ("_s", "_b", "_rm", "_riv"))
    if cx.test:
        return (x - rm.view(shp)) / torch.sqrt(rv.view(shp) + eps) * s.view(shp) + b.view(shp)
    dims = [0] + list(range(2, x.dim()))
    n = x.numel() // x.shape[1]
    mu = x.mean(dims)
    var = ((x - mu.view(shp)) ** 2).mean(dims)
    with torch.no_grad():
        cx.B[prefix + "_rm"] = momentum * rm + (1.0 - momentum) * mu
        cx.B[prefix + "_riv"] = momentum * rv + (1.0 - momentum) * var * (n / max(n - 1, 1))
    return (x - mu.view(shp)) / torch.sqrt(var.view(shp) + eps) * s.view(shp) + b.view(shp)


def _norm(cx, x, prefix, nonlocal_block=False):
    """the layer after a conv: AffineNd (frozen BN, every shipped config) or SpatialBN"""
    cfg = cx.cfg
    if nonlocal_block:
        if cfg.NONLOCAL.USE_BN:
            return _bn(cx, x, prefix, cfg.NONLOCAL.BN_EPSILON, cfg.NONLOCAL.BN_MOMENTUM)
        return _affine(x, cx.P, prefix)
    if cfg.MODEL.USE_AFFINE:
        return _affine(x, cx.P, prefix)
    return _bn(cx, x, prefix, cfg.MODEL.BN_EPSILON, cfg.MODEL.BN_MOMENTUM)


def _conv(x, P, name, stride=(1, 1, 1), pad=(0, 0, 0), dil=(1, 1, 1), groups=1):
    """ConvNd, NCTHW cross-correlation; bias iff `<name>_b` exists; `groups` as Caffe2's `group` argument"""
    return F.conv3d(x, P[name + "_w"], P.get(name + "_b"), stride, pad, dil, groups)


def _conv_affine(cx, x, prefix, stride=(1, 1, 1), pad=(0, 0, 0), dil=(1, 1, 1), groups=1):
    """ModelBuilder.Conv3dAffine / Conv3dBN (lib/models/model_builder_video.py:176-221).  Conv3dBN does not forward
    `dilations=` to ConvNd (:176-183, it lands in **kwargs) while the caller still pads for the dilated kernel
    (resnet_helper.py:57): a batch-norm graph runs the undilated kernel on the dilated pads, and with cfg.DILATIONS = 2
    the residual Sum of res5_0 then fails on its shapes (torch raises here, Caffe2 there)."""
    if not cx.cfg.MODEL.USE_AFFINE:
        dil = (1, 1, 1)
    return _norm(cx, _conv(x, cx.P, prefix, stride, pad, dil, groups), prefix + "_bn")


def _bottleneck(cx, x, prefix, dim_in, dim_out, stride, utc, dilation):
    """bottleneck_transformation_3d + _add_shortcut_3d + _generic_residual_block_3d
    (lib/models/resnet_helper.py:35-119)"""
    h = _relu(cx, _conv_affine(cx, x, prefix + "_branch2a", pad=(utc, 0, 0)), prefix + "_branch2a_bn")
    h = _relu(cx, _conv_affine(cx, h, prefix + "_branch2b", stride=(1, stride, stride),
                               pad=(0, dilation, dilation), dil=(1, dilation, dilation), groups=cx.cfg.RESNETS.NUM_GROUPS),
              prefix + "_branch2b_bn")
    h = _conv_affine(cx, h, prefix + "_branch2c")
    if dim_in == dim_out and stride == 1:
        sc = x
    else:
        sc = _conv_affine(cx, x, prefix + "_branch1", stride=(1, stride, stride))
    y = _relu(cx, h + sc, prefix + "_branch2c_bn")
    cx.B[prefix + "_branch2c_bn"] = y  # the in-place Sum/ReLU make this the block output blob
    return y


def _spacetime_nonlocal(cx, x, prefix, dim_inner):
    """lib/models/nonlocal_helper.py:29-160"""
    cfg, P = cx.cfg, cx.P
    Bn = x.shape[0]
    theta = _conv(x, P, prefix + "_theta")
    xp = _max_pool(cx, x, prefix + "_pool", (1, 2, 2), (1, 2, 2)) if cfg.NONLOCAL.USE_MAXPOOL else x
    phi = _conv(xp, P, prefix + "_phi")
    g = _conv(xp, P, prefix + "_g")
    shp5 = theta.shape
    theta, phi, g = (t.reshape(Bn, dim_inner, -1) for t in (theta, phi, g))
    aff = torch.bmm(theta.transpose(1, 2), phi)                    # BatchMatMul(trans_a=1)
    if cfg.NONLOCAL.USE_SOFTMAX:
        if cfg.NONLOCAL.USE_SCALE:
            aff = aff * dim_inner ** -0.5
        p = torch.softmax(aff, dim=2)
        cx.B[prefix + "_affinity_prob"] = p
    else:
        # dot-product variant (nonlocal_helper.py:107-119): ConstantFill(1) -> ReduceBackSum = number of keys, broadcast,
        # StopGradient, Div -- no scale, no softmax
        p = aff / float(aff.shape[2])
        cx.B[prefix + "_affinity_sc"] = p
    t = torch.bmm(g, p.transpose(1, 2)).reshape(shp5)               # BatchMatMul(trans_b=1)
    out = _conv(t, P, prefix + "_out")
    return _norm(cx, out, prefix + "_bn", nonlocal_block=True)


def _add_nonlocal(cx, x, prefix, dim_inner, group_size=None, pool_stride=None):
    """add_nonlocal / add_nonlocal_group (lib/models/nonlocal_helper.py:163-213)"""
    if group_size is None:
        y = x + _spacetime_nonlocal(cx, x, prefix, dim_inner)
    else:
        N, C, T, H, W = x.shape
        G = int(pool_stride / group_size)
        assert pool_stride % group_size == 0
        if G > 1:
            xg = x.transpose(1, 2).reshape(N * G, group_size, C, H, W).transpose(1, 2)
        else:
            xg = x
        yg = xg + _spacetime_nonlocal(cx, xg, prefix, dim_inner)
        cx.B[prefix + "_sum"] = yg   # the Sum blob lives in the grouped (N*G, C, 4, H, W) shape
        return yg.transpose(1, 2).reshape(N, T, C, H, W).transpose(1, 2) if G > 1 else yg
    cx.B[prefix + "_sum"] = y
    return y


def _layer_norm(x):
    """Caffe2 LayerNorm(axis=1), eps 1e-5, no affine (lib/models/lfb_helper.py:160-166,252-256)"""
    return F.layer_norm(x, x.shape[1:], eps=1e-5)


def _dropout(cx, x, name, ratio):
    """Caffe2 Dropout(is_test=0): mask from oracle.rng over the reference layout"""
    keep = orng.dropout_keep_mask(cx.seed_fn(name), tuple(x.shape), ratio)
    keep = torch.from_numpy(keep)
    cx.B[name + "_mask"] = keep
    return torch.where(keep, x / (1.0 - ratio), torch.zeros_like(x))


def _nl_core(cx, A, Bk, prefix, latent, num_feat2):
    """NLCore (lib/models/lfb_helper.py:170-263)"""
    cfg, P = cx.cfg, cx.P
    theta = _conv(A, P, prefix + "_theta")
    phi = _conv(Bk, P, prefix + "_phi")
    g = _conv(Bk, P, prefix + "_g")
    shp5 = theta.shape
    theta = theta.reshape(-1, latent, 1)
    phi = phi.reshape(-1, latent, num_feat2)
    g = g.reshape(-1, latent, num_feat2)
    aff = torch.bmm(theta.transpose(1, 2), phi)
    if cfg.FBO_NL.SCALE:
        aff = aff * latent ** -0.5
    p = torch.softmax(aff, dim=2)
    cx.B[prefix + "_affinity_prob"] = p
    t = torch.bmm(g, p.transpose(1, 2)).reshape(shp5)
    if cfg.FBO_NL.PRE_ACT:
        if cfg.FBO_NL.PRE_ACT_LN:
            t = _layer_norm(t)
        t = _relu(cx, t, prefix + "_y_ln_relu")
    out = _conv(t, P, prefix + "_out")
    if not cfg.FBO_NL.PRE_ACT:
        out = _layer_norm(out)
        drop_name = prefix + "_ln_drop"
    else:
        drop_name = prefix + "_out_drop"
    if cfg.FBO_NL.LFB_DROPOUT_ON and not cx.test:
        out = _dropout(cx, out, drop_name, cfg.FBO_NL.DROPOUT_RATE)
    return out


def _fbo_head(cx, x, x_name, lfb, num_lfb_feat):
    """add_fbo_head and friends (lib/models/lfb_helper.py:43-127, 266-338)"""
    cfg, P = cx.cfg, cx.P
    bank = lfb.transpose(1, 2).reshape(lfb.shape[0], cfg.LFB.LFB_DIM, num_lfb_feat, 1, 1)  # NTC_to_NCT11
    if cfg.LFB.FBO_TYPE == "avg":
        return F.avg_pool3d(bank, (num_lfb_feat, 1, 1), (1, 1, 1))
    if cfg.LFB.FBO_TYPE == "max":
        return F.max_pool3d(bank, (num_lfb_feat, 1, 1), (1, 1, 1))
    lat = cfg.FBO_NL.LATENT_DIM
    A = x
    if cfg.FBO_NL.INPUT_REDUCE_DIM:
        A = _conv(A, P, x_name + "_fbonl_reduc")
        a_name = x_name + "_fbonl_reduc"
    else:
        a_name = x_name
    if cfg.FBO_NL.INPUT_DROPOUT_ON and not cx.test:
        A = _dropout(cx, A, a_name + "_fbonl_drop", cfg.FBO_NL.DROPOUT_RATE)
    Bk = _conv(bank, P, "lfb_1x1")
    if cfg.FBO_NL.LFB_DROPOUT_ON and not cx.test:
        Bk = _dropout(cx, Bk, "lfb_1x1_drop", cfg.FBO_NL.DROPOUT_RATE)
    for l in range(cfg.FBO_NL.NUM_LAYERS):
        pre = "lfb_nl%d" % l
        A = _nl_core(cx, A, Bk, pre, lat, num_lfb_feat) + A
        if not cfg.FBO_NL.PRE_ACT:
            A = _relu(cx, A, pre + "_relu")
        cx.B[pre + ("_sum" if cfg.FBO_NL.PRE_ACT else "_relu")] = A
    return A


def sigmoid_cross_entropy(logits, labels, scale):
    """Detectron SigmoidCrossEntropyLoss (normalize=1): SURVEY.md Appendix B"""
    t = labels.to(logits.dtype)
    valid = (labels >= 0).to(logits.dtype)
    pos = (logits >= 0).to(logits.dtype)
    l = -logits * (t - pos) + torch.log(1 + torch.exp(logits - 2 * logits * pos))
    normalizer = torch.clamp(valid.sum(), min=1e-5)
    return scale * (l * valid).sum() / normalizer


def softmax_with_loss(logits, labels, scale):
    """Caffe2 SoftmaxWithLoss, integer labels, no weights (upstream
    caffe2/operators/softmax_with_loss_op.cc): loss = scale * sum_i -log(max(P[i][label_i], 1e-20)) / N"""
    p = torch.softmax(logits, dim=1)
    picked = p.gather(1, labels.to(torch.int64).view(-1, 1)).clamp_min(1e-20)
    return scale * (-torch.log(picked)).sum() / logits.shape[0]


def forward(cfg, params, inputs, split="train", lfb_infer_only=False, dtype=torch.float64,
            dropout_seed_fn=None, suffix="", num_gpus=None, decisions=None):
    """resnet_video.create_model (lib/models/resnet_video.py:133-351).
    params: {name: tensor} in reference layouts; inputs: data/labels/proposals/lfb tensors.
    Returns an OrderedDict of named blobs (reference names, NCTHW)."""
    B = OrderedDict()
    cx = _Ctx(cfg, params, split, dropout_seed_fn or (lambda name: 0), B, decisions)
    if decisions is not None:      # which supplied decisions were consumed / which sites decided for themselves
        decisions["_used"], decisions["_missing"] = cx.dec_used, cx.dec_missing
    arc = temporal_arc(cfg)
    n1, n2, n3, n4 = BLOCK_CONFIG[cfg.MODEL.DEPTH]
    x = inputs["data"].to(dtype)
    crop = x.shape[-1]
    frames = x.shape[2]
    pool_stride = int(frames / 2)
    utc1 = arc[0][0]
    x = F.conv3d(x, params["conv1_w"], None, (1, 2, 2), (utc1, 3, 3))
    x = _relu(cx, _norm(cx, x, "res_conv1_bn"), "res_conv1_bn")
    B["res_conv1_bn"] = x
    x = _max_pool(cx, x, "pool1", (1, 3, 3), (1, 2, 2), (0, 1, 1))
    B["pool1"] = x
    w = cfg.RESNETS.NUM_GROUPS * cfg.RESNETS.WIDTH_PER_GROUP
    mod3 = 2 if cfg.MODEL.DEPTH == 101 else cfg.NONLOCAL.LAYER_MOD
    if not cfg.NONLOCAL.CONV3_NONLOCAL:
        mod3 = 1000
    mod4 = cfg.NONLOCAL.LAYER_MOD * 4 - 1 if cfg.MODEL.DEPTH == 101 else cfg.NONLOCAL.LAYER_MOD
    if not cfg.NONLOCAL.CONV4_NONLOCAL:
        mod4 = 1000

    def stage(x, prefix, nblocks, din, dout, stride, utcs, dilation, mod, nlname, group):
        for i in range(nblocks):
            s = 2 if (i == 0 and stride == 2) else 1
            x = _bottleneck(cx, x, "%s_%d" % (prefix, i), din, dout, s, utcs[i], dilation)
            din = dout
            if i % mod == mod - 1:
                if group:
                    x = _add_nonlocal(cx, x, "%s_%d" % (nlname, i), dout // 2, group_size=4, pool_stride=pool_stride)
                else:
                    x = _add_nonlocal(cx, x, "%s_%d" % (nlname, i), dout // 2)
        return x

    x = stage(x, "res2", n1, 64, 256, 1, arc[1], 1, 1000, None, False)
    x = _max_pool(cx, x, "pool2", (2, 1, 1), (2, 1, 1))
    B["pool2"] = x
    # res3 non-local blocks are grouped only when BN is frozen (resnet_video.py:248-272)
    x = stage(x, "res3", n2, 256, 512, 2, arc[2], 1, mod3, "nonlocal_conv3", bool(cfg.MODEL.USE_AFFINE))
    x = stage(x, "res4", n3, 512, 1024, 2, arc[3], 1, mod4, "nonlocal_conv4", False)
    dil5 = 2 if cfg.MODEL.DILATIONS_AFTER_CONV5 else 1
    x = stage(x, "res5", n4, 1024, 2048, 1, arc[4], dil5, 1000, None, False)
    last = "res5_%d_branch2c_bn" % (n4 - 1)
    if cfg.MODEL.FREEZE_BACKBONE:
        x = x.detach()  # StopGradient (resnet_video.py:303-304)

    out_sp = crop // 16
    heads = []
    if cfg.DATASET == "ava":
        # roi_pool (lib/models/head_helper.py:88-123)
        pooled = F.avg_pool3d(x, (frames // 2, 1, 1), (1, 1, 1)).squeeze(2)
        B["blob_pooled_4d"] = pooled
        res = cfg.ROI.XFORM_RESOLUTION
        rois = inputs["proposals"]
        rf = roi_align_torch(pooled, np.asarray(rois, dtype=np.float32), res, 1.0 / cfg.ROI.SCALE_FACTOR)
        B["roi_feat_3d"] = rf
        if res > 1:
            if decisions is not None and decisions.get("roi_bin") is not None:   # the caller's arg-max bin per (RoI, channel)
                binidx = torch.from_numpy(np.ascontiguousarray(decisions["roi_bin"])).to(torch.int64)
                rf = rf.flatten(2).gather(2, binidx.view(rf.shape[0], rf.shape[1], 1)).view(rf.shape[0], rf.shape[1], 1, 1)
                cx.dec_used.add("roi_bin")
            else:
                if cx.record:
                    decisions["roi_bin"] = rf.detach().flatten(2).argmax(2).numpy()
                rf = F.max_pool2d(rf, (res, res), (1, 1))
        feat = rf.reshape(-1, 2048, 1, 1, 1)
        B["box_pooled"] = feat
        x_name = "box_pooled"
        num_lfb_feat = cfg.LFB.WINDOW_SIZE * cfg.AVA.LFB_MAX_NUM_FEAT_PER_STEP
    else:
        # add_basic_head (lib/models/head_helper.py:32-58)
        feat = F.avg_pool3d(x, (pool_stride, out_sp, out_sp), (1, 1, 1))
        x_name = last + "_pooled"
        B[x_name] = feat
        num_lfb_feat = cfg.LFB.WINDOW_SIZE
    heads.append(feat)
    if cfg.LFB.ENABLED and not lfb_infer_only:
        heads.append(_fbo_head(cx, feat, x_name, inputs["lfb"].to(dtype), num_lfb_feat))
    pool5 = torch.cat(heads, dim=1)
    B["pool5"] = pool5
    if lfb_infer_only:
        return B
    h = pool5
    if cfg.TRAIN.DROPOUT_RATE > 0 and not cx.test:
        h = _dropout(cx, h, "pool5_dropout", cfg.TRAIN.DROPOUT_RATE)
    logits = F.linear(h.flatten(1), params["pred_w"], params["pred_b"])
    B["pred"] = logits
    scale = 1.0 / (num_gpus if num_gpus is not None else cfg.NUM_GPUS)
    if cfg.MODEL.MULTI_LABEL:
        B["prob"] = torch.sigmoid(logits)
        if split == "train":
            B["loss"] = sigmoid_cross_entropy(logits, inputs["labels"], scale)
    else:
        # Softmax / SoftmaxWithLoss (resnet_video.py:339-350; EPIC-Kitchens verb / noun classification)
        B["prob"] = torch.softmax(logits, dim=1)
        if split == "train":
            B["loss"] = softmax_with_loss(logits, inputs["labels"], scale)
    return B


def run(cfg, params_np, inputs_np, split="train", dtype=torch.float64, backward=True,
        dropout_seed_fn=None, lfb_infer_only=False, num_gpus=None, threads=None, decisions=None):
    """Convenience wrapper: numpy in, (blobs, grads) out (all torch tensors of `dtype`).

    decisions (optional) = {"relu": {blob: bool array, reference layout}, "pool": {blob: selected tap per output
    element}, "roi_bin": (R, C) arg-max bin}: the discrete decisions of ANOTHER evaluation of the same network (the
    engine under test, vlfb.engine.Engine.discrete_decisions).  The network is piecewise linear in between, and a
    pre-activation within fp32 rounding of zero is decided one way in fp64 and possibly the other way in fp32; ONE such
    unit of res5 moves every upstream gradient by ~1e-3 when the loss gradient is sparse (AVA: a few RoIs).  With the
    decisions supplied the comparison measures the arithmetic on identical branches."""
    if threads:
        torch.set_num_threads(threads)
    P = OrderedDict()
    spec = param_spec(cfg, lfb_infer_only)
    for k, v in params_np.items():
        t = torch.from_numpy(np.asarray(v)).to(dtype)
        if backward and spec.get(k, {}).get("trainable", False):
            t.requires_grad_(True)
        P[k] = t
    I = {}
    for k, v in inputs_np.items():
        t = torch.from_numpy(np.asarray(v))
        I[k] = t if t.dtype in (torch.int32, torch.int64) or k == "proposals" else t.to(dtype)
    blobs = forward(cfg, P, I, split, lfb_infer_only, dtype, dropout_seed_fn, num_gpus=num_gpus, decisions=decisions)
    grads = OrderedDict()
    if backward and "loss" in blobs:
        blobs["loss"].backward()
        for k, t in P.items():
            if t.requires_grad and t.grad is not None:
                grads[k] = t.grad
    return blobs, grads
