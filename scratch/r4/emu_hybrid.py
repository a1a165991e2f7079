"""CPU error-budget experiment for the HYBRID path (test infrastructure, not product): exact forward (the split-bf16
forward is within 7e-6 of fp64), backward with fp16 roundings where the hybrid engine rounds: stored activation
gradients (branch / residual stream), the saved-activation operand of WGRAD, the weight operand of DGRAD.
Usage: python scratch/r4/emu_hybrid.py [preset] [bits]"""
import sys, collections
sys.path.insert(0, 'video-long-term-feature-banks_amd/lib'); sys.path.insert(0, '.')
import numpy as np, torch
import torch.nn.functional as F
from vlfb.presets import load_preset
from core.config import config as cfg
from oracle import model as om

import os
NCLIP = int(os.environ.get("EMU_CLIPS", 2)); FR = int(os.environ.get("EMU_FRAMES", 16)); CROP = int(os.environ.get("EMU_CROP", 64))
SMALL = ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", NCLIP, "TRAIN.VIDEO_LENGTH", FR, "TRAIN.CROP_SIZE", CROP]
BITS = 11
QG = [False]


def rq(t, bits=None):
    bits = bits or BITS
    m, e = torch.frexp(t)
    s = float(1 << bits)
    return torch.ldexp(torch.round(m * s) / s, e)


LOSS_SCALE = [None]      # EMU_LOG2S: stored gradients are true fp16 values of g * 2^EMU_LOG2S (range, subnormals, inf)


def rqg(g):
    """rounding of a STORED GRADIENT: 11 significant bits with unbounded exponent, or -- with a loss scale -- real fp16"""
    if LOSS_SCALE[0] is None:
        return rq(g)
    S = LOSS_SCALE[0]
    return (g * S).to(torch.float16).to(g.dtype) / S


class Store(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bwd_r):
        ctx.bwd_r = bwd_r
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return (rqg(g) if ctx.bwd_r else g), None


_ORIG_CONV3D = F.conv3d


class ConvQ(torch.autograd.Function):
    """exact forward; backward contracts ROUNDED operands: dgrad with q(w), wgrad with q(x)"""
    @staticmethod
    def forward(ctx, x, w, b, stride, pad, dil, qx, qw):
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, dil, qx, qw, b is not None)
        return _ORIG_CONV3D(x, w, b, stride, pad, dil)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        stride, pad, dil, qx, qw, has_b = ctx.cfg
        if QG[0]:
            g = rqg(g)
        gi = gw = gb = None
        if ctx.needs_input_grad[0]:
            gi = torch.nn.grad.conv3d_input(x.shape, rq(w) if qw else w, g, stride, pad, dil)
        if ctx.needs_input_grad[1]:
            gw = torch.nn.grad.conv3d_weight(rq(x) if qx else x, w.shape, g, stride, pad, dil)
        if has_b and ctx.needs_input_grad[2]:
            gb = g.sum((0, 2, 3, 4))
        return gi, gw, gb, None, None, None, None, None


_ORIG_LINEAR = F.linear


class LinQ(torch.autograd.Function):
    """the classifier (FCStep): fp16 dlogits, single-term fp16 weights in the input gradient, fp16 operands in the weight gradient"""
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return _ORIG_LINEAR(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = rqg(g)
        return rqg(g @ rq(w)), g.t() @ rq(x), g.sum(0)


def rel(a, b):
    d = float(b.norm())
    return float((a - b).norm()) / (d if d > 0 else 1.0)


def run(preset, variant):
    load_preset(preset, SMALL)
    inputs = om.synth_inputs(cfg, NCLIP, "train", seed=cfg.RNG_SEED, rois_per_clip=([2, 3] * NCLIP)[:NCLIP] if cfg.DATASET == "ava" else None,
                             crop=CROP, frames=FR)
    params = om.synth_params(cfg, seed=cfg.RNG_SEED)
    B_ACT, B_RES = variant.get("bwd", False), variant.get("bwd_res", False)
    QX, QW = variant.get("qx", False), variant.get("qw", False)
    QG[0] = variant.get("qg", False)
    orig_conv, orig_ca, orig_bott, orig_nl = om._conv, om._conv_affine, om._bottleneck, om._add_nonlocal
    orig_fconv = F.conv3d

    QW_IN = variant.get("qw_in")            # single-term fp16 DGRAD weights only in the layers whose name starts with one of these

    def conv(x, P, name, stride=(1, 1, 1), pad=(0, 0, 0), dil=(1, 1, 1), groups=1):
        assert groups == 1
        qw = QW or (QW_IN is not None and name.startswith(tuple(QW_IN)))
        y = ConvQ.apply(x, P[name + "_w"], P.get(name + "_b"), stride, pad, dil, QX, qw)
        if "_branch" not in name and name != "conv1":
            y = Store.apply(y, B_ACT)
        return y

    def conv_affine(cx, x, prefix, *a, **k):
        y = orig_ca(cx, x, prefix, *a, **k)
        if prefix.endswith("_branch2c"):
            return y
        return Store.apply(y, B_ACT)

    STAGE = variant.get("stage", False)     # single-term fp16 trunk gradients where the engine has no two-term slot:
    # the inputs of the stage-first blocks (two conv contributors: rounded twice) and the tensors the pools write

    def bott(cx, x, prefix, *a, **k):
        y = Store.apply(orig_bott(cx, x, prefix, *a, **k), B_RES)
        if STAGE and prefix == "res2_2":
            y = Store.apply(y, True)                      # written by the pool2 backward
        return y

    NLIN = variant.get("nlin", False)       # ... and one rounding of the trunk gradient at the input of every non-local block
    # (its slot is two-term, but the pool-backward contribution of the phi / g branch is added to the high term only)

    def add_nl(cx, x, prefix, *a, **k):
        if NLIN:
            x = Store.apply(x, True)
        y = Store.apply(orig_nl(cx, x, prefix, *a, **k), B_RES)
        if STAGE and prefix in ("nonlocal_conv3_3", "nonlocal_conv4_5"):
            y = Store.apply(Store.apply(y, True), True)   # input of res4_0 / res5_0: shortcut dgrad, then + 2a dgrad
        return y

    orig_pool = om._max_pool

    def max_pool(cx, x, name, *a, **k):
        y = orig_pool(cx, x, name, *a, **k)
        if STAGE and name in ("pool1", "pool2"):
            y = Store.apply(Store.apply(y, True), True)   # input of res2_0 / res3_0
        return y
    om._max_pool = max_pool

    STEM = variant.get("stem", False)

    def fconv(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        # the stem (oracle/model.py calls F.conv3d directly for conv1): the clip as an fp16 copy in WGRAD, its output gradient
        # stored in fp16 like every branch gradient
        if STEM and w.dim() == 5 and w.shape[1] == 3 and groups == 1:
            y = ConvQ.apply(x, w, b, stride, padding, (1, 1, 1), True, False)
            return Store.apply(y, True)
        return _ORIG_CONV3D(x, w, b, stride, padding, dilation, groups)

    HEAD = variant.get("head", False)       # the head's fp16 gradient storage: classifier, dropout, pools, RoIAlign, FBO core
    orig_avg, orig_roi, orig_drop, orig_ln, orig_core = F.avg_pool3d, om.roi_align_torch, om._dropout, om._layer_norm, om._nl_core

    HEAD_FC = variant.get("head_fc", HEAD)      # the classifier alone
    HEAD_REST = variant.get("head_rest", HEAD)  # everything between the classifier and res5

    def st(t):
        return Store.apply(t, True) if (HEAD_REST and torch.is_tensor(t) and t.requires_grad) else t

    def linear(x, w, b=None):
        return LinQ.apply(x, w, b) if HEAD_FC else _ORIG_LINEAR(x, w, b)

    def avg_pool3d(x, *a, **k):
        return st(orig_avg(st(x), *a, **k))

    def roi_align(x, *a, **k):
        return st(orig_roi(st(x), *a, **k))

    def dropout(cx, x, *a, **k):
        return st(orig_drop(cx, st(x), *a, **k))

    def layer_norm(x):
        return st(orig_ln(st(x)))

    def nl_core(cx, A, Bk, *a, **k):
        return st(orig_core(cx, st(A), st(Bk), *a, **k))

    om._conv, om._conv_affine, om._bottleneck, om._add_nonlocal = conv, conv_affine, bott, add_nl
    om.F.linear, om.F.avg_pool3d, om.roi_align_torch, om._dropout, om._layer_norm, om._nl_core = \
        linear, avg_pool3d, roi_align, dropout, layer_norm, nl_core
    om.F.conv3d = fconv
    try:
        blobs, grads = om.run(cfg, params, inputs, "train", torch.float64, True, lambda name: 7)
    finally:
        om.F.conv3d = _ORIG_CONV3D
        om._max_pool = orig_pool
        om.F.linear, om.F.avg_pool3d, om.roi_align_torch, om._dropout, om._layer_norm, om._nl_core = \
            _ORIG_LINEAR, orig_avg, orig_roi, orig_drop, orig_ln, orig_core
        om._conv, om._conv_affine, om._bottleneck, om._add_nonlocal = orig_conv, orig_ca, orig_bott, orig_nl
    return blobs, grads


VARIANTS = collections.OrderedDict([
    ("all", dict(bwd=True, bwd_res=True, qx=True, qw=True)),
    ("grads_only", dict(bwd=True, bwd_res=True)),
    ("branch_grads+ops", dict(bwd=True, qx=True, qw=True)),
    ("res_grads_only", dict(bwd_res=True)),
    ("ops_only", dict(qx=True, qw=True)),
    ("qx_only", dict(qx=True)),
    ("qw_only", dict(qw=True)),
    ("e_fp32store_qg_qx", dict(qg=True, qx=True)),
    ("h1_grads+qx", dict(bwd=True, bwd_res=True, qx=True)),
    ("h1_branch+qx", dict(bwd=True, qx=True)),
    ("branch_only", dict(bwd=True)),
    ("mix_like+stem", dict(bwd=True, qx=True, stem=True)),
    ("stem_only", dict(stem=True)),
    ("mix_like+stem+stage", dict(bwd=True, qx=True, stem=True, stage=True)),
    ("mix_like+stem+stage+nlin", dict(bwd=True, qx=True, stem=True, stage=True, nlin=True)),
    ("w2_off_res2", dict(bwd=True, qx=True, stem=True, qw_in=("res2",))),
    ("w2_off_res23", dict(bwd=True, qx=True, stem=True, qw_in=("res2", "res3", "nonlocal_conv3"))),
    ("w2_off_res234", dict(bwd=True, qx=True, stem=True, qw_in=("res2", "res3", "nonlocal_conv3", "res4", "nonlocal_conv4"))),
    ("mix_like+head", dict(bwd=True, qx=True, head=True)),
    ("head_only", dict(head=True)),
    ("head_fc_only", dict(head_fc=True, head_rest=False)),
    ("head_rest_only", dict(head_fc=False, head_rest=True)),
    ("mix_like+stem+stage+head", dict(bwd=True, qx=True, stem=True, stage=True, head=True)),
])

if __name__ == "__main__":
    preset = sys.argv[1] if len(sys.argv) > 1 else "ava_r50_lfb_nl"
    if len(sys.argv) > 2:
        BITS = int(sys.argv[2])
    torch.set_num_threads(8)
    if os.environ.get("EMU_LOG2S"):
        LOSS_SCALE[0] = 2.0 ** float(os.environ["EMU_LOG2S"])
    ref_b, ref_g = run(preset, {})
    only = os.environ.get("EMU_ONLY")
    for v, var in VARIANTS.items():
        if only and v not in only.split(","):
            continue
        b, g = run(preset, var)
        errs = sorted(((rel(g[n], ref_g[n]), n) for n in ref_g if float(ref_g[n].norm()) > 1e-12), reverse=True)
        e = np.array([x for x, _ in errs])
        print("%-18s grads median %.2e p90 %.2e max %.2e (%s) 2nd %.2e (%s) | conv1_w %.2e" % (
            v, np.median(e), np.sort(e)[int(0.9 * (len(e) - 1))], e[0], errs[0][1], e[1], errs[1][1],
            rel(g["conv1_w"], ref_g["conv1_w"]) if "conv1_w" in g else -1), flush=True)
