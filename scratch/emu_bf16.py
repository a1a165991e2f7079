"""CPU error-budget experiment (test infrastructure, not product): emulate WHERE the bf16 engine rounds
(stored activations, stored activation gradients, MFMA weight operands) inside the fp64 oracle and
measure the per-parameter gradient error of each rounding class, to decide which accumulators must be
fp32.  Usage: python scratch/emu_bf16.py [preset] [variant ...]"""
import sys, collections
sys.path.insert(0, 'video-long-term-feature-banks_amd/lib'); sys.path.insert(0, '.')
import numpy as np, torch
from vlfb.presets import load_preset
from core.config import config as cfg
from oracle import model as om

SMALL = ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 2, "TRAIN.VIDEO_LENGTH", 16, "TRAIN.CROP_SIZE", 64]


def r16(t):
    return t.to(torch.bfloat16).to(t.dtype)


class Store(torch.autograd.Function):
    """a tensor the engine keeps in HBM: value rounded (fwd_r), finished gradient rounded (bwd_r)"""
    @staticmethod
    def forward(ctx, x, fwd_r, bwd_r):
        ctx.bwd_r = bwd_r
        return r16(x) if fwd_r else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (r16(g) if ctx.bwd_r else g), None, None


def rel(a, b):
    d = float(b.norm())
    return float((a - b).norm()) / (d if d > 0 else 1.0)


def run(preset, variant):
    load_preset(preset, SMALL)
    inputs = om.synth_inputs(cfg, 2, "train", seed=cfg.RNG_SEED, rois_per_clip=[2, 3] if cfg.DATASET == "ava" else None,
                             crop=64, frames=16)
    params = om.synth_params(cfg, seed=cfg.RNG_SEED)
    F_ACT = variant.get("fwd", False)        # activations stored bf16
    B_ACT = variant.get("bwd", False)        # branch gradients stored bf16
    B_RES = variant.get("bwd_res", False)    # residual-stream (block output / NL sum) gradients stored bf16
    W_OP = variant.get("w", False)           # weights rounded to bf16 MFMA operands
    orig_conv, orig_ca, orig_bott, orig_nl = om._conv, om._conv_affine, om._bottleneck, om._add_nonlocal

    def conv(x, P, name, *a, **k):
        if W_OP:
            P = dict(P); P[name + "_w"] = Store.apply(P[name + "_w"], True, False)
        y = orig_conv(x, P, name, *a, **k)
        if "_branch" not in name and name != "conv1":     # NL / FBO convs: outputs are stored blobs
            y = Store.apply(y, F_ACT, B_ACT)
        return y

    def conv_affine(cx, x, prefix, *a, **k):
        y = orig_ca(cx, x, prefix, *a, **k)
        if prefix.endswith("_branch2c"):
            return y                                       # fused into the block-exit epilogue
        return Store.apply(y, F_ACT, B_ACT)

    def bott(cx, x, prefix, *a, **k):
        return Store.apply(orig_bott(cx, x, prefix, *a, **k), F_ACT, B_RES)

    def add_nl(cx, x, prefix, *a, **k):
        return Store.apply(orig_nl(cx, x, prefix, *a, **k), F_ACT, B_RES)

    om._conv, om._conv_affine, om._bottleneck, om._add_nonlocal = conv, conv_affine, bott, add_nl
    try:
        blobs, grads = om.run(cfg, params, inputs, "train", torch.float64, True, lambda name: 7)
    finally:
        om._conv, om._conv_affine, om._bottleneck, om._add_nonlocal = orig_conv, orig_ca, orig_bott, orig_nl
    return blobs, grads


VARIANTS = collections.OrderedDict([
    ("exact", {}),
    ("all_bf16", dict(fwd=True, bwd=True, bwd_res=True, w=True)),
    ("fwd_only", dict(fwd=True, w=True)),
    ("bwd_only", dict(bwd=True, bwd_res=True)),
    ("bwd_branch_only", dict(bwd=True)),
    ("bwd_res_only", dict(bwd_res=True)),
    ("all_but_res_grad", dict(fwd=True, bwd=True, w=True)),
    ("w_only", dict(w=True)),
    ("act_fwd_only", dict(fwd=True)),
])

if __name__ == "__main__":
    preset = sys.argv[1] if len(sys.argv) > 1 else "charades_r50_baseline"
    names = sys.argv[2:] or list(VARIANTS)
    torch.set_num_threads(8)
    ref_b, ref_g = run(preset, {})
    for v in names:
        if v == "exact":
            continue
        b, g = run(preset, VARIANTS[v])
        errs = sorted(((rel(g[n], ref_g[n]), n) for n in ref_g if float(ref_g[n].norm()) > 1e-12), reverse=True)
        e = np.array([x for x, _ in errs])
        print("%-18s prob %.2e loss %.2e | grads median %.2e p90 %.2e max %.2e (%s) | conv1_w %.2e" % (
            v, rel(b["prob"], ref_b["prob"]), abs(float(b["loss"] - ref_b["loss"])) / abs(float(ref_b["loss"])),
            np.median(e), np.sort(e)[int(0.9 * (len(e) - 1))], e[0], errs[0][1],
            rel(g["conv1_w"], ref_g["conv1_w"]) if "conv1_w" in g else -1))
