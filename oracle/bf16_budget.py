"""bf16 error budget of the hot path -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The throughput path stores activations and activation gradients in bf16 and feeds bf16 operands to the
MFMAs (fp32 accumulation, fp32 affinity / softmax / loss, fp32 parameter gradients).  This module
re-runs the fp64 oracle with a bf16 ROUNDING inserted at exactly those storage points, one class at a
time, so that tests can (a) state which rounding class the measured gradient error of the engine comes
from and (b) bound the engine's per-tensor error by the budget instead of by a guessed tolerance.

Classes (`variant` keys):
  fwd      values of every stored activation blob (conv/affine/ReLU outputs, block exits, non-local /
           FBO convs) are rounded to bf16
  w        MFMA weight operands are rounded to bf16
  bwd      finished gradients of branch activations are rounded to bf16
  bwd_res  finished gradients of the residual stream (bottleneck exits, non-local sums) are rounded

Measured on the 2-clip 16x64^2 charades_r50_baseline test case (scratch/emu_bf16.py, relative L2 of the
parameter gradients against the exact oracle; median / p90 / max):
  bwd + bwd_res only          1.5e-3 / 3.2e-3 / 7.8e-3      <- ALL backward roundings together
  bwd_res only                1.4e-3 / 2.3e-3 / 6.7e-3
  w only                      1.3e-2 / 2.3e-2 / 1.2e-1
  fwd only                    1.6e-2 / 2.8e-2 / 1.7e-1
  everything                  1.8e-2 / 3.9e-2 / 1.7e-1 (conv1_w)
i.e. the gradient error of the bf16 path is set by the FORWARD perturbation (bf16 activations / weights
flip ReLU and max-pool decisions and perturb every saved activation the wgrads contract with), not by
bf16 gradient accumulation: keeping the residual-stream gradients in fp32 would move the p90 from
3.9e-2 to 3.8e-2 at twice the gradient traffic.
"""
import numpy as np
import torch

from . import model as om


def _r16(t):
    return t.to(torch.bfloat16).to(t.dtype)


class _Store(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fwd_r, bwd_r):
        ctx.bwd_r = bwd_r
        return _r16(x) if fwd_r else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (_r16(g) if ctx.bwd_r else g), None, None


ALL = dict(fwd=True, w=True, bwd=True, bwd_res=True)


def run(cfg, params, inputs, variant=None, dropout_seed_fn=None, dtype=torch.float64):
    """oracle.model.run with bf16 roundings of the given classes; returns (blobs, grads)"""
    v = dict(ALL if variant is None else variant)
    f_act, b_act, b_res, w_op = v.get("fwd", False), v.get("bwd", False), v.get("bwd_res", False), v.get("w", False)
    saved = om._conv, om._conv_affine, om._bottleneck, om._add_nonlocal

    def conv(x, P, name, *a, **k):
        if w_op:
            P = dict(P)
            P[name + "_w"] = _Store.apply(P[name + "_w"], True, False)
        y = saved[0](x, P, name, *a, **k)
        if "_branch" not in name and name != "conv1":
            y = _Store.apply(y, f_act, b_act)
        return y

    def conv_affine(cx, x, prefix, *a, **k):
        y = saved[1](cx, x, prefix, *a, **k)
        return y if prefix.endswith("_branch2c") else _Store.apply(y, f_act, b_act)

    def bottleneck(cx, x, prefix, *a, **k):
        return _Store.apply(saved[2](cx, x, prefix, *a, **k), f_act, b_res)

    def add_nonlocal(cx, x, prefix, *a, **k):
        return _Store.apply(saved[3](cx, x, prefix, *a, **k), f_act, b_res)

    om._conv, om._conv_affine, om._bottleneck, om._add_nonlocal = conv, conv_affine, bottleneck, add_nonlocal
    try:
        return om.run(cfg, params, inputs, "train", dtype, True, dropout_seed_fn)
    finally:
        om._conv, om._conv_affine, om._bottleneck, om._add_nonlocal = saved


def rel(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    d = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (d if d > 0 else 1.0))


def gradient_budget(cfg, params, inputs, exact_grads, variant=None, dropout_seed_fn=None):
    """{param: relative L2 error of its gradient under the emulated roundings}"""
    _, g = run(cfg, params, inputs, variant, dropout_seed_fn)
    return {n: rel(g[n].detach().numpy(), exact_grads[n].detach().numpy()) for n in exact_grads if n in g}
