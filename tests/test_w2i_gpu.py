"""The two-term fp16 DGRAD with the weight terms interleaved per 64-channel k-tile (vlfb_conv_desc.math = VLFB_MATH_F16W2,
weights VLFB_MIX_W2I / VLFB_MIXH_W2I; gemm_nt_kernel<.., W2I>): one gradient tile in LDS per (Wh, Wl) pair of weight tiles.

It computes the SAME product as the doubled-tap form of VLFB_MIX_W2 (kt' = 2 kt, dt = 0), so the checks are
  * the weight copy is the doubled-tap copy re-ordered, bit for bit;
  * the result equals the doubled-tap launch up to the fp32 accumulation order (the terms alternate per k-tile instead of
    running one after the other): 2e-6 on fp32 outputs, one fp16 rounding on fp16 outputs;
  * against fp64 torch on the 22-bit weights: the fp32-accumulate bar of a 16-bit DGRAD;
  * the epilogues of the backward (ReLU mask + second contribution, fp32 output, the two-term gradient R_lo / O_lo).
"""
import math

import pytest
import torch
import torch.nn.functional as F

from gpu_util import dev, rel_err, to_ncthw, to_nthwc, w_to_kernel
from test_kernels_gpu import conv_out_dims, geom_kwargs

pytestmark = pytest.mark.gpu

hip = None


def setup_module(module):
    from vlfb import hip as h
    module.hip = h
    h.lib()


def gpu(t, dtype=None):
    t = t.to(dev())
    return t.to(dtype) if dtype is not None else t


CASES = {
    # name: (N, Cin, Cout, T, H, W, k, stride, pad, dil)      (DGRAD: rows = input positions, K = taps * Cout, columns = Cin)
    "pw_64_columns": (1, 64, 256, 2, 9, 9, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),          # 128 x 64 tiles, 4 pairs of k-tiles
    "pw_one_pair": (2, 256, 64, 2, 9, 9, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),            # ONE pair; two column tiles
    "pw_ragged_columns": (1, 136, 128, 2, 7, 7, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),     # 136 = 128 + 8 columns
    "spatial3": (2, 128, 128, 3, 14, 14, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),            # 18 pairs, padding compares
    "temporal3": (1, 256, 128, 4, 7, 7, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),
    "spatial3_dil2": (1, 128, 64, 2, 14, 14, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2)),
    "deep": (1, 128, 1024, 2, 7, 7, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),                 # 16 pairs of one tap
}


def weights(w, scale):
    """device copies of w [Cout][Cin][k] (+ per-Cout scale): (doubled-tap two-term copy, interleaved copy, the 22-bit value)"""
    Cout, Cin = w.shape[:2]
    taps = w[0, 0].numel()
    wk = gpu(w_to_kernel(w).contiguous())
    sc = gpu(scale)
    w2 = torch.empty(Cin, 2, taps, Cout, device=dev(), dtype=torch.float16)
    wi = torch.empty(Cin, taps, Cout // 64, 2, 64, device=dev(), dtype=torch.float16)
    hip.call("vlfb_weight_prep", hip.ptr(wk), hip.ptr(sc), None, hip.ptr(w2), hip.MIX_W2, Cout, taps, Cin)
    hip.call("vlfb_weight_prep", hip.ptr(wk), hip.ptr(sc), None, hip.ptr(wi), hip.MIX_W2I, Cout, taps, Cin)
    torch.cuda.synchronize()
    # [Cin][term][taps][Cout] -> [Cin][taps][Cout / 64][term][64]
    assert torch.equal(wi.cpu(), w2.cpu().view(Cin, 2, taps, Cout // 64, 64).permute(0, 2, 3, 1, 4).contiguous())
    val = (w2[:, 0].double() + w2[:, 1].double()) / hip.MIX_W2_SCALE                 # [Cin][taps][Cout]
    return w2, wi, val.cpu()


@pytest.mark.parametrize("case", sorted(CASES))
def test_interleaved_two_term_dgrad(case):
    N, Cin, Cout, T, H, W, k, s, p, d = CASES[case]
    gen = torch.Generator().manual_seed(sum(map(ord, case)))
    taps = k[0] * k[1] * k[2]
    w = torch.randn(Cout, Cin, *k, generator=gen) * (1.0 / math.sqrt(Cout * taps))
    scale = torch.rand(Cout, generator=gen) + 0.5
    To, Ho, Wo = conv_out_dims(T, H, W, k, s, p, d)
    dy = torch.randn(N, Cout, To, Ho, Wo, generator=gen).half()
    w2, wi, wval = weights(w, scale)
    # fp64 reference on the 22-bit weights
    wref = wval.view(Cin, *k, Cout).permute(4, 0, 1, 2, 3).contiguous()             # [Cout][Cin][k]
    xd = torch.zeros(N, Cin, T, H, W, dtype=torch.float64, requires_grad=True)
    gx, = torch.autograd.grad(F.conv3d(xd, wref, None, s, p, d), xd, dy.double())
    G = gpu(to_nthwc(dy.float()), torch.float16)
    rows = dict(N=N, Tr=T, Hr=H, Wr=W, Ts=To, Hs=Ho, Ws=Wo, Cs=Cout, Cn=Cin)
    g1 = geom_kwargs(k, s, p, d)
    assert k[0] == 1 or (k[1] == 1 and k[2] == 1)
    if k[0] == 1:
        g2, rows2 = dict(g1, kt=2, dt=0), rows
    else:   # (k x 1 x 1: the doubled-tap form runs T as H -- ConvStep._w2_geometry)
        g2 = dict(kt=2, kh=k[0], kw=1, st=1, sh=1, sw=1, pt=0, ph=p[0], pw=0, dt=0, dh=d[0], dw=1)
        rows2 = dict(N=N, Tr=1, Hr=T, Wr=H * W, Ts=1, Hs=To, Ws=Ho * Wo, Cs=Cout, Cn=Cin)
    alpha = 0.5 / hip.MIX_W2_SCALE
    for out_dtype, tdt in ((hip.F32, torch.float32), (hip.F16, torch.float16)):
        d_i = hip.conv_desc(mode=hip.DGRAD, dtype=hip.F16, out_dtype=out_dtype, alpha=alpha, math=hip.MATH_F16W2, **rows, **g1)
        d_2 = hip.conv_desc(mode=hip.DGRAD, dtype=hip.F16, out_dtype=out_dtype, alpha=alpha, **rows2, **g2)
        assert " w2" in hip.conv_plan(d_i), hip.conv_plan(d_i)
        o_i = torch.full((N, T, H, W, Cin), float("nan"), device=dev(), dtype=tdt)
        o_2 = torch.full((N, T, H, W, Cin), float("nan"), device=dev(), dtype=tdt)
        # (fp32 output: O_lo alone = the output rounded to fp16, the copy a 16-bit reader of this gradient takes)
        o_h = torch.full((N, T, H, W, Cin), float("nan"), device=dev(), dtype=torch.float16) if tdt == torch.float32 else None
        hip.conv_run(d_i, G, wi, None, o_i, O_lo=o_h)
        hip.conv_run(d_2, G, w2, None, o_2)
        ref = 0.5 * gx
        if tdt == torch.float32:
            assert torch.equal(o_h, o_i.half()), "the fp16 rounding next to the fp32 output"
            assert rel_err(to_ncthw(o_i), ref) < 2e-6, "fp32 output vs fp64"
            assert rel_err(o_i, o_2) < 2e-6, "vs the doubled-tap launch"
        else:
            assert rel_err(to_ncthw(o_i.float()), ref) < 4e-4, "fp16 output vs fp64"
            # (the fp32 accumulators differ by the accumulation order -- relative to the sum of |products|, not to the result --
            # and then round to fp16 once)
            diff = (o_i.float() - o_2.float()).abs()
            slack = 4e-6 * float(o_2.float().abs().max())
            assert (diff <= torch.clamp(o_2.float().abs() * 2.0 ** -10, min=2.0 ** -24) + slack).all(), "one fp16 rounding from the doubled-tap launch"
            assert (diff > 0).float().mean() < 0.02
    # ---- the backward's epilogue: ReLU mask of the block input + the other contribution to its gradient
    mask = torch.randn(N, Cin, T, H, W, generator=gen).half()
    add = torch.randn(N, Cin, T, H, W, generator=gen).half()
    d_i = hip.conv_desc(mode=hip.DGRAD, dtype=hip.F16, out_dtype=hip.F16, alpha=alpha, math=hip.MATH_F16W2, **rows, **g1)
    o_i = torch.full((N, T, H, W, Cin), float("nan"), device=dev(), dtype=torch.float16)
    hip.conv_run(d_i, G, wi, None, o_i, R=gpu(to_nthwc(add.float()), torch.float16), mask=gpu(to_nthwc(mask.float()), torch.float16))
    ref = torch.where(mask.double() > 0, 0.5 * gx + add.double(), torch.zeros_like(gx))
    assert rel_err(to_ncthw(o_i.float()), ref) < 4e-4, "mask + residual"
    # ---- two-term gradient in and out (the residual stream of the "mix" backward): O + O_lo = alpha * acc + R + R_lo
    add_lo = (torch.randn(N, Cin, T, H, W, generator=gen) * 1e-4).half()
    o_hi = torch.full((N, T, H, W, Cin), float("nan"), device=dev(), dtype=torch.float16)
    o_lo = torch.full((N, T, H, W, Cin), float("nan"), device=dev(), dtype=torch.float16)
    hip.conv_run(d_i, G, wi, None, o_hi, R=gpu(to_nthwc(add.float()), torch.float16), R_lo=gpu(to_nthwc(add_lo.float()), torch.float16), O_lo=o_lo)
    ref = 0.5 * gx + add.double() + add_lo.double()
    assert rel_err(to_ncthw(o_hi.double() + o_lo.double()), ref) < 3e-6, "two-term output"


def test_interleaved_two_term_dgrad_refusals():
    ok = dict(mode=hip.DGRAD, dtype=hip.F16, out_dtype=hip.F16, math=hip.MATH_F16W2, N=1, Tr=2, Hr=8, Wr=8, Ts=2, Hs=8, Ws=8, Cs=128, Cn=64)
    hip.conv_workspace_bytes(hip.conv_desc(**ok))
    for bad in (dict(Cs=96), dict(mode=hip.FPROP), dict(dtype=hip.F32, out_dtype=hip.F32), dict(sh=2, sw=2, Hs=4, Ws=4), dict(algo=hip.ALGO_PIPE256)):
        with pytest.raises(hip.VlfbError):
            hip.conv_workspace_bytes(hip.conv_desc(**dict(ok, **bad)))
    # the interleaved copy needs whole 64-channel runs
    w = torch.zeros(96 * 64, device=dev())
    out = torch.empty(2 * 96 * 64, device=dev(), dtype=torch.float16)
    with pytest.raises(hip.VlfbError):
        hip.call("vlfb_weight_prep", hip.ptr(w), None, None, hip.ptr(out), hip.MIX_W2I, 96, 1, 64)


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_fp32_output_of_a_16_bit_launch_leaves_its_rounding(tdt):
    """vlfb_conv_args.O_lo next to out_dtype VLFB_F32 on 16-bit operands (the 128-row kernel, batched plain rows: the d theta
    product of a non-local block; the 256-row pipelined kernel: the out conv's DGRAD in res4): O_lo = T(O), bit for bit"""
    code = hip.dtype_code(tdt)
    gen = torch.Generator().manual_seed(11)
    B, M, K, Nc = 3, 200, 160, 136
    a = gpu(torch.randn(B, M, K, generator=gen), tdt)
    w = gpu(torch.randn(B, Nc, K, generator=gen) * 0.1, tdt)
    d = hip.conv_desc(mode=hip.FPROP, dtype=code, out_dtype=hip.F32, N=1, Tr=1, Hr=1, Wr=M, Ts=1, Hs=1, Ws=M, Cs=K, Cn=Nc, batch=B,
                      a_bstride=M * K, b_bstride=Nc * K, o_bstride=M * Nc, alpha=0.25)
    o = torch.full((B, M, Nc), float("nan"), device=dev(), dtype=torch.float32)
    oh = torch.full((B, M, Nc), float("nan"), device=dev(), dtype=tdt)
    hip.conv_run(d, a, w, None, o, O_lo=oh)
    assert rel_err(o, 0.25 * torch.einsum("bmk,bnk->bmn", a.double().cpu(), w.double().cpu())) < 1e-5
    assert torch.equal(oh, o.to(tdt))
    # the 256-row kernel (1x1x1 DGRAD, 512 columns, K = 1024)
    N, T, H, W, Cout, Cin = 1, 2, 28, 28, 1024, 512
    dy = gpu(torch.randn(N, T, H, W, Cout, generator=gen), tdt)
    wd = gpu(torch.randn(Cin, Cout, generator=gen) * 0.03, tdt)
    d8 = hip.conv_desc(mode=hip.DGRAD, dtype=code, out_dtype=hip.F32, N=N, Tr=T, Hr=H, Wr=W, Ts=T, Hs=H, Ws=W, Cs=Cout, Cn=Cin,
                       algo=hip.ALGO_PIPE256)
    assert hip.conv_plan(d8).startswith("nt8 ")
    o = torch.full((N, T, H, W, Cin), float("nan"), device=dev(), dtype=torch.float32)
    oh = torch.full((N, T, H, W, Cin), float("nan"), device=dev(), dtype=tdt)
    hip.conv_run(d8, dy, wd, None, o, O_lo=oh)
    assert rel_err(o, dy.double().cpu() @ wd.double().cpu().t()) < 1e-5
    assert torch.equal(oh, o.to(tdt))
    # not with a low residual term, not on fp32 operands
    with pytest.raises(hip.VlfbError):
        hip.conv_run(d8, dy, wd, None, o, R=o, R_lo=oh, O_lo=oh)
