"""Two-plane fp16 math (csrc/vlfb_gemm_pair.hip; vlfb_conv_desc.math = VLFB_MATH_F16X3): the forward contractions of the "mix"
path against fp64 torch -- plain rows, the scalar tap cursor (3x3, 3x1x1, strided, dilated), the packed stem, ragged
M / Cn tiles, residual + ReLU epilogues with two-plane and fp32 outputs -- plus the pieces around it: vlfb_pair_split /
vlfb_pair_join, the VLFB_MIXH weight planes, the two-plane pools, and the split-bf16 launch that writes two planes.

Bars (relative L2 against fp64 on arbitrary fp32 inputs): a stored value keeps ~22 bits (2^-22 per plane pair), a product
drops lo.lo (2^-22): 1.5e-6 for a whole conv -- 25x tighter than the three-term bf16 form (4e-5, tests/test_split_gpu.py).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import dev, rel_err, to_ncthw, to_nthwc, w_to_kernel
from test_kernels_gpu import CONV_CASES, conv_out_dims, geom_kwargs

pytestmark = pytest.mark.gpu

hip = None
TOL = 1.5e-6


def setup_module(module):
    from vlfb import hip as h
    module.hip = h
    h.lib()


def gpu(t):
    return t.to(dev())


def pair_ref(x):
    """reference expansion of fp32 values: hi = fp16(x), lo = fp16(x - hi)"""
    hi = x.float().half()
    lo = (x.float() - hi.float()).half()
    return torch.stack([hi, lo])


def pair_of(x):
    """two-plane device tensor [2][numel] of an fp32 CPU tensor, through vlfb_pair_split"""
    xg = gpu(x.contiguous().float())
    out = torch.empty(2 * xg.numel(), device=dev(), dtype=torch.float16)
    hip.call("vlfb_pair_split", hip.ptr(xg), hip.ptr(out), xg.numel())
    return out


def pair_value(t, shape):
    """fp32 CPU tensor of a two-plane device tensor, through vlfb_pair_join"""
    n = t.numel() // 2
    out = torch.empty(n, device=dev(), dtype=torch.float32)
    hip.call("vlfb_pair_join", hip.ptr(t), hip.ptr(out), n)
    return out.cpu().view(shape)


def test_pair_split_join_and_weight_planes():
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(5, 24, generator=gen) * torch.tensor([1e-6, 1e-3, 1.0, 50.0, 3000.0]).view(5, 1)
    p = pair_of(x)
    ref = pair_ref(x.view(-1))
    assert torch.equal(p.cpu().view(2, -1), ref)
    back = pair_value(p, x.shape)
    assert torch.equal(back, (ref[0].float() + ref[1].float()).view(x.shape))
    # 22 bits where the low term is a normal number, the 2^-24 absolute floor below
    big = x.abs() > 0.25
    assert ((back - x).abs()[big] <= x.abs()[big] * 2.0 ** -21).all()
    assert ((back - x).abs() <= torch.clamp(x.abs() * 2.0 ** -21, min=2.0 ** -24)).all()
    # weight planes: [2][Cout][taps][Cin] of (w * s) * 1024; DGRAD copy plain fp16 / two-term as VLFB_MIX / VLFB_MIX_W2
    cout, taps, cin = 40, 3, 72
    w = torch.randn(cout, taps, cin, generator=gen) * 0.07
    s = torch.rand(cout, generator=gen) + 0.5
    ws = (w * s.view(-1, 1, 1))
    for code, two in ((hip.MIXH, False), (hip.MIXH_W2, True)):
        wf = torch.empty(2, cout, taps, cin, device=dev(), dtype=torch.float16)
        wd = torch.empty((2 if two else 1) * cin * taps * cout, device=dev(), dtype=torch.float16)
        hip.call("vlfb_weight_prep", hip.ptr(gpu(w)), hip.ptr(gpu(s)), hip.ptr(wf), hip.ptr(wd), code, cout, taps, cin)
        assert torch.equal(wf.cpu(), pair_ref(ws * 1024.0))
        wt = ws.permute(2, 1, 0).contiguous()
        if two:
            assert torch.equal(wd.cpu().view(cin, 2, taps, cout), pair_ref(wt * 1024.0).permute(1, 0, 2, 3))
        else:
            assert torch.equal(wd.cpu().view(cin, taps, cout), wt.half())


PAIR_CASES = dict(CONV_CASES)
PAIR_CASES["big_k3"] = (2, 128, 256, 3, 14, 14, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1))     # several row / column tiles, 36 k-tiles
PAIR_CASES["big_pw"] = (1, 512, 136, 2, 15, 15, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))     # ragged column tile (136 = 128 + 8)
PAIR_CASES["thin_k"] = (1, 32, 256, 2, 9, 9, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))        # ONE k-tile
PAIR_CASES["t3_wide"] = (1, 256, 64, 4, 7, 7, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1))       # 64-column tiles, 24 k-tiles


@pytest.mark.parametrize("case", sorted(PAIR_CASES))
def test_pair_conv_fprop(case):
    N, Cin, Cout, T, H, W, k, s, p, d = PAIR_CASES[case]
    gen = torch.Generator().manual_seed(sum(map(ord, case)))
    x = torch.randn(N, Cin, T, H, W, generator=gen)
    taps = k[0] * k[1] * k[2]
    w = torch.randn(Cout, Cin, *k, generator=gen) * (1.0 / math.sqrt(Cin * taps))
    To, Ho, Wo = conv_out_dims(T, H, W, k, s, p, d)
    bias = torch.randn(Cout, generator=gen)
    res = torch.randn(N, Cout, To, Ho, Wo, generator=gen)
    y_lin = F.conv3d(x.double(), w.double(), None, s, p, d)
    y_ref = torch.relu(y_lin + bias.double().view(1, -1, 1, 1, 1) + res.double())

    A = pair_of(to_nthwc(x))
    Wf = torch.empty(2, Cout, taps, Cin, device=dev(), dtype=torch.float16)
    hip.call("vlfb_weight_prep", hip.ptr(gpu(w_to_kernel(w).contiguous())), None, hip.ptr(Wf), None, hip.MIXH, Cout, taps, Cin)
    R = pair_of(to_nthwc(res))
    n_out = N * To * Ho * Wo * Cout
    base = dict(mode=hip.FPROP, dtype=hip.F16, N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T, Hs=H, Ws=W, Cs=Cin, Cn=Cout, relu=1,
                bias_mode=hip.BIAS_COL, math=hip.MATH_F16X3, a_pstride=x.numel(), b_pstride=Cout * taps * Cin,
                alpha=1.0 / hip.MIX_W2_SCALE, **geom_kwargs(k, s, p, d))
    # two-plane output with a two-plane residual
    O = torch.full((2 * n_out,), float("nan"), device=dev(), dtype=torch.float16)
    desc = hip.conv_desc(out_dtype=hip.F16, **base)
    assert hip.conv_plan(desc).startswith(("nt_pair f16x3", "nt8_pair f16x3")), hip.conv_plan(desc)
    hip.conv_run(desc, A, Wf, None, O, bias=gpu(bias), R=R, R_lo=R[n_out:], O_lo=O[n_out:])
    got = to_ncthw(pair_value(O, (N, To, Ho, Wo, Cout)))
    assert rel_err(got, y_ref) < TOL, ("pair out", rel_err(got, y_ref))
    # the low plane is a rounding remainder of the high one (hi alone is what the fp16 backward reads): |lo| <= ulp(hi) / 2
    hi, lo = O.cpu().view(2, -1).float()
    assert (lo.abs() <= torch.clamp(hi.abs() * 2.0 ** -11, min=2.0 ** -25)).all()
    assert rel_err(to_ncthw(hi.view(N, To, Ho, Wo, Cout)), y_ref) < 4e-4
    # fp32 output, no residual; O_lo = the fp16 copy of the output (what the fp16 backward reads: positive stays positive)
    O32 = torch.full((N, To, Ho, Wo, Cout), float("nan"), device=dev(), dtype=torch.float32)
    Oh = torch.full((N, To, Ho, Wo, Cout), float("nan"), device=dev(), dtype=torch.float16)
    hip.conv_run(hip.conv_desc(out_dtype=hip.F32, **base), A, Wf, None, O32, bias=gpu(bias), O_lo=Oh)
    e32 = rel_err(to_ncthw(O32), torch.relu(y_lin + bias.double().view(1, -1, 1, 1, 1)))
    assert e32 < TOL, ("f32 out", e32)
    ref_h = torch.empty_like(Oh)
    hip.call("vlfb_half_copy", hip.ptr(O32), hip.ptr(ref_h), O32.numel())
    assert torch.equal(Oh.cpu(), ref_h.cpu())
    hip.conv_run(hip.conv_desc(out_dtype=hip.F32, **base), A, Wf, None, O32, bias=gpu(bias))      # (and without the copy)


def test_pair_stem_fprop():
    """conv1 (resnet_video.py:169-179): 3 -> 64 channels, 5x7x7, stride (1, 2, 2), as the packed kw-by-channel GEMM on a
    W-padded 4-channel clip whose two fp16 planes were made by vlfb_pair_split"""
    gen = torch.Generator().manual_seed(11)
    N, T, H, W, Cout = 1, 6, 20, 24, 64
    k, s, p = (5, 7, 7), (1, 2, 2), (2, 3, 3)
    x = torch.randn(N, 3, T, H, W, generator=gen)
    w = torch.randn(Cout, 3, *k, generator=gen) * 0.05
    bias = torch.randn(Cout, generator=gen)
    To, Ho, Wo = conv_out_dims(T, H, W, k, s, p, (1, 1, 1))
    y_ref = torch.relu(F.conv3d(x.double(), w.double(), None, s, p) + bias.double().view(1, -1, 1, 1, 1))
    wpad = 4
    xs = torch.zeros(N, T, H, W + 2 * wpad, 4)
    xs[:, :, :, wpad:wpad + W, :3] = x.permute(0, 2, 3, 4, 1)
    A = pair_of(xs)
    wk = torch.zeros(Cout, 5, 7, 8, 4)
    wk[:, :, :, :7, :3] = w.permute(0, 2, 3, 4, 1)
    Wf = torch.empty(2, Cout, 5 * 7, 32, device=dev(), dtype=torch.float16)
    hip.call("vlfb_weight_prep", hip.ptr(gpu(wk)), None, hip.ptr(Wf), None, hip.MIXH, Cout, 35, 32)
    n_out = N * To * Ho * Wo * Cout
    O = torch.full((2 * n_out,), float("nan"), device=dev(), dtype=torch.float16)
    desc = hip.conv_desc(mode=hip.FPROP, dtype=hip.F16, out_dtype=hip.F16, N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T, Hs=H, Ws=W + 2 * wpad, Cs=4, Cn=Cout,
                         pack_w=8, relu=1, bias_mode=hip.BIAS_COL, math=hip.MATH_F16X3, a_pstride=xs.numel(), b_pstride=Cout * 35 * 32,
                         alpha=1.0 / hip.MIX_W2_SCALE, kt=5, kh=7, kw=7, st=1, sh=2, sw=2, pt=2, ph=3, pw=3 - wpad, dt=1, dh=1, dw=1)
    hip.conv_run(desc, A, Wf, None, O, bias=gpu(bias), O_lo=O[n_out:])
    got = to_ncthw(pair_value(O, (N, To, Ho, Wo, Cout)))
    assert rel_err(got, y_ref) < TOL, rel_err(got, y_ref)


@pytest.mark.parametrize("shape", [(2, 5, 50), (1, 3, 224), (3, 2, 17)], ids=["cut_last_block", "full_frame", "two_blocks"])
@pytest.mark.parametrize("epi", [(True, True), (False, False)], ids=["bias_relu", "plain"])
def test_pair_stem_direct_kernel_is_bit_identical_to_the_tiled_kernel(shape, epi):
    """conv1 at the benchmarked width (112-wide output rows): the two-plane DIRECT convolution (csrc/vlfb_stem.hip,
    stem_fprop_pair_kernel: raw input rows and the tap's weights of both planes in one LDS stage) against the tiled two-plane
    kernel (algo = TILE128): same (a, b) order, same three MFMAs per accumulator and k-step, same epilogue -- both output
    planes bit for bit, with row blocks cut by the frame height and temporal padding on both sides; and against fp64"""
    N, T, H = shape
    relu, has_bias = epi
    W, Cout, wpad = 224, 64, 4
    gen = torch.Generator().manual_seed(17)
    x = torch.randn(N, 3, T, H, W, generator=gen)
    w = torch.randn(Cout, 3, 5, 7, 7, generator=gen) * 0.05
    bias = torch.randn(Cout, generator=gen) * 0.3
    xs = torch.zeros(N, T, H, W + 2 * wpad, 4)
    xs[:, :, :, wpad:wpad + W, :3] = x.permute(0, 2, 3, 4, 1)
    A = pair_of(xs)
    wk = torch.zeros(Cout, 5, 7, 8, 4)
    wk[:, :, :, :7, :3] = w.permute(0, 2, 3, 4, 1)
    Wf = torch.empty(2, Cout, 35, 32, device=dev(), dtype=torch.float16)
    hip.call("vlfb_weight_prep", hip.ptr(gpu(wk)), None, hip.ptr(Wf), None, hip.MIXH, Cout, 35, 32)
    Ho, Wo = (H + 6 - 7) // 2 + 1, 112
    n_out = N * T * Ho * Wo * Cout
    outs = {}
    for algo in (hip.ALGO_AUTO, hip.ALGO_TILE128):
        desc = hip.conv_desc(mode=hip.FPROP, dtype=hip.F16, out_dtype=hip.F16, N=N, Tr=T, Hr=Ho, Wr=Wo, Ts=T, Hs=H, Ws=W + 2 * wpad, Cs=4,
                             Cn=Cout, pack_w=8, relu=int(relu), bias_mode=hip.BIAS_COL if has_bias else hip.BIAS_NONE, math=hip.MATH_F16X3,
                             a_pstride=xs.numel(), b_pstride=Cout * 35 * 32, alpha=0.75 / hip.MIX_W2_SCALE, algo=algo,
                             kt=5, kh=7, kw=7, st=1, sh=2, sw=2, pt=2, ph=3, pw=3 - wpad, dt=1, dh=1, dw=1)
        assert hip.conv_plan(desc).startswith("stem_fprop_pair" if algo == hip.ALGO_AUTO else "nt_pair"), hip.conv_plan(desc)
        O = torch.full((2 * n_out,), float("nan"), device=dev(), dtype=torch.float16)
        hip.conv_run(desc, A, Wf, None, O, bias=gpu(bias) if has_bias else None, O_lo=O[n_out:])
        torch.cuda.synchronize()
        outs[algo] = O
    a, t = outs[hip.ALGO_AUTO], outs[hip.ALGO_TILE128]
    assert not torch.isnan(a.float()).any()
    assert torch.equal(a.view(torch.int16), t.view(torch.int16))
    ref = F.conv3d(x.double(), w.double(), None, (1, 2, 2), (2, 3, 3)) * 0.75
    if has_bias:
        ref = ref + bias.double().view(1, -1, 1, 1, 1)
    if relu:
        ref = torch.relu(ref)
        assert float((a[:n_out].float() == 0).float().mean()) > 0.2
    got = to_ncthw(pair_value(a, (N, T, Ho, Wo, Cout)))
    assert rel_err(got, ref) < TOL, rel_err(got, ref)


def test_split_launch_with_two_plane_output():
    """the `out` conv of a non-local block (nonlocal_helper.py:123-160): fp32 attention output in (split-bf16 products), the
    block input added as a two-plane residual, a two-plane output"""
    gen = torch.Generator().manual_seed(5)
    M, Cin, Cout = 300, 64, 136
    x = torch.randn(M, Cin, generator=gen)
    w = torch.randn(Cout, Cin, generator=gen) / math.sqrt(Cin)
    res = torch.randn(M, Cout, generator=gen)
    bias = torch.randn(Cout, generator=gen)
    y_ref = x.double() @ w.double().t() + bias.double() + res.double()
    Wf = torch.empty(3, Cout, 1, Cin, device=dev(), dtype=torch.bfloat16)
    hip.call("vlfb_weight_prep", hip.ptr(gpu(w)), None, hip.ptr(Wf), None, hip.SPLIT, Cout, 1, Cin)
    R = pair_of(res)
    O = torch.full((2 * M * Cout,), float("nan"), device=dev(), dtype=torch.float16)
    desc = hip.conv_desc(mode=hip.FPROP, dtype=hip.F32, out_dtype=hip.F16, N=1, Tr=1, Hr=1, Wr=M, Ts=1, Hs=1, Ws=M, Cs=Cin, Cn=Cout,
                         bias_mode=hip.BIAS_COL, math=hip.MATH_BF16X3, b_pstride=Cout * Cin)
    hip.conv_run(desc, gpu(x), Wf, None, O, bias=gpu(bias), R=R, R_lo=R[M * Cout:], O_lo=O[M * Cout:])
    got = pair_value(O, (M, Cout))
    assert rel_err(got, y_ref) < 4e-5, rel_err(got, y_ref)
    hi, lo = O.cpu().view(2, -1).float()
    assert (lo.abs() <= torch.clamp(hi.abs() * 2.0 ** -11, min=2.0 ** -25)).all()


@pytest.mark.parametrize("case", ["pw", "k133", "k311_s2", "ragged"])
def test_pair_256_row_pipelined_kernel_is_bit_identical_to_the_128_row_kernel(case):
    """vlfb_gemm_nt8.h PAIR (algo = PIPE256; the planner takes it for K >= 512): same products, same accumulation order"""
    N, Cin, Cout, T, H, W, k, s, p, d = {
        "pw": (2, 512, 256, 4, 14, 14, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),
        "k133": (2, 128, 256, 3, 14, 14, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),
        "k311_s2": (1, 64, 128, 6, 30, 30, (3, 1, 1), (1, 2, 2), (1, 0, 0), (1, 1, 1)),
        "ragged": (1, 160, 136, 2, 25, 25, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))}[case]
    gen = torch.Generator().manual_seed(sum(map(ord, case)))
    x = torch.randn(N, Cin, T, H, W, generator=gen)
    taps = k[0] * k[1] * k[2]
    w = torch.randn(Cout, Cin, *k, generator=gen) * (1.0 / math.sqrt(Cin * taps))
    To, Ho, Wo = conv_out_dims(T, H, W, k, s, p, d)
    bias = torch.randn(Cout, generator=gen)
    res = torch.randn(N, Cout, To, Ho, Wo, generator=gen)
    y_ref = torch.relu(F.conv3d(x.double(), w.double(), None, s, p, d) + bias.double().view(1, -1, 1, 1, 1) + res.double())
    A = pair_of(to_nthwc(x))
    Wf = torch.empty(2, Cout, taps, Cin, device=dev(), dtype=torch.float16)
    hip.call("vlfb_weight_prep", hip.ptr(gpu(w_to_kernel(w).contiguous())), None, hip.ptr(Wf), None, hip.MIXH, Cout, taps, Cin)
    R = pair_of(to_nthwc(res))
    n_out = N * To * Ho * Wo * Cout
    base = dict(mode=hip.FPROP, dtype=hip.F16, out_dtype=hip.F16, N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T, Hs=H, Ws=W, Cs=Cin, Cn=Cout, relu=1,
                bias_mode=hip.BIAS_COL, math=hip.MATH_F16X3, a_pstride=x.numel(), b_pstride=Cout * taps * Cin,
                alpha=1.0 / hip.MIX_W2_SCALE, **geom_kwargs(k, s, p, d))
    outs = {}
    for algo in (hip.ALGO_TILE128, hip.ALGO_PIPE256):
        O = torch.full((2 * n_out,), float("nan"), device=dev(), dtype=torch.float16)
        desc = hip.conv_desc(algo=algo, **base)
        assert hip.conv_plan(desc).startswith("nt8_pair" if algo == hip.ALGO_PIPE256 else "nt_pair"), hip.conv_plan(desc)
        hip.conv_run(desc, A, Wf, None, O, bias=gpu(bias), R=R, R_lo=R[n_out:], O_lo=O[n_out:])
        outs[algo] = O
    assert torch.equal(outs[hip.ALGO_TILE128], outs[hip.ALGO_PIPE256])
    got = to_ncthw(pair_value(outs[hip.ALGO_PIPE256], (N, To, Ho, Wo, Cout)))
    assert rel_err(got, y_ref) < TOL
    # fp32 output + fp16 copy
    O32 = torch.full((n_out,), float("nan"), device=dev(), dtype=torch.float32)
    Oh = torch.full((n_out,), float("nan"), device=dev(), dtype=torch.float16)
    d32 = dict(base)
    d32["out_dtype"] = hip.F32
    hip.conv_run(hip.conv_desc(algo=hip.ALGO_PIPE256, **d32), A, Wf, None, O32, bias=gpu(bias), O_lo=Oh)
    O32b = torch.full((n_out,), float("nan"), device=dev(), dtype=torch.float32)
    hip.conv_run(hip.conv_desc(algo=hip.ALGO_TILE128, **d32), A, Wf, None, O32b, bias=gpu(bias))
    assert torch.equal(O32, O32b)
    ref_h = torch.empty_like(Oh)
    hip.call("vlfb_half_copy", hip.ptr(O32), hip.ptr(ref_h), n_out)
    assert torch.equal(Oh, ref_h)


@pytest.mark.parametrize("geom", [((1, 3, 3), (1, 2, 2), (0, 1, 1)), ((2, 1, 1), (2, 1, 1), (0, 0, 0)), ((1, 2, 2), (1, 2, 2), (0, 0, 0))])
def test_pair_maxpool(geom):
    k, s, p = geom
    gen = torch.Generator().manual_seed(7)
    N, C, T, H, W = 2, 16, 4, 10, 10
    x = torch.randn(N, C, T, H, W, generator=gen)
    # make ties in the hi plane that only the lo plane resolves
    x[:, :, :, ::2, :] = (x[:, :, :, ::2, :].half().float() + 1e-5 * torch.rand(N, C, T, H // 2, W, generator=gen))
    To, Ho, Wo = [(n + 2 * pp - kk) // ss + 1 for n, kk, ss, pp in zip((T, H, W), k, s, p)]
    X = pair_of(to_nthwc(x))
    xv = pair_value(X, (N, T, H, W, C))               # what the kernel sees
    ref, idx = F.max_pool3d(to_ncthw(xv).double(), k, s, p, return_indices=True)
    Y = torch.full((2 * N * To * Ho * Wo * C,), float("nan"), device=dev(), dtype=torch.float16)
    desc = hip.pool_desc(hip.F16PAIR, N, T, H, W, C, To, Ho, Wo, k, s, p)
    am = torch.zeros(N * To * Ho * Wo * C, device=dev(), dtype=torch.uint8)
    hip.call("vlfb_maxpool_fwd", hip.C.byref(desc), hip.ptr(X), hip.ptr(Y), hip.ptr(am))
    got = to_ncthw(pair_value(Y, (N, To, Ho, Wo, C)))
    assert torch.equal(got.double(), ref)             # the pooled value IS an input value: exact
    # arg-max tap = the window element torch selected
    amc = am.cpu().view(N, To, Ho, Wo, C).permute(0, 4, 1, 2, 3).long()
    a, r = amc // (k[1] * k[2]), amc % (k[1] * k[2])
    b, c = r // k[2], r % k[2]
    to, ho, wo = torch.meshgrid(torch.arange(To), torch.arange(Ho), torch.arange(Wo), indexing="ij")
    lin = ((to * s[0] - p[0] + a) * H + (ho * s[1] - p[1] + b)) * W + (wo * s[2] - p[2] + c)
    assert torch.equal(lin, idx)


def test_pair_avgpool():
    gen = torch.Generator().manual_seed(9)
    N, C, T, H, W = 2, 32, 4, 7, 7
    x = torch.randn(N, C, T, H, W, generator=gen)
    X = pair_of(to_nthwc(x))
    xv = to_ncthw(pair_value(X, (N, T, H, W, C))).double()
    for k, out in (((T, 1, 1), (1, H, W)), ((T, H, W), (1, 1, 1))):
        Y = torch.full((N,) + out + (C,), float("nan"), device=dev(), dtype=torch.float32)
        desc = hip.pool_desc(hip.F16PAIR, N, T, H, W, C, out[0], out[1], out[2], k, (1, 1, 1), (0, 0, 0))
        hip.call("vlfb_avgpool_fwd", hip.C.byref(desc), hip.ptr(X), hip.ptr(Y))
        ref = F.avg_pool3d(xv, k, (1, 1, 1))
        assert rel_err(to_ncthw(Y), ref) < 3e-7
