"""DGRAD of the (1, 2, 2)-strided 1x3x3 convolutions (res3_0 / res4_0 branch2b, resnet_helper.py:35-119).  An input position only meets the taps of its own (h, w) parity, so the tiled kernel
enumerates its rows parity class by parity class and a tile walks only the taps of its class (csrc/vlfb_gemm.hip,
GP::s2) instead of multiplying structural zeros.  The taps that remain are walked in the order of the full walk, so
the result must be BIT-IDENTICAL to the plain enumeration (algo = TILE128), for every epilogue (residual, mask,
both, alpha), bf16 / fp16 / fp32, and match fp64 autograd."""
import math

import pytest
import torch
import torch.nn.functional as F

from gpu_util import dev, q, rel_err, to_ncthw, to_nthwc

pytestmark = pytest.mark.gpu

# name: (N, Cin, Cout, T, H, W, k, pad)      (stride (1, 2, 2) always; H, W even)
CASES = {
    "conv3x3": (2, 128, 128, 2, 12, 20, (1, 3, 3), (0, 1, 1)),
    "conv1x3": (1, 128, 64, 3, 10, 14, (1, 1, 3), (0, 0, 1)),            # one strided dimension with a single tap row
    "conv3x3_64": (1, 64, 64, 1, 34, 18, (1, 3, 3), (0, 1, 1)),          # several row tiles per class
    "temporal_3x3x3": (1, 64, 128, 4, 8, 8, (3, 3, 3), (1, 1, 1)),       # a temporal extent beside the strided dims
}


@pytest.mark.parametrize("tdt", [torch.bfloat16, torch.float16, torch.float32], ids=["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("case", sorted(CASES))
def test_class_major_strided_dgrad_is_bit_identical_and_matches_fp64(case, tdt):
    from vlfb import hip
    hip.lib()
    hdt = hip.dtype_code(tdt)
    tol = {torch.bfloat16: 1e-2, torch.float16: 2e-3, torch.float32: 2e-5}[tdt]
    N, Cin, Cout, T, H, W, k, p = CASES[case]
    s, d = (1, 2, 2), (1, 1, 1)
    gen = torch.Generator().manual_seed(sum(map(ord, case)))
    x = q(torch.randn(N, Cin, T, H, W, generator=gen), tdt)
    w = q(torch.randn(Cout, Cin, *k, generator=gen) / math.sqrt(Cin * k[0] * k[1] * k[2]), tdt)
    To, Ho, Wo = [(a + 2 * pp - (kk - 1) - 1) // ss + 1 for a, kk, ss, pp in zip((T, H, W), k, s, p)]
    dy = q(torch.randn(N, Cout, To, Ho, Wo, generator=gen), tdt)
    mask_src = q(torch.randn(N, Cin, T, H, W, generator=gen), tdt)
    add_src = q(torch.randn(N, Cin, T, H, W, generator=gen), tdt)
    G = to_nthwc(dy).to(dev(), tdt)
    Wd = w.permute(1, 2, 3, 4, 0).contiguous().to(dev(), tdt)
    Rm, Mm = to_nthwc(add_src).to(dev(), tdt), to_nthwc(mask_src).to(dev(), tdt)
    geom = dict(kt=k[0], kh=k[1], kw=k[2], st=1, sh=2, sw=2, pt=p[0], ph=p[1], pw=p[2], dt=1, dh=1, dw=1)
    view = torch.int16 if tdt != torch.float32 else torch.int32
    got = {}
    for algo in (hip.ALGO_TILE128, hip.ALGO_AUTO):
        outs = []
        for kw in (dict(R=Rm, mask=Mm), dict(R=Rm), dict(mask=Mm), dict()):
            DX = torch.full((N, T, H, W, Cin), float("nan"), device=dev(), dtype=tdt)
            desc = hip.conv_desc(mode=hip.DGRAD, dtype=hdt, out_dtype=hdt, N=N, Tr=T, Hr=H, Wr=W, Ts=To, Hs=Ho, Ws=Wo,
                                 Cs=Cout, Cn=Cin, alpha=0.5 if not kw else 1.0, algo=algo, **geom)
            hip.conv_run(desc, G, Wd, None, DX, **kw)
            torch.cuda.synchronize()
            outs.append(DX)
        got[algo] = outs
    for a, b in zip(got[hip.ALGO_AUTO], got[hip.ALGO_TILE128]):
        assert not torch.isnan(a.float()).any()
        assert torch.equal(a.view(view), b.view(view))
    xd = x.double().requires_grad_(True)
    gx, = torch.autograd.grad(F.conv3d(xd, w.double(), None, s, p, d), xd, dy.double())
    assert rel_err(to_ncthw(got[hip.ALGO_AUTO][0].float()), torch.where(mask_src.double() > 0, gx + add_src.double(), torch.zeros_like(gx))) < tol
    assert rel_err(to_ncthw(got[hip.ALGO_AUTO][3].float()), 0.5 * gx) < tol


SPLIT_CASES = dict(CASES)
SPLIT_CASES["conv1x1"] = (2, 256, 64, 2, 12, 20, (1, 1, 1), (0, 0, 0))           # three of the four classes have no tap at all
SPLIT_CASES["conv3x3_ragged"] = (1, 72, 128, 2, 18, 14, (1, 3, 3), (0, 1, 1))    # Cin not a multiple of the column tile


@pytest.mark.parametrize("case", sorted(SPLIT_CASES))
def test_class_major_strided_dgrad_split_bf16(case):
    """the same walk in the split-bf16 kernel (csrc/vlfb_gemm_split.hip, gemm_nt_sp_kernel<.., S2>: scalar tap cursor
    over the class's taps, fp32 storage, three bf16 products per product): bit-identical to the plain enumeration for
    every epilogue, planes of the result included, and within the three-term bar of fp64 autograd"""
    from vlfb import hip
    hip.lib()
    N, Cin, Cout, T, H, W, k, p = SPLIT_CASES[case]
    s, d = (1, 2, 2), (1, 1, 1)
    gen = torch.Generator().manual_seed(sum(map(ord, case)) + 7)
    x = torch.randn(N, Cin, T, H, W, generator=gen)
    w = torch.randn(Cout, Cin, *k, generator=gen) / math.sqrt(Cin * k[0] * k[1] * k[2])
    To, Ho, Wo = [(a + 2 * pp - (kk - 1) - 1) // ss + 1 for a, kk, ss, pp in zip((T, H, W), k, s, p)]
    taps = k[0] * k[1] * k[2]
    dy = torch.randn(N, Cout, To, Ho, Wo, generator=gen)
    mask_src = torch.randn(N, Cin, T, H, W, generator=gen)
    add_src = torch.randn(N, Cin, T, H, W, generator=gen)
    G = to_nthwc(dy).to(dev())
    wk = w.permute(0, 2, 3, 4, 1).contiguous().to(dev())
    Wf = torch.empty(3, Cout, taps, Cin, device=dev(), dtype=torch.bfloat16)
    Wd = torch.empty(2, Cin, taps, Cout, device=dev(), dtype=torch.bfloat16)
    hip.call("vlfb_weight_prep", hip.ptr(wk), None, hip.ptr(Wf), hip.ptr(Wd), hip.SPLIT, Cout, taps, Cin)
    Rm, Mm = to_nthwc(add_src).to(dev()), to_nthwc(mask_src).to(dev())
    geom = dict(kt=k[0], kh=k[1], kw=k[2], st=1, sh=2, sw=2, pt=p[0], ph=p[1], pw=p[2], dt=1, dh=1, dw=1)
    Mi = N * T * H * W
    got = {}
    for algo in (hip.ALGO_TILE128, hip.ALGO_CLASSES):       # (AUTO takes the class walk for kh * kw > 1 only)
        outs = []
        for kw in (dict(R=Rm, mask=Mm), dict(R=Rm), dict(mask=Mm), dict()):
            DX = torch.full((N, T, H, W, Cin), float("nan"), device=dev())
            DXp = torch.full((2, Mi, Cin), float("nan"), device=dev(), dtype=torch.bfloat16)
            desc = hip.conv_desc(mode=hip.DGRAD, dtype=hip.F32, out_dtype=hip.F32, N=N, Tr=T, Hr=H, Wr=W, Ts=To, Hs=Ho, Ws=Wo,
                                 Cs=Cout, Cn=Cin, alpha=0.5 if not kw else 1.0, algo=algo, math=hip.MATH_BF16X3,
                                 b_pstride=Cin * taps * Cout, o_planes=2, o_pstride=Mi * Cin, **geom)
            hip.conv_run(desc, G, Wd, None, DX, O_planes=DXp, **kw)
            torch.cuda.synchronize()
            outs.append((DX, DXp))
        got[algo] = outs
    for (a, ap), (b, bp) in zip(got[hip.ALGO_CLASSES], got[hip.ALGO_TILE128]):
        assert not torch.isnan(a).any() and not torch.isnan(ap.float()).any()
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
        assert torch.equal(ap.view(torch.int16), bp.view(torch.int16))
    xd = x.double().requires_grad_(True)
    gx, = torch.autograd.grad(F.conv3d(xd, w.double(), None, s, p, d), xd, dy.double())
    want = torch.where(mask_src.double() > 0, gx + add_src.double(), torch.zeros_like(gx))
    assert rel_err(to_ncthw(got[hip.ALGO_CLASSES][0][0]), want) < 2e-5
    assert rel_err(to_ncthw(got[hip.ALGO_CLASSES][3][0]), 0.5 * gx) < 2e-5


SHORTCUTS = {
    # name: (N, Cin, Cout, T, H, W)   1x1x1, stride (1, 2, 2), no padding: the projection shortcuts of res3_0 / res4_0
    "res3_like": (2, 256, 512, 2, 12, 20),
    "res4_like_ragged_tiles": (1, 192, 128, 3, 10, 14),      # the class's rows do not fill whole 128-row tiles
}


@pytest.mark.parametrize("two_term", [False, True], ids=["one_term", "two_term_weights"])
@pytest.mark.parametrize("tdt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("case", sorted(SHORTCUTS))
def test_strided_shortcut_dgrad_as_an_in_place_accumulate(case, tdt, two_term):
    """algo = CLASS0: the DGRAD of a strided 1x1x1 conv computes and writes ONLY the rows the conv reads (even h, even w) and
    leaves every other row of O as it was.  Run in place on a buffer that holds an earlier contribution (R = O, and R_lo =
    O_lo for a two-term gradient) it must give, bit for bit, what the ordinary launch gives with that contribution as its
    residual operand -- on every row: a row the conv does not read is `0 + R`, i.e. untouched.  Also with the doubled term
    dimension of two-term fp16 weights (kt = 2, dt = 0: the `mix` path's DGRAD form)."""
    from vlfb import hip
    hip.lib()
    hdt = hip.dtype_code(tdt)
    N, Cin, Cout, T, H, W = SHORTCUTS[case]
    Ho, Wo = H // 2, W // 2
    gen = torch.Generator().manual_seed(sum(map(ord, case)) + 3)
    w = torch.randn(Cout, Cin, generator=gen) / math.sqrt(Cin)
    dy = q(torch.randn(N, Cout, T, Ho, Wo, generator=gen), tdt)
    first = q(torch.randn(N, Cin, T, H, W, generator=gen), tdt)
    first_lo = q(torch.randn(N, Cin, T, H, W, generator=gen) * 2.0 ** -12, tdt)
    G = to_nthwc(dy).to(dev(), tdt)
    terms = 2 if two_term else 1
    if two_term:                                     # [Cin][term][Cout]: hi + lo of the weight
        wh = q(w, tdt)
        wl = q(w - wh.float(), tdt)
        Wd = torch.stack([wh.t(), wl.t()], 1).contiguous().to(dev(), tdt)
    else:
        Wd = q(w, tdt).t().contiguous().to(dev(), tdt)
    geom = dict(kt=terms, kh=1, kw=1, st=1, sh=2, sw=2, pt=0, ph=0, pw=0, dt=0 if two_term else 1, dh=1, dw=1)
    rows = dict(N=N, Tr=T, Hr=H, Wr=W, Ts=T, Hs=Ho, Ws=Wo)
    view = torch.int16
    full = hip.conv_desc(mode=hip.DGRAD, dtype=hdt, out_dtype=hdt, Cs=Cout, Cn=Cin, algo=hip.ALGO_TILE128, **rows, **geom)
    sparse = hip.conv_desc(mode=hip.DGRAD, dtype=hdt, out_dtype=hdt, Cs=Cout, Cn=Cin, algo=hip.ALGO_CLASS0, **rows, **geom)
    assert "class0" in hip.conv_plan(sparse)
    R = to_nthwc(first).to(dev(), tdt)
    R_lo = to_nthwc(first_lo).to(dev(), tdt)
    # (a) one-term gradient
    want = torch.full((N, T, H, W, Cin), float("nan"), device=dev(), dtype=tdt)
    hip.conv_run(full, G, Wd, None, want, R=R)
    got = R.clone()
    hip.conv_run(sparse, G, Wd, None, got, R=got)
    torch.cuda.synchronize()
    assert torch.equal(got.view(view), want.view(view))
    # rows the conv does not read: exactly the first contribution
    odd = torch.ones(H, W, dtype=torch.bool)
    odd[0::2, 0::2] = False
    assert torch.equal(got[:, :, odd].view(view), R[:, :, odd].view(view))
    # (b) two-term gradient: hi and lo in place
    want, want_lo = (torch.full((N, T, H, W, Cin), float("nan"), device=dev(), dtype=tdt) for _ in range(2))
    hip.conv_run(full, G, Wd, None, want, R=R, R_lo=R_lo, O_lo=want_lo)
    got, got_lo = R.clone(), R_lo.clone()
    hip.conv_run(sparse, G, Wd, None, got, R=got, R_lo=got_lo, O_lo=got_lo)
    torch.cuda.synchronize()
    ev = ~odd
    assert torch.equal(got[:, :, ev].view(view), want[:, :, ev].view(view))
    assert torch.equal(got_lo[:, :, ev].view(view), want_lo[:, :, ev].view(view))
    assert torch.equal(got[:, :, odd].view(view), R[:, :, odd].view(view)) and torch.equal(got_lo[:, :, odd].view(view), R_lo[:, :, odd].view(view))
    # and against fp64: the sum of both contributions
    wd = (w if not two_term else (wh.double() + wl.double())).double()
    xd = torch.zeros(N, Cin, T, H, W, dtype=torch.float64, requires_grad=True)
    gx, = torch.autograd.grad(F.conv3d(xd, wd.reshape(Cout, Cin, 1, 1, 1), None, (1, 2, 2)), xd, dy.double())
    tol = 1e-2 if tdt == torch.bfloat16 else 2e-3
    assert rel_err(to_ncthw(got.double() + got_lo.double()), gx + first.double() + first_lo.double()) < tol
