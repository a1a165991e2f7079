"""Split-bf16 math on fp32 storage (csrc/vlfb_gemm_split.hip; vlfb_conv_desc.math = BF16X6 / BF16X3) against fp64 torch:
every gather the convs of the model use (identity rows, scalar-tap cursor, per-lane decode for strided / dilated convs,
the packed stem), both tile widths, ragged M / Cn / K tiles, all epilogues, the batched attention products, the
pre-split weight planes of vlfb_weight_prep* / vlfb_split_planes.

Bars (relative L2; inputs are arbitrary fp32 values, NOT pre-rounded to bf16):
  BF16X6 (three bf16 terms per operand, 6 MFMAs per product)  2e-6: the fp32 grade the forward pass needs
  BF16X3 (two terms, 3 MFMAs)                                 4e-5: ~2^-17 per product, averaged over the contraction
and BF16X6 must be at least 8x closer to fp64 than BF16X3 on the same problem (it would not be if a term were dropped).
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
TOL6, TOL3 = 2e-6, 4e-5


def setup_module(module):
    from vlfb import hip as h
    module.hip = h
    h.lib()


def gpu(t):
    return t.to(dev())


def planes_of(w, n):
    """reference expansion: h = bf16(w), m = bf16(w - h), l = bf16(w - h - m) as a [n, ...] bf16 tensor"""
    out, r = [], w.clone().float()
    for _ in range(n):
        t = r.to(torch.bfloat16)
        out.append(t)
        r = r - t.float()
    return torch.stack(out)


def test_weight_prep_split_planes_are_the_bf16_expansion():
    gen = torch.Generator().manual_seed(1)
    cout, taps, cin = 40, 3, 72
    w = torch.randn(cout, taps, cin, generator=gen) * 0.07
    s = torch.rand(cout, generator=gen) + 0.5
    wf = torch.empty(3, cout, taps, cin, device=dev(), dtype=torch.bfloat16)
    wd = torch.empty(2, cin, taps, cout, device=dev(), dtype=torch.bfloat16)
    wg, sg = gpu(w), gpu(s)
    hip.call("vlfb_weight_prep", hip.ptr(wg), hip.ptr(sg), hip.ptr(wf), hip.ptr(wd), hip.SPLIT, cout, taps, cin)
    ws = w * s.view(-1, 1, 1)
    assert torch.equal(wf.cpu(), planes_of(ws, 3))
    assert torch.equal(wd.cpu(), planes_of(ws.permute(2, 1, 0).contiguous(), 2))
    # three terms carry the fp32 value to ~2^-24, two to ~2^-16
    assert rel_err(wf.float().sum(0), ws) < 1e-7 and 1e-7 < rel_err(wd.float().sum(0), ws.permute(2, 1, 0)) < 1e-5
    # activations: plain and transposed per batch element
    x = torch.randn(3, 10, 24, generator=gen)
    for npl in (2, 3):
        for tr in (0, 1):
            dst = torch.empty(npl, 3, 24 if tr else 10, 10 if tr else 24, device=dev(), dtype=torch.bfloat16)
            xg = gpu(x)
            hip.call("vlfb_split_planes", hip.ptr(xg), hip.ptr(dst), npl, 3, 10, 24, tr)
            assert torch.equal(dst.cpu(), planes_of(x.transpose(1, 2).contiguous() if tr else x, npl))


@pytest.mark.parametrize("case", sorted(CONV_CASES) + ["big_k3", "big_pw"])
def test_split_conv_fprop_dgrad_wgrad(case):
    cases = dict(CONV_CASES)
    cases["big_k3"] = (2, 128, 256, 3, 14, 14, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1))     # several row / column tiles, 36 k-tiles
    cases["big_pw"] = (1, 512, 136, 2, 15, 15, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))     # ragged column tile (136 = 128 + 8)
    N, Cin, Cout, T, H, W, k, s, p, d = cases[case]
    gen = torch.Generator().manual_seed(sum(map(ord, case)))
    x = torch.randn(N, Cin, T, H, W, generator=gen)
    w = torch.randn(Cout, Cin, *k, generator=gen) * (1.0 / math.sqrt(Cin * k[0] * k[1] * k[2]))
    To, Ho, Wo = conv_out_dims(T, H, W, k, s, p, d)
    bias = torch.randn(Cout, generator=gen)
    res = torch.randn(N, Cout, To, Ho, Wo, generator=gen)
    taps = k[0] * k[1] * k[2]
    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    y_lin = F.conv3d(xd, wd, None, s, p, d)
    y_ref = torch.relu(y_lin + bias.double().view(1, -1, 1, 1, 1) + res.double())

    A = gpu(to_nthwc(x))
    wk = gpu(w_to_kernel(w).contiguous())
    Wf = torch.empty(3, Cout, taps, Cin, device=dev(), dtype=torch.bfloat16)
    Wd = torch.empty(2, Cin, taps, Cout, device=dev(), dtype=torch.bfloat16)
    hip.call("vlfb_weight_prep", hip.ptr(wk), None, hip.ptr(Wf), hip.ptr(Wd), hip.SPLIT, Cout, taps, Cin)
    plane = Cout * taps * Cin
    errs = {}
    for math_, tol in ((hip.MATH_BF16X6, TOL6), (hip.MATH_BF16X3, TOL3)):
        O = torch.full((N, To, Ho, Wo, Cout), float("nan"), device=dev(), dtype=torch.float32)
        desc = hip.conv_desc(mode=hip.FPROP, dtype=hip.F32, out_dtype=hip.F32, N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T, Hs=H, Ws=W,
                             Cs=Cin, Cn=Cout, relu=1, bias_mode=hip.BIAS_COL, math=math_, b_pstride=plane, **geom_kwargs(k, s, p, d))
        hip.conv_run(desc, A, Wf, None, O, bias=gpu(bias), R=gpu(to_nthwc(res)))
        errs[math_] = rel_err(to_ncthw(O), y_ref)
        assert errs[math_] < tol, ("fprop", math_, errs[math_])
    assert errs[hip.MATH_BF16X6] * 8 < errs[hip.MATH_BF16X3], errs

    # ---- dgrad (+ accumulate into an existing gradient, + ReLU mask) ----
    dy = torch.randn(N, Cout, To, Ho, Wo, generator=gen)
    gx, gw = torch.autograd.grad(y_lin, (xd, wd), dy.double())
    mask_src = torch.randn(N, Cin, T, H, W, generator=gen)
    add_src = torch.randn(N, Cin, T, H, W, generator=gen)
    dx_ref = torch.where(mask_src.double() > 0, gx + add_src.double(), torch.zeros_like(gx))
    G = gpu(to_nthwc(dy))
    for math_, tol in ((hip.MATH_BF16X3, TOL3),):
        DX = torch.full((N, T, H, W, Cin), float("nan"), device=dev(), dtype=torch.float32)
        desc = hip.conv_desc(mode=hip.DGRAD, dtype=hip.F32, out_dtype=hip.F32, N=N, Tr=T, Hr=H, Wr=W, Ts=To, Hs=Ho, Ws=Wo,
                             Cs=Cout, Cn=Cin, math=math_, b_pstride=plane, **geom_kwargs(k, s, p, d))
        hip.conv_run(desc, G, Wd, None, DX, R=gpu(to_nthwc(add_src)), mask=gpu(to_nthwc(mask_src)))
        assert rel_err(to_ncthw(DX), dx_ref) < tol, ("dgrad", math_)

    # ---- wgrad: library-chosen split, direct, forced split; row scale; accumulate ----
    scale = torch.rand(Cout, generator=gen) + 0.5
    gw_ref = w_to_kernel(gw * scale.double().view(-1, 1, 1, 1, 1))
    for splits in (0, 1, 3):
        DW = torch.full((Cout, k[0], k[1], k[2], Cin), float("nan"), device=dev(), dtype=torch.float32)
        desc = hip.conv_desc(mode=hip.WGRAD, dtype=hip.F32, out_dtype=hip.F32, N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T, Hs=H, Ws=W,
                             Cs=Cin, Cn=Cout, splits=splits, math=hip.MATH_BF16X3, **geom_kwargs(k, s, p, d))
        ws = torch.empty(max(hip.conv_workspace_bytes(desc), 16) // 4, device=dev(), dtype=torch.float32)
        hip.conv_run(desc, A, None, G, DW, rowscale=gpu(scale), workspace=ws)
        assert rel_err(DW, gw_ref) < TOL3, "wgrad splits=%d" % splits
    base = torch.randn(Cout, k[0], k[1], k[2], Cin, generator=gen)
    DW = gpu(base.clone())
    desc = hip.conv_desc(mode=hip.WGRAD, dtype=hip.F32, out_dtype=hip.F32, N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T, Hs=H, Ws=W,
                         Cs=Cin, Cn=Cout, splits=2, accumulate=1, math=hip.MATH_BF16X3, **geom_kwargs(k, s, p, d))
    ws = torch.empty(max(hip.conv_workspace_bytes(desc), 16) // 4, device=dev(), dtype=torch.float32)
    hip.conv_run(desc, A, None, G, DW, workspace=ws)
    assert rel_err(DW, base.double() + w_to_kernel(gw)) < TOL3, "wgrad acc"


@pytest.mark.parametrize("hw", [(20, 20), (10, 224)])
def test_split_stem_packed(hw):
    """conv1 (resnet_video.py:169-179) in the packed [kw_pad = 8][4] layout: FPROP with six terms, WGRAD with three"""
    N, T, Cout = 2, 6, 64
    H, W = hw
    k, s, p, d = (5, 7, 7), (1, 2, 2), (2, 3, 3), (1, 1, 1)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(N, 3, T, H, W, generator=gen)
    w = torch.randn(Cout, 3, *k, generator=gen) * 0.05
    To, Ho, Wo = conv_out_dims(T, H, W, k, s, p, d)
    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    y_ref = F.conv3d(xd, wd, None, s, p, d)
    WP = W + 8
    X4 = torch.empty(N, T, H, WP, 4, device=dev(), dtype=torch.float32)
    xg = gpu(x)
    hip.call("vlfb_ncthw_to_nthwc_wpad", hip.ptr(xg), hip.ptr(X4), hip.F32, N, 3, T * H, W, 4, 4, WP)
    wp = torch.zeros(Cout, 5, 7, 8, 4)
    wp[:, :, :, :7, :3] = w.permute(0, 2, 3, 4, 1)
    wpg = gpu(wp)
    Wf = torch.empty(3, Cout, 35, 32, device=dev(), dtype=torch.bfloat16)
    hip.call("vlfb_weight_prep", hip.ptr(wpg), None, hip.ptr(Wf), None, hip.SPLIT, Cout, 35 * 8, 4)
    O = torch.empty(N, To, Ho, Wo, Cout, device=dev(), dtype=torch.float32)
    gk = geom_kwargs(k, s, p, d)
    gk["pw"] = p[2] - 4
    desc = hip.conv_desc(mode=hip.FPROP, dtype=hip.F32, out_dtype=hip.F32, N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T, Hs=H, Ws=WP, Cs=4,
                         Cn=Cout, pack_w=8, math=hip.MATH_BF16X6, b_pstride=Cout * 1120, **gk)
    hip.conv_run(desc, X4, Wf, None, O)
    assert rel_err(to_ncthw(O), y_ref) < TOL6
    dy = torch.randn(N, Cout, To, Ho, Wo, generator=gen)
    (gw,) = torch.autograd.grad(y_ref, (wd,), dy.double())
    G = gpu(to_nthwc(dy))
    DW = torch.empty(Cout, 5, 7, 8, 4, device=dev(), dtype=torch.float32)
    desc = hip.conv_desc(mode=hip.WGRAD, dtype=hip.F32, out_dtype=hip.F32, N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T, Hs=H, Ws=WP, Cs=4,
                         Cn=Cout, pack_w=8, math=hip.MATH_BF16X3, **gk)
    ws = torch.empty(max(hip.conv_workspace_bytes(desc), 16) // 4, device=dev(), dtype=torch.float32)
    hip.conv_run(desc, X4, None, G, DW, workspace=ws)
    assert rel_err(DW[:, :, :, :7, :3].permute(0, 4, 1, 2, 3), gw) < TOL3
    assert float(DW[..., 3].abs().max()) == 0.0
    # the same two launches on pre-split operands (the clip's and the output gradient's term planes from a split pass)
    nin, nout = X4.numel(), G.numel()
    Xp = torch.empty(3 * nin, device=dev(), dtype=torch.bfloat16)
    Gp = torch.empty(2 * nout, device=dev(), dtype=torch.bfloat16)
    hip.call("vlfb_split_planes", hip.ptr(X4), hip.ptr(Xp), 3, 1, nin // 8, 8, 0)
    hip.call("vlfb_split_planes", hip.ptr(G), hip.ptr(Gp), 2, 1, nout // 8, 8, 0)
    O2 = torch.empty_like(O)
    desc = hip.conv_desc(mode=hip.FPROP, dtype=hip.F32, out_dtype=hip.F32, N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T, Hs=H, Ws=WP, Cs=4,
                         Cn=Cout, pack_w=8, math=hip.MATH_BF16X6, b_pstride=Cout * 1120, a_planes=3, a_pstride=nin, **gk)
    hip.conv_run(desc, Xp, Wf, None, O2)
    assert rel_err(to_ncthw(O2), y_ref) < TOL6
    DW2 = torch.empty_like(DW)
    desc = hip.conv_desc(mode=hip.WGRAD, dtype=hip.F32, out_dtype=hip.F32, N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T, Hs=H, Ws=WP, Cs=4,
                         Cn=Cout, pack_w=8, math=hip.MATH_BF16X3, a_planes=3, a_pstride=nin, p_planes=2, p_pstride=nout, **gk)
    ws = torch.empty(max(hip.conv_workspace_bytes(desc), 16) // 4, device=dev(), dtype=torch.float32)
    hip.conv_run(desc, Xp, None, Gp, DW2, workspace=ws)
    assert rel_err(DW2[:, :, :, :7, :3].permute(0, 4, 1, 2, 3), gw) < TOL3
    assert float(DW2[..., 3].abs().max()) == 0.0


def test_split_batched_products_of_the_nonlocal_block():
    """theta.phi^T and P.g forward (six terms; K = L2 = 72 is 2.25 k-tiles), dY.g^T / dS.phi and the two
    contract-over-L1 products backward (three terms), B operands through vlfb_split_planes (nonlocal_helper.py:94-121)"""
    B, L1, L2, Ci = 3, 200, 72, 64
    gen = torch.Generator().manual_seed(5)
    theta = torch.randn(B, L1, Ci, generator=gen)
    phi = torch.randn(B, L2, Ci, generator=gen)
    g = torch.randn(B, L2, Ci, generator=gen)
    th, ph, gg = gpu(theta), gpu(phi), gpu(g)
    pl = torch.empty(3 * B * L2 * Ci, device=dev(), dtype=torch.bfloat16)
    gemm = lambda **kw: hip.conv_desc(mode=hip.FPROP, dtype=hip.F32, out_dtype=hip.F32, N=1, Tr=1, Hr=1, Wr=L1, Ts=1, Hs=1,
                                      Ws=L1, batch=B, b_pstride=B * L2 * Ci, **kw)
    S = torch.empty(B, L1, L2, device=dev(), dtype=torch.float32)
    hip.call("vlfb_split_planes", hip.ptr(ph), hip.ptr(pl), 3, B, L2, Ci, 0)
    hip.conv_run(gemm(Cs=Ci, Cn=L2, a_bstride=L1 * Ci, b_bstride=L2 * Ci, o_bstride=L1 * L2, math=hip.MATH_BF16X6), th, pl, None, S)
    S_ref = torch.einsum("blc,bmc->blm", theta.double(), phi.double())
    assert rel_err(S, S_ref) < TOL6
    P = torch.softmax(S_ref.float() * Ci ** -0.5, dim=2)
    Pg = gpu(P)
    Y = torch.empty(B, L1, Ci, device=dev(), dtype=torch.float32)
    hip.call("vlfb_split_planes", hip.ptr(gg), hip.ptr(pl), 3, B, L2, Ci, 1)
    hip.conv_run(gemm(Cs=L2, Cn=Ci, a_bstride=L1 * L2, b_bstride=Ci * L2, o_bstride=L1 * Ci, math=hip.MATH_BF16X6), Pg, pl, None, Y)
    assert rel_err(Y, torch.einsum("blm,bmc->blc", P.double(), g.double())) < TOL6
    # backward NT products
    dY = torch.randn(B, L1, Ci, generator=gen)
    dYg = gpu(dY)
    dP = torch.empty(B, L1, L2, device=dev(), dtype=torch.float32)
    hip.call("vlfb_split_planes", hip.ptr(gg), hip.ptr(pl), 2, B, L2, Ci, 0)
    hip.conv_run(gemm(Cs=Ci, Cn=L2, a_bstride=L1 * Ci, b_bstride=L2 * Ci, o_bstride=L1 * L2, math=hip.MATH_BF16X3), dYg, pl, None, dP)
    assert rel_err(dP, torch.einsum("blc,bmc->blm", dY.double(), g.double())) < TOL3
    dS = torch.randn(B, L1, L2, generator=gen)
    dSg = gpu(dS)
    dth = torch.empty(B, L1, Ci, device=dev(), dtype=torch.float32)
    hip.call("vlfb_split_planes", hip.ptr(ph), hip.ptr(pl), 2, B, L2, Ci, 1)
    hip.conv_run(gemm(Cs=L2, Cn=Ci, a_bstride=L1 * L2, b_bstride=Ci * L2, o_bstride=L1 * Ci, math=hip.MATH_BF16X3, alpha=0.5), dSg, pl, None, dth)
    assert rel_err(dth, 0.5 * torch.einsum("blm,bmc->blc", dS.double(), phi.double())) < TOL3
    # contract over L1 (TN, batched, direct epilogue)
    dphi = torch.empty(B, L2, Ci, device=dev(), dtype=torch.float32)
    desc = hip.conv_desc(mode=hip.WGRAD, dtype=hip.F32, out_dtype=hip.F32, N=1, Tr=1, Hr=1, Wr=L1, Ts=1, Hs=1, Ws=L1, Cs=Ci, Cn=L2,
                         batch=B, a_bstride=L1 * Ci, p_bstride=L1 * L2, o_bstride=L2 * Ci, splits=1, math=hip.MATH_BF16X3)
    hip.conv_run(desc, th, None, dSg, dphi)
    assert rel_err(dphi, torch.einsum("blm,blc->bmc", dS.double(), theta.double())) < TOL3


def test_split_math_is_refused_where_it_does_not_apply():
    d = hip.conv_desc(mode=hip.FPROP, dtype=hip.BF16, out_dtype=hip.BF16, N=1, Tr=1, Hr=1, Wr=64, Ts=1, Hs=1, Ws=64, Cs=64, Cn=64,
                      math=hip.MATH_BF16X6)
    t = torch.zeros(64 * 64 * 3, device=dev(), dtype=torch.bfloat16)
    with pytest.raises(hip.VlfbError):
        hip.conv_run(d, t, t, None, t)
    d = hip.conv_desc(mode=hip.WGRAD, dtype=hip.F32, out_dtype=hip.F32, N=1, Tr=1, Hr=1, Wr=64, Ts=1, Hs=1, Ws=64, Cs=64, Cn=64,
                      math=hip.MATH_BF16X6)
    f = torch.zeros(64 * 64, device=dev(), dtype=torch.float32)
    with pytest.raises(hip.VlfbError):
        hip.conv_run(d, f, None, f, f)


@pytest.mark.parametrize("case", ["pw_ident", "temporal3", "spatial3", "wide", "big_k3", "big_pw", "spatial3_s2"])
def test_split_products_on_presplit_operands(case):
    """Activations / gradients handed in as bf16 term planes (vlfb_conv_desc.a_planes / p_planes), the format an earlier
    launch wrote through o_planes: DGRAD and FPROP read them by DMA with no VALU in the k-loop, WGRAD takes both operands
    as planes (DMA + transposed LDS reads, three MFMAs per fragment pair).  Same bars as the in-kernel split, and the
    planes an epilogue writes are bit for bit the expansion of its fp32 output."""
    cases = dict(CONV_CASES)
    cases["big_k3"] = (2, 128, 256, 3, 14, 14, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1))
    cases["big_pw"] = (1, 512, 136, 2, 15, 15, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))
    N, Cin, Cout, T, H, W, k, s, p, d = cases[case]
    strided = s != (1, 1, 1)
    gen = torch.Generator().manual_seed(sum(map(ord, case)) + 1)
    x = torch.randn(N, Cin, T, H, W, generator=gen)
    w = torch.randn(Cout, Cin, *k, generator=gen) * (1.0 / math.sqrt(Cin * k[0] * k[1] * k[2]))
    To, Ho, Wo = conv_out_dims(T, H, W, k, s, p, d)
    taps = k[0] * k[1] * k[2]
    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    y_lin = F.conv3d(xd, wd, None, s, p, d)
    dy = torch.randn(N, Cout, To, Ho, Wo, generator=gen)
    gx, gw = torch.autograd.grad(y_lin, (xd, wd), dy.double())
    A = gpu(to_nthwc(x))
    G = gpu(to_nthwc(dy))
    wk = gpu(w_to_kernel(w).contiguous())
    Wf = torch.empty(3, Cout, taps, Cin, device=dev(), dtype=torch.bfloat16)
    Wd = torch.empty(2, Cin, taps, Cout, device=dev(), dtype=torch.bfloat16)
    hip.call("vlfb_weight_prep", hip.ptr(wk), None, hip.ptr(Wf), hip.ptr(Wd), hip.SPLIT, Cout, taps, Cin)
    plane = Cout * taps * Cin
    Mi, Mo = N * T * H * W, N * To * Ho * Wo
    Ap = torch.empty(3, Mi, Cin, device=dev(), dtype=torch.bfloat16)
    Gp = torch.empty(2, Mo, Cout, device=dev(), dtype=torch.bfloat16)
    hip.call("vlfb_split_planes", hip.ptr(A), hip.ptr(Ap), 3, 1, Mi, Cin, 0)
    hip.call("vlfb_split_planes", hip.ptr(G), hip.ptr(Gp), 2, 1, Mo, Cout, 0)
    g = geom_kwargs(k, s, p, d)
    # FPROP on a pre-split activation (three terms), writing the output's planes as well
    O = torch.full((N, To, Ho, Wo, Cout), float("nan"), device=dev(), dtype=torch.float32)
    Op = torch.full((2, Mo, Cout), float("nan"), device=dev(), dtype=torch.bfloat16)
    desc = hip.conv_desc(mode=hip.FPROP, dtype=hip.F32, out_dtype=hip.F32, N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T, Hs=H, Ws=W, Cs=Cin, Cn=Cout,
                         relu=1, math=hip.MATH_BF16X6, b_pstride=plane, a_planes=3, a_pstride=Mi * Cin, o_planes=2, o_pstride=Mo * Cout, **g)
    hip.conv_run(desc, Ap, Wf, None, O, O_planes=Op)
    assert rel_err(to_ncthw(O), torch.relu(y_lin)) < TOL6, "fprop on planes"
    assert torch.equal(Op.cpu(), planes_of(O.cpu().reshape(Mo, Cout), 2)), "o_planes must be the expansion of the fp32 output"
    # the in-kernel-split kernel writes the same planes
    Op2 = torch.empty_like(Op)
    O2 = torch.empty_like(O)
    desc = hip.conv_desc(mode=hip.FPROP, dtype=hip.F32, out_dtype=hip.F32, N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T, Hs=H, Ws=W, Cs=Cin, Cn=Cout,
                         relu=1, math=hip.MATH_BF16X6, b_pstride=plane, o_planes=2, o_pstride=Mo * Cout, **g)
    hip.conv_run(desc, A, Wf, None, O2, O_planes=Op2)
    assert torch.equal(Op2.cpu(), planes_of(O2.cpu().reshape(Mo, Cout), 2))
    # three-term FPROP (the engine's default forward math) on the first two planes, with a residual, against the same
    # launch on the fp32 operand: the two kernels form the same products in the same order
    Rs = gpu(torch.randn(N, To, Ho, Wo, Cout, generator=gen))
    O3, O4 = torch.full_like(O, float("nan")), torch.full_like(O, float("nan"))
    Op3 = torch.full_like(Op, float("nan"))
    common = dict(mode=hip.FPROP, dtype=hip.F32, out_dtype=hip.F32, N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T, Hs=H, Ws=W, Cs=Cin, Cn=Cout,
                  relu=1, math=hip.MATH_BF16X3, b_pstride=plane, **g)
    hip.conv_run(hip.conv_desc(a_planes=2, a_pstride=Mi * Cin, o_planes=2, o_pstride=Mo * Cout, **common), Ap, Wf, None, O3, R=Rs,
                 O_planes=Op3)
    hip.conv_run(hip.conv_desc(**common), A, Wf, None, O4, R=Rs)
    want = torch.relu(y_lin + Rs.cpu().double().permute(0, 4, 1, 2, 3))
    assert rel_err(to_ncthw(O3), want) < TOL3 and rel_err(to_ncthw(O4), want) < TOL3, "three-term fprop"
    assert torch.equal(Op3.cpu(), planes_of(O3.cpu().reshape(Mo, Cout), 2))
    # DGRAD on the pre-split gradient (two terms); strided convs have no scalar tap cursor and keep the fp32 operand
    if not strided:
        DX = torch.full((N, T, H, W, Cin), float("nan"), device=dev(), dtype=torch.float32)
        desc = hip.conv_desc(mode=hip.DGRAD, dtype=hip.F32, out_dtype=hip.F32, N=N, Tr=T, Hr=H, Wr=W, Ts=To, Hs=Ho, Ws=Wo, Cs=Cout, Cn=Cin,
                             math=hip.MATH_BF16X3, b_pstride=plane, a_planes=2, a_pstride=Mo * Cout, **g)
        hip.conv_run(desc, Gp, Wd, None, DX)
        assert rel_err(to_ncthw(DX), gx) < TOL3, "dgrad on planes"
    else:
        with pytest.raises(hip.VlfbError):
            desc = hip.conv_desc(mode=hip.DGRAD, dtype=hip.F32, out_dtype=hip.F32, N=N, Tr=T, Hr=H, Wr=W, Ts=To, Hs=Ho, Ws=Wo, Cs=Cout,
                                 Cn=Cin, math=hip.MATH_BF16X3, b_pstride=plane, a_planes=2, a_pstride=Mo * Cout, **g)
            hip.conv_run(desc, Gp, Wd, None, torch.empty(N, T, H, W, Cin, device=dev()))
    # WGRAD on both operands pre-split: library split, direct, forced split
    scale = torch.rand(Cout, generator=gen) + 0.5
    gw_ref = w_to_kernel(gw * scale.double().view(-1, 1, 1, 1, 1))
    for splits in (0, 1, 3):
        DW = torch.full((Cout, k[0], k[1], k[2], Cin), float("nan"), device=dev(), dtype=torch.float32)
        desc = hip.conv_desc(mode=hip.WGRAD, dtype=hip.F32, out_dtype=hip.F32, N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T, Hs=H, Ws=W, Cs=Cin, Cn=Cout,
                             splits=splits, math=hip.MATH_BF16X3, a_planes=3, a_pstride=Mi * Cin, p_planes=2, p_pstride=Mo * Cout, **g)
        ws = torch.empty(max(hip.conv_workspace_bytes(desc), 16) // 4, device=dev(), dtype=torch.float32)
        hip.conv_run(desc, Ap, None, Gp, DW, rowscale=gpu(scale), workspace=ws)
        assert rel_err(DW, gw_ref) < TOL3, "wgrad on planes, splits=%d" % splits


@pytest.mark.parametrize("M,K,Cn", [(33, 512, 512), (33, 2048, 512), (64, 512, 2048), (7, 128, 40), (1, 512, 520)])
def test_split_products_on_a_handful_of_rows(M, K, Cn):
    """gemm_skinny_nt_sp_kernel (csrc/vlfb_gemm_skinny.hip): the FBO convs of the fp32 head on one row per RoI -- plain fp32
    rows, two-term weight planes (BF16X3), FPROP with bias + residual + ReLU and DGRAD with a mask, against fp64 and against
    the 128 x 128 split kernel (the same terms, another accumulation order)"""
    gen = torch.Generator().manual_seed(M * 7 + K)
    x = torch.randn(M, K, generator=gen)
    w = torch.randn(Cn, K, generator=gen) * (1.0 / math.sqrt(K))
    bias = torch.randn(Cn, generator=gen)
    res = torch.randn(M, Cn, generator=gen)
    wp = torch.empty(3, Cn, K, device=dev(), dtype=torch.bfloat16)
    hip.call("vlfb_weight_prep", hip.ptr(gpu(w)), None, hip.ptr(wp), None, hip.SPLIT, Cn, 1, K)
    rows = dict(N=1, Tr=1, Hr=1, Wr=M, Ts=1, Hs=1, Ws=M, Cs=K, Cn=Cn)
    base = dict(mode=hip.FPROP, dtype=hip.F32, out_dtype=hip.F32, math=hip.MATH_BF16X3, b_pstride=Cn * K, relu=1, bias_mode=hip.BIAS_COL,
                alpha=0.5, **rows)
    d = hip.conv_desc(**base)
    d128 = hip.conv_desc(algo=hip.ALGO_TILE128, **base)
    assert hip.conv_plan(d).startswith("nt_skinny_split") and hip.conv_plan(d128).startswith("nt_split")
    xg, rg, bg = gpu(x), gpu(res), gpu(bias)
    o = torch.full((M, Cn), float("nan"), device=dev())
    o128 = torch.full((M, Cn), float("nan"), device=dev())
    hip.conv_run(d, xg, wp, None, o, bias=bg, R=rg)
    hip.conv_run(d128, xg, wp, None, o128, bias=bg, R=rg)
    ref = torch.relu(0.5 * (x.double() @ w.double().t()) + bias.double() + res.double())
    assert rel_err(o, ref) < TOL3 and rel_err(o, o128) < 2e-6
    # DGRAD form: rows of the output gradient, the [Cin][Cout] two-term copy, a ReLU mask on the input gradient
    wd = torch.empty(2, K, Cn, device=dev(), dtype=torch.bfloat16)          # (conv Cn -> K seen from the gradient: "Cout" = Cn)
    wt = torch.randn(Cn, K, generator=gen) * (1.0 / math.sqrt(Cn))          # conv weight [Cout = Cn][Cin = K]
    hip.call("vlfb_weight_prep", hip.ptr(gpu(wt)), None, None, hip.ptr(wd), hip.SPLIT, Cn, 1, K)
    dy = torch.randn(M, Cn, generator=gen)
    mask = torch.randn(M, K, generator=gen)
    dd = hip.conv_desc(mode=hip.DGRAD, dtype=hip.F32, out_dtype=hip.F32, math=hip.MATH_BF16X3, b_pstride=K * Cn,
                       N=1, Tr=1, Hr=1, Wr=M, Ts=1, Hs=1, Ws=M, Cs=Cn, Cn=K)
    ok_k = Cn % 32 == 0 and Cn >= 128
    assert hip.conv_plan(dd).startswith("nt_skinny_split" if ok_k else "nt_split")
    dx = torch.full((M, K), float("nan"), device=dev())
    hip.conv_run(dd, gpu(dy), wd, None, dx, mask=gpu(mask))
    ref = torch.where(mask.double() > 0, dy.double() @ wt.double(), torch.zeros(M, K, dtype=torch.float64))
    assert rel_err(dx, ref) < TOL3
    # the fp16 copy of the output (o_planes = 1: what the "mix" engine asks of its forward convs) comes out of the same launch
    dh = hip.conv_desc(o_planes=1, **base)
    assert hip.conv_plan(dh).startswith("nt_skinny_split")
    o2 = torch.full((M, Cn), float("nan"), device=dev())
    oh = torch.full((M, Cn), float("nan"), device=dev(), dtype=torch.float16)
    hip.conv_run(dh, xg, wp, None, o2, bias=bg, R=rg, O_planes=oh)
    assert torch.equal(o2, o)
    want = o.half()
    want = torch.where((o > 0) & (want == 0), torch.full_like(want, 2.0 ** -24), want)        # (positive stays positive: vlfb_half_copy)
    assert torch.equal(oh, want)
    # not with two-plane outputs (the tiled kernel carries that epilogue)
    assert hip.conv_plan(hip.conv_desc(**dict(base, out_dtype=hip.F16))).startswith("nt_split")
