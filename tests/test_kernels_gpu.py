"""GPU parity tests of every C-ABI kernel against plain torch-CPU restatements of the same op.

All calls go through libvlfb_hip.so (ctypes); inputs are rounded through the kernel's element type
first so the comparison measures the kernel, not the input quantisation.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import DTYPES, TOL, dev, q, rel_err, to_ncthw, to_nthwc, w_to_kernel

pytestmark = pytest.mark.gpu

hip = None


def setup_module(module):
    from vlfb import hip as h
    module.hip = h
    h.lib()


_KEEP = []


def gpu(t, dtype=None):
    t = t.to(dev())
    return t.to(dtype) if dtype is not None else t


def gp(t, dtype=None):
    """device copy of `t` kept alive until the end of the test; returns its device pointer.
    (`gp(x)` inline would free the temporary before the kernel is even launched.)"""
    g = gpu(t, dtype)
    _KEEP.append(g)
    return g.data_ptr()


@pytest.fixture(autouse=True)
def _release_kept_tensors():
    yield
    torch.cuda.synchronize()
    del _KEEP[:]


# ------------------------------------------------------------------------------------------------
def test_affine_nd_matches_reference_formula():
    g = torch.Generator().manual_seed(0)
    for shape in [(2, 5, 3, 4, 4), (1, 7, 1, 1, 1), (3, 64, 4, 7, 7)]:
        x = torch.randn(shape, generator=g)
        s = torch.rand(shape[1], generator=g) + 0.5
        b = torch.randn(shape[1], generator=g)
        n, c = shape[0], shape[1]
        inner = int(np.prod(shape[2:]))
        xg, sg, bg = gpu(x), gpu(s), gpu(b)
        y = torch.empty_like(xg)
        hip.call("vlfb_affine_nd_fwd", hip.ptr(xg), hip.ptr(sg), hip.ptr(bg), hip.ptr(y), n, c, inner)
        ref = x * s.view(1, -1, 1, 1, 1) + b.view(1, -1, 1, 1, 1)
        assert torch.equal(y.cpu(), ref) or rel_err(y, ref) < 1e-7
        dx = torch.empty_like(xg)
        hip.call("vlfb_affine_nd_bwd", hip.ptr(xg), hip.ptr(sg), hip.ptr(dx), n, c, inner)
        assert rel_err(dx, x * s.view(1, -1, 1, 1, 1)) < 1e-7
        # in place, as the schema allows (affine_nd_op.cc:35-43)
        hip.call("vlfb_affine_nd_fwd", hip.ptr(xg), hip.ptr(sg), hip.ptr(bg), hip.ptr(xg), n, c, inner)
        assert rel_err(xg, ref) < 1e-7


# ------------------------------------------------------------------------------------------------
CONV_CASES = {
    # name: (N, Cin, Cout, T, H, W, k, stride, pad, dil)
    "pw_ident": (2, 64, 128, 2, 9, 9, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),
    "pw_small_cout": (1, 128, 64, 2, 7, 7, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),
    "temporal3": (2, 64, 64, 5, 6, 6, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),
    "spatial3": (2, 64, 64, 2, 10, 10, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),
    "spatial3_s2": (2, 64, 128, 2, 12, 12, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 1, 1)),
    "spatial3_dil2": (1, 64, 64, 2, 9, 9, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2)),
    "shortcut_s2": (2, 64, 128, 2, 10, 10, (1, 1, 1), (1, 2, 2), (0, 0, 0), (1, 1, 1)),
    "wide": (1, 256, 192, 2, 8, 8, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),
}


def conv_out_dims(T, H, W, k, s, p, d):
    return tuple((x + 2 * pp - dd * (kk - 1) - 1) // ss + 1 for x, kk, ss, pp, dd in zip((T, H, W), k, s, p, d))


def geom_kwargs(k, s, p, d):
    return dict(kt=k[0], kh=k[1], kw=k[2], st=s[0], sh=s[1], sw=s[2], pt=p[0], ph=p[1], pw=p[2],
                dt=d[0], dh=d[1], dw=d[2])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", sorted(CONV_CASES))
def test_conv_fprop_dgrad_wgrad(case, dtype):
    N, Cin, Cout, T, H, W, k, s, p, d = CONV_CASES[case]
    gen = torch.Generator().manual_seed(sum(map(ord, case)))
    x = q(torch.randn(N, Cin, T, H, W, generator=gen), dtype)
    w = q(torch.randn(Cout, Cin, *k, generator=gen) * (1.0 / math.sqrt(Cin * k[0] * k[1] * k[2])), dtype)
    To, Ho, Wo = conv_out_dims(T, H, W, k, s, p, d)
    bias = torch.randn(Cout, generator=gen)
    res = q(torch.randn(N, Cout, To, Ho, Wo, generator=gen), dtype)
    code = hip.dtype_code(dtype)

    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    y_lin = F.conv3d(xd, wd, None, s, p, d)
    y_ref = torch.relu(y_lin + bias.double().view(1, -1, 1, 1, 1) + res.double())

    A = gpu(to_nthwc(x), dtype)
    Bw = gpu(w_to_kernel(w), dtype)
    O = torch.empty(N, To, Ho, Wo, Cout, device=dev(), dtype=dtype)
    desc = hip.conv_desc(mode=hip.FPROP, dtype=code, out_dtype=code, N=N, Tr=To, Hr=Ho, Wr=Wo,
                         Ts=T, Hs=H, Ws=W, Cs=Cin, Cn=Cout, relu=1, bias_mode=hip.BIAS_COL,
                         **geom_kwargs(k, s, p, d))
    hip.conv_run(desc, A, Bw, None, O, bias=gpu(bias), R=gpu(to_nthwc(res), dtype))
    assert rel_err(to_ncthw(O.float()), y_ref) < TOL[dtype], "fprop"

    # mask epilogue + fp32 output
    if dtype == torch.bfloat16:
        O32 = torch.empty(N, To, Ho, Wo, Cout, device=dev(), dtype=torch.float32)
        desc2 = hip.conv_desc(mode=hip.FPROP, dtype=code, out_dtype=hip.F32, N=N, Tr=To, Hr=Ho, Wr=Wo,
                              Ts=T, Hs=H, Ws=W, Cs=Cin, Cn=Cout, **geom_kwargs(k, s, p, d))
        hip.conv_run(desc2, A, Bw, None, O32)
        assert rel_err(to_ncthw(O32), y_lin) < 2e-5, "fprop fp32-out"

    # ---- dgrad -------------------------------------------------------------------------------
    dy = q(torch.randn(N, Cout, To, Ho, Wo, generator=gen), dtype)
    gx, gw = torch.autograd.grad(y_lin, (xd, wd), dy.double())
    mask_src = q(torch.randn(N, Cin, T, H, W, generator=gen), dtype)
    add_src = q(torch.randn(N, Cin, T, H, W, generator=gen), dtype)
    dx_ref = torch.where(mask_src.double() > 0, gx + add_src.double(), torch.zeros_like(gx))
    G = gpu(to_nthwc(dy), dtype)
    Wd = gpu(w.permute(1, 2, 3, 4, 0).contiguous(), dtype)  # [Cin][taps][Cout]
    DX = torch.empty(N, T, H, W, Cin, device=dev(), dtype=dtype)
    desc = hip.conv_desc(mode=hip.DGRAD, dtype=code, out_dtype=code, N=N, Tr=T, Hr=H, Wr=W,
                         Ts=To, Hs=Ho, Ws=Wo, Cs=Cout, Cn=Cin, **geom_kwargs(k, s, p, d))
    hip.conv_run(desc, G, Wd, None, DX, R=gpu(to_nthwc(add_src), dtype), mask=gpu(to_nthwc(mask_src), dtype))
    assert rel_err(to_ncthw(DX.float()), dx_ref) < TOL[dtype], "dgrad"

    # ---- wgrad (library-chosen split, forced split, and direct) --------------------------------
    scale = torch.rand(Cout, generator=gen) + 0.5
    gw_ref = w_to_kernel(gw * scale.double().view(-1, 1, 1, 1, 1))
    for splits in (0, 1, 3):
        DW = torch.full((Cout, k[0], k[1], k[2], Cin), float("nan"), device=dev(), dtype=torch.float32)
        desc = hip.conv_desc(mode=hip.WGRAD, dtype=code, out_dtype=hip.F32, N=N, Tr=To, Hr=Ho, Wr=Wo,
                             Ts=T, Hs=H, Ws=W, Cs=Cin, Cn=Cout, splits=splits, **geom_kwargs(k, s, p, d))
        nbytes = hip.conv_workspace_bytes(desc)
        ws = torch.empty(max(nbytes, 16) // 4, device=dev(), dtype=torch.float32)
        hip.conv_run(desc, A, None, G, DW, rowscale=gpu(scale), workspace=ws)
        assert rel_err(DW, gw_ref) < (2e-5 if dtype == torch.float32 else 2e-3), "wgrad splits=%d" % splits
    # accumulate
    base = torch.randn(Cout, k[0], k[1], k[2], Cin, generator=gen)
    DW = gpu(base.clone())
    desc = hip.conv_desc(mode=hip.WGRAD, dtype=code, out_dtype=hip.F32, N=N, Tr=To, Hr=Ho, Wr=Wo,
                         Ts=T, Hs=H, Ws=W, Cs=Cin, Cn=Cout, splits=2, accumulate=1, **geom_kwargs(k, s, p, d))
    ws = torch.empty(max(hip.conv_workspace_bytes(desc), 16) // 4, device=dev(), dtype=torch.float32)
    hip.conv_run(desc, A, None, G, DW, workspace=ws)
    assert rel_err(DW, base.double() + w_to_kernel(gw)) < (2e-5 if dtype == torch.float32 else 2e-3), "wgrad acc"


@pytest.mark.parametrize("hw", [(20, 20), (12, 80), (10, 224)])
@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_stem_packed(dtype, hw):
    """conv1: 5x7x7, stride (1,2,2), pad (2,3,3), Cin=3 packed as [kw_pad=8][4] runs (resnet_video.py:169-179).
    Output rows of 10 positions use the generic kernels, rows of 40 / 112 the whole-row bf16 stem
    wgrad kernel (partial last k-step: positions 32..39 / 96..111 live, the rest masked)."""
    N, T, Cout = 2, 6, 64
    H, W = hw
    k, s, p, d = (5, 7, 7), (1, 2, 2), (2, 3, 3), (1, 1, 1)
    gen = torch.Generator().manual_seed(3)
    x = q(torch.randn(N, 3, T, H, W, generator=gen), dtype)
    w = q(torch.randn(Cout, 3, *k, generator=gen) * 0.05, dtype)
    To, Ho, Wo = conv_out_dims(T, H, W, k, s, p, d)
    code = hip.dtype_code(dtype)
    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    y_ref = F.conv3d(xd, wd, None, s, p, d)
    # [N][T][H][W + 8][4]: C padded 3 -> 4 and 4 zero pixels on both sides of every row, via the
    # library's own mover (the packed stem kernels rely on that padding instead of w-bounds tests)
    WP = W + 8
    X4 = torch.empty(N, T, H, WP, 4, device=dev(), dtype=dtype)
    hip.call("vlfb_ncthw_to_nthwc_wpad", gp(x), hip.ptr(X4), code, N, 3, T * H, W, 4, 4, WP)
    assert torch.equal(X4[:, :, :, 4:4 + W, :3].float().cpu(), to_nthwc(x))
    assert float(X4[..., 3].abs().max()) == 0.0 and float(X4[:, :, :, :4].abs().max()) == 0.0 \
        and float(X4[:, :, :, 4 + W:].abs().max()) == 0.0
    wp = torch.zeros(Cout, 5, 7, 8, 4)
    wp[:, :, :, :7, :3] = w.permute(0, 2, 3, 4, 1)
    Bw = gpu(wp, dtype)
    O = torch.empty(N, To, Ho, Wo, Cout, device=dev(), dtype=dtype)
    gk = geom_kwargs(k, s, p, d)
    gk["pw"] = p[2] - 4
    desc = hip.conv_desc(mode=hip.FPROP, dtype=code, out_dtype=code, N=N, Tr=To, Hr=Ho, Wr=Wo,
                         Ts=T, Hs=H, Ws=WP, Cs=4, Cn=Cout, pack_w=8, **gk)
    hip.conv_run(desc, X4, Bw, None, O)
    assert rel_err(to_ncthw(O.float()), y_ref) < TOL[dtype]
    # wgrad in the packed layout
    dy = q(torch.randn(N, Cout, To, Ho, Wo, generator=gen), dtype)
    (gw,) = torch.autograd.grad(y_ref, (wd,), dy.double())
    G = gpu(to_nthwc(dy), dtype)
    DW = torch.empty(Cout, 5, 7, 8, 4, device=dev(), dtype=torch.float32)
    desc = hip.conv_desc(mode=hip.WGRAD, dtype=code, out_dtype=hip.F32, N=N, Tr=To, Hr=Ho, Wr=Wo,
                         Ts=T, Hs=H, Ws=WP, Cs=4, Cn=Cout, pack_w=8, **gk)
    ws = torch.empty(max(hip.conv_workspace_bytes(desc), 16) // 4, device=dev(), dtype=torch.float32)
    hip.conv_run(desc, X4, None, G, DW, workspace=ws)
    got = DW[:, :, :, :7, :3].permute(0, 4, 1, 2, 3)
    assert rel_err(got, gw) < (2e-5 if dtype == torch.float32 else 2e-3)
    assert float(DW[..., 3].abs().max()) == 0.0  # the zero-padded input channel


@pytest.mark.parametrize("dtype", DTYPES)
def test_batched_gemms_of_the_nonlocal_block(dtype):
    """theta.phi^T (fp32 out), P.g (K = L2 not a multiple of the 64-wide k tile), and the two
    contract-over-L1 products of the backward (nonlocal_helper.py:94,121 and their gradients)."""
    B, L1, L2, Ci = 3, 200, 72, 64
    gen = torch.Generator().manual_seed(5)
    code = hip.dtype_code(dtype)
    theta = q(torch.randn(B, L1, Ci, generator=gen), dtype)
    phi = q(torch.randn(B, L2, Ci, generator=gen), dtype)
    S = torch.empty(B, L1, L2, device=dev(), dtype=torch.float32)
    desc = hip.conv_desc(mode=hip.FPROP, dtype=code, out_dtype=hip.F32, N=1, Tr=1, Hr=1, Wr=L1,
                         Ts=1, Hs=1, Ws=L1, Cs=Ci, Cn=L2, batch=B, a_bstride=L1 * Ci,
                         b_bstride=L2 * Ci, o_bstride=L1 * L2)
    hip.conv_run(desc, gpu(theta, dtype), gpu(phi, dtype), None, S)
    S_ref = torch.einsum("blc,bmc->blm", theta.double(), phi.double())
    assert rel_err(S, S_ref) < 2e-5

    P = q(torch.softmax(S_ref.float() * Ci ** -0.5, dim=2), dtype)
    g = q(torch.randn(B, L2, Ci, generator=gen), dtype)
    gT = g.transpose(1, 2).contiguous()  # [B][Ci][L2]
    Y = torch.empty(B, L1, Ci, device=dev(), dtype=dtype)
    desc = hip.conv_desc(mode=hip.FPROP, dtype=code, out_dtype=code, N=1, Tr=1, Hr=1, Wr=L1,
                         Ts=1, Hs=1, Ws=L1, Cs=L2, Cn=Ci, batch=B, a_bstride=L1 * L2,
                         b_bstride=Ci * L2, o_bstride=L1 * Ci)
    hip.conv_run(desc, gpu(P, dtype), gpu(gT, dtype), None, Y)
    Y_ref = torch.einsum("blm,bmc->blc", P.double(), g.double())
    assert rel_err(Y.float(), Y_ref) < TOL[dtype]

    # library transpose agrees with torch
    gT_dev = torch.empty(B, Ci, L2, device=dev(), dtype=dtype)
    hip.call("vlfb_transpose2d", gp(g, dtype), hip.ptr(gT_dev), code, B, L2, Ci)
    assert torch.equal(gT_dev.float().cpu(), gT)

    # dphi[b][m][c] = sum_l dS[b][l][m] * theta[b][l][c]   (TN, batched, direct epilogue)
    dS = q(torch.randn(B, L1, L2, generator=gen), dtype)
    dphi = torch.empty(B, L2, Ci, device=dev(), dtype=dtype)
    desc = hip.conv_desc(mode=hip.WGRAD, dtype=code, out_dtype=code, N=1, Tr=1, Hr=1, Wr=L1,
                         Ts=1, Hs=1, Ws=L1, Cs=Ci, Cn=L2, batch=B, a_bstride=L1 * Ci,
                         p_bstride=L1 * L2, o_bstride=L2 * Ci)
    hip.conv_run(desc, gpu(theta, dtype), None, gpu(dS, dtype), dphi)
    dphi_ref = torch.einsum("blm,blc->bmc", dS.double(), theta.double())
    assert rel_err(dphi.float(), dphi_ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_row_bias_and_swapped_roles_gives_transposed_conv_output(dtype):
    """g^T = W_g . x^T with a per-row bias: the transposed 1x1x1 conv output the non-local block needs"""
    M, Cin, Cout = 150, 64, 128
    gen = torch.Generator().manual_seed(7)
    code = hip.dtype_code(dtype)
    x = q(torch.randn(M, Cin, generator=gen), dtype)
    w = q(torch.randn(Cout, Cin, generator=gen) * 0.1, dtype)
    b = torch.randn(Cout, generator=gen)
    O = torch.empty(Cout, M, device=dev(), dtype=dtype)
    desc = hip.conv_desc(mode=hip.FPROP, dtype=code, out_dtype=code, N=1, Tr=1, Hr=1, Wr=Cout,
                         Ts=1, Hs=1, Ws=Cout, Cs=Cin, Cn=M, bias_mode=hip.BIAS_ROW)
    # M = 150 is not a multiple of 4 -> scalar-store epilogue path
    hip.conv_run(desc, gpu(w, dtype), gpu(x, dtype), None, O, bias=gpu(b))
    ref = (x.double() @ w.double().t() + b.double()).t()
    assert rel_err(O.float(), ref) < TOL[dtype]


# ------------------------------------------------------------------------------------------------
POOLS = {
    "pool1": ((1, 3, 3), (1, 2, 2), (0, 1, 1), (2, 3, 11, 11)),   # resnet_video.py:190-196
    "pool2": ((2, 1, 1), (2, 1, 1), (0, 0, 0), (2, 4, 5, 5)),     # resnet_video.py:219-225
    "nlpool": ((1, 2, 2), (1, 2, 2), (0, 0, 0), (2, 2, 6, 6)),    # nonlocal_helper.py:48-54
    "other_window": ((1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 2, 7, 7)),   # not a compiled window shape: the generic backward kernel
    "pool1_ragged": ((1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 2, 14, 10)),  # even extents: the last window column / row is cut
    "pool1_bands": ((1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 3, 29, 23)),   # five bands of six input rows, the last one cut
    "pool1_nopad": ((1, 3, 3), (1, 2, 2), (0, 0, 0), (1, 2, 15, 13)),   # pads 0 as resnet_video.py:190-196 writes them
}


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", sorted(POOLS))
def test_maxpool_fwd_bwd(name, dtype):
    k, s, p, (N, T, H, W) = POOLS[name]
    Cc = 16
    gen = torch.Generator().manual_seed(11)
    code = hip.dtype_code(dtype)
    x = q(torch.randn(N, Cc, T, H, W, generator=gen), dtype)
    xd = x.double().requires_grad_(True)
    y_ref = F.max_pool3d(xd, k, s, p)
    To, Ho, Wo = y_ref.shape[2:]
    X = gpu(to_nthwc(x), dtype)
    Y = torch.empty(N, To, Ho, Wo, Cc, device=dev(), dtype=dtype)
    AM = torch.empty(N, To, Ho, Wo, Cc, device=dev(), dtype=torch.uint8)
    d = hip.pool_desc(code, N, T, H, W, Cc, To, Ho, Wo, k, s, p)
    import ctypes as C
    hip.call("vlfb_maxpool_fwd", C.byref(d), hip.ptr(X), hip.ptr(Y), hip.ptr(AM))
    assert torch.equal(to_ncthw(Y.float()).cpu(), y_ref.float())
    dy = q(torch.randn(N, Cc, To, Ho, Wo, generator=gen), dtype)
    (gx,) = torch.autograd.grad(y_ref, (xd,), dy.double())
    add = q(torch.randn(N, Cc, T, H, W, generator=gen), dtype)
    DX = torch.empty(N, T, H, W, Cc, device=dev(), dtype=dtype)
    # no residual operand, no mask (pool1 windows: the band kernel with the pooled rows staged in LDS)
    hip.call("vlfb_maxpool_bwd", C.byref(d), gp(to_nthwc(dy), dtype), hip.ptr(AM), hip.ptr(DX), None, None)
    assert rel_err(to_ncthw(DX.float()), gx) < TOL[dtype]
    hip.call("vlfb_maxpool_bwd", C.byref(d), gp(to_nthwc(dy), dtype), hip.ptr(AM), hip.ptr(DX),
             gp(to_nthwc(add), dtype), hip.ptr(X))
    ref = torch.where(x.double() > 0, gx + add.double(), torch.zeros_like(gx))
    assert rel_err(to_ncthw(DX.float()), ref) < TOL[dtype]
    # pool of a ReLU output with no other consumer: mask from the pooled tensor == mask from the input, bit for bit
    XR = torch.relu(X)
    hip.call("vlfb_maxpool_fwd", C.byref(d), hip.ptr(XR), hip.ptr(Y), hip.ptr(AM))
    D1, D2 = torch.empty_like(DX), torch.empty_like(DX)
    DY = gp(to_nthwc(dy), dtype)
    hip.call("vlfb_maxpool_bwd", C.byref(d), DY, hip.ptr(AM), hip.ptr(D1), None, hip.ptr(XR))
    hip.call("vlfb_maxpool_relu_bwd", C.byref(d), DY, hip.ptr(AM), hip.ptr(Y), hip.ptr(D2))
    assert torch.equal(D1, D2) and float(D1.float().abs().sum()) > 0


@pytest.mark.parametrize("dtype", DTYPES)
def test_avgpool_global_and_temporal(dtype):
    import ctypes as C
    gen = torch.Generator().manual_seed(13)
    code = hip.dtype_code(dtype)
    N, Cc, T, H, W = 2, 64, 4, 5, 5
    x = q(torch.randn(N, Cc, T, H, W, generator=gen), dtype)
    X = gpu(to_nthwc(x), dtype)
    for k in [(T, H, W), (T, 1, 1)]:  # head_helper.py:37-40 and :92-98
        To, Ho, Wo = T - k[0] + 1, H - k[1] + 1, W - k[2] + 1
        xd = x.double().requires_grad_(True)
        y_ref = F.avg_pool3d(xd, k, (1, 1, 1))
        Y = torch.empty(N, To, Ho, Wo, Cc, device=dev(), dtype=dtype)
        d = hip.pool_desc(code, N, T, H, W, Cc, To, Ho, Wo, k, (1, 1, 1), (0, 0, 0))
        hip.call("vlfb_avgpool_fwd", C.byref(d), hip.ptr(X), hip.ptr(Y))
        assert rel_err(to_ncthw(Y.float()), y_ref) < TOL[dtype]
        dy = q(torch.randn(N, Cc, To, Ho, Wo, generator=gen), dtype)
        (gx,) = torch.autograd.grad(y_ref, (xd,), dy.double())
        DX = torch.empty(N, T, H, W, Cc, device=dev(), dtype=dtype)
        hip.call("vlfb_avgpool_bwd", C.byref(d), gp(to_nthwc(dy), dtype), hip.ptr(DX), None, hip.ptr(X))
        ref = torch.where(x.double() > 0, gx, torch.zeros_like(gx))
        assert rel_err(to_ncthw(DX.float()), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_avgpool_bwd_from_an_fp32_gradient_into_two_terms(dtype):
    """vlfb_avgpool_bwd_two_term (where the "mix" path's fp32 head gradient re-enters the 16-bit backward): hi + lo reproduce the
    fp64 pool backward of the fp32 pooled gradient to ~2^-20 (the one-term kernel on the rounded gradient: ~2^-9 .. 2^-12, and
    the SAME error at every position of a channel); hi is the rounding of the exact value, masked by the forward values"""
    import ctypes as C
    gen = torch.Generator().manual_seed(14)
    code = hip.dtype_code(dtype)
    N, Cc, T, H, W = 2, 64, 4, 5, 5
    x = q(torch.randn(N, Cc, T, H, W, generator=gen), dtype)
    X = gpu(to_nthwc(x), dtype)
    for k in [(T, H, W), (T, 1, 1)]:  # head_helper.py:37-40 and :92-98
        To, Ho, Wo = T - k[0] + 1, H - k[1] + 1, W - k[2] + 1
        xd = x.double().requires_grad_(True)
        y_ref = F.avg_pool3d(xd, k, (1, 1, 1))
        dy = torch.randn(N, Cc, To, Ho, Wo, generator=gen) * 64.0       # (gradients carry the loss scale: fp16 normal range)
        (gx,) = torch.autograd.grad(y_ref, (xd,), dy.double())
        ref = torch.where(x.double() > 0, gx, torch.zeros_like(gx))
        d = hip.pool_desc(code, N, T, H, W, Cc, To, Ho, Wo, k, (1, 1, 1), (0, 0, 0))
        HI, LO = (torch.empty(N, T, H, W, Cc, device=dev(), dtype=dtype) for _ in range(2))
        DY = gpu(to_nthwc(dy))
        hip.call("vlfb_avgpool_bwd_two_term", C.byref(d), hip.ptr(DY), hip.ptr(HI), hip.ptr(LO), hip.ptr(X))
        two = to_ncthw(HI.double() + LO.double())
        assert rel_err(two, ref) < (2e-6 if dtype == torch.float16 else 4e-5), rel_err(two, ref)
        assert rel_err(to_ncthw(HI.float()), ref) < TOL[dtype]
        ONE = torch.empty_like(HI)                      # the one-term path: rounded gradient in, one rounding out
        hip.call("vlfb_avgpool_bwd", C.byref(d), gp(to_nthwc(q(dy, dtype)), dtype), hip.ptr(ONE), None, hip.ptr(X))
        assert rel_err(two, ref) < 0.05 * rel_err(to_ncthw(ONE.float()), ref)


@pytest.mark.parametrize("dtype", DTYPES)
def test_softmax_fwd_bwd(dtype):
    gen = torch.Generator().manual_seed(17)
    code = hip.dtype_code(dtype)
    rows, cols, scale = 37, 784, 512 ** -0.5
    s = torch.randn(rows, cols, generator=gen) * 8
    S = gpu(s)
    P = torch.empty(rows, cols, device=dev(), dtype=dtype)
    hip.call("vlfb_softmax_fwd", hip.ptr(S), hip.ptr(P), code, rows, cols, scale)
    sd = s.double().requires_grad_(True)
    p_ref = torch.softmax(sd * scale, dim=1)
    assert rel_err(P.float(), p_ref) < (1e-5 if dtype == torch.float32 else 4e-3)
    dp = torch.randn(rows, cols, generator=gen)
    # backward uses the stored (rounded) probabilities, as the engine does
    pq = P.float().cpu().double()
    ds_ref = scale * pq * (dp.double() - (dp.double() * pq).sum(1, keepdim=True))
    DS = torch.empty(rows, cols, device=dev(), dtype=dtype)
    hip.call("vlfb_softmax_bwd", gp(dp), hip.ptr(P), hip.ptr(DS), code, rows, cols, scale)
    assert rel_err(DS.float(), ds_ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_add_relu_colsum(dtype):
    gen = torch.Generator().manual_seed(19)
    code = hip.dtype_code(dtype)
    n = 8 * 1000 + 3
    a = q(torch.randn(n, generator=gen), dtype)
    b = q(torch.randn(n, generator=gen), dtype)
    m = q(torch.randn(n, generator=gen), dtype)
    Y = torch.empty(n, device=dev(), dtype=dtype)
    hip.call("vlfb_add", gp(a, dtype), gp(b, dtype), hip.ptr(Y), gp(m, dtype), code, n, 1)
    ref = torch.where(m > 0, torch.relu(a + b), torch.zeros_like(a))
    assert rel_err(Y.float(), q(ref, dtype)) < 1e-6
    rows, cols = 1000, 96
    g = q(torch.randn(rows, cols, generator=gen), dtype)
    out = torch.empty(cols, device=dev(), dtype=torch.float32)
    hip.call("vlfb_colsum", gp(g, dtype), code, rows, cols, cols, hip.ptr(out), 0)
    assert rel_err(out, g.double().sum(0)) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
def test_layernorm_dropout(dtype):
    from oracle import rng as orng
    gen = torch.Generator().manual_seed(23)
    code = hip.dtype_code(dtype)
    rows, cols = 9, 512
    x = q(torch.randn(rows, cols, generator=gen) * 3 + 1, dtype)
    xd = x.double().requires_grad_(True)
    y_ref = F.layer_norm(xd, (cols,), eps=1e-5)
    Y = torch.empty(rows, cols, device=dev(), dtype=dtype)
    rstd = torch.empty(rows, device=dev(), dtype=torch.float32)
    hip.call("vlfb_layernorm_fwd", gp(x, dtype), hip.ptr(Y), hip.ptr(rstd), code, rows, cols, 1e-5)
    assert rel_err(Y.float(), y_ref) < TOL[dtype]
    dy = q(torch.randn(rows, cols, generator=gen), dtype)
    (gx,) = torch.autograd.grad(y_ref, (xd,), dy.double())
    DX = torch.empty(rows, cols, device=dev(), dtype=dtype)
    hip.call("vlfb_layernorm_bwd", gp(dy, dtype), hip.ptr(Y), hip.ptr(rstd), hip.ptr(DX), code, rows, cols)
    assert rel_err(DX.float(), gx) < (1e-4 if dtype == torch.float32 else 2e-2)

    # dropout: mask must equal the oracle's generator, defined over the REFERENCE (r, c, k) index
    R, K, Cc, ratio, seed = 3, 5, 16, 0.2, 0x1234567890ABCDEF
    xs = q(torch.randn(R, K, Cc, generator=gen), dtype)  # stored [(r*K + k)*C + c]
    Yd = torch.empty(R, K, Cc, device=dev(), dtype=dtype)
    Md = torch.empty(R, K, Cc, device=dev(), dtype=torch.uint8)
    hip.call("vlfb_dropout_fwd", gp(xs, dtype), hip.ptr(Yd), hip.ptr(Md), code, R, K, Cc, ratio, seed)
    keep_ref = orng.dropout_keep_mask(seed, (R, Cc, K), ratio)  # reference layout (R, C, K)
    keep_ref = torch.from_numpy(keep_ref).permute(0, 2, 1)
    assert torch.equal(Md.cpu().bool(), keep_ref)
    ref = torch.where(keep_ref, xs / (1 - ratio), torch.zeros_like(xs))
    assert rel_err(Yd.float(), q(ref, dtype)) < 1e-6 if dtype == torch.float32 else rel_err(Yd.float(), ref) < 4e-3
    DXd = torch.empty(R, K, Cc, device=dev(), dtype=dtype)
    hip.call("vlfb_dropout_bwd", gp(xs, dtype), hip.ptr(Md), hip.ptr(DXd), code, R * K * Cc, ratio)
    assert rel_err(DXd.float(), ref) < 4e-3


@pytest.mark.parametrize("dtype", DTYPES)
def test_fc_and_sigmoid_ce(dtype):
    gen = torch.Generator().manual_seed(29)
    code = hip.dtype_code(dtype)
    rows, cin, cout = 6, 2560, 157
    x = q(torch.randn(rows, cin, generator=gen), dtype)
    w = torch.randn(cout, cin, generator=gen) * 0.01
    b = torch.randn(cout, generator=gen) * 0.1
    labels = (torch.rand(rows, cout, generator=gen) < 0.05).to(torch.int32)
    labels[0, 3] = -1  # ignored entry
    scale = 1.0 / 8
    xd, wd, bd = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    logits_ref = xd @ wd.t() + bd
    t = labels.double()
    valid = t >= 0
    pos = (logits_ref >= 0).double()
    l = -logits_ref * (t - pos) + torch.log(1 + torch.exp(logits_ref - 2 * logits_ref * pos))
    loss_ref = scale * (l * valid).sum() / valid.sum()
    gx, gw, gb = torch.autograd.grad(loss_ref, (xd, wd, bd))
    X = gpu(x, dtype)
    L = torch.empty(rows, cout, device=dev())
    hip.call("vlfb_fc_fwd", hip.ptr(X), code, gp(w), gp(b), hip.ptr(L), rows, cin, cout)
    assert rel_err(L, logits_ref) < 1e-5
    prob = torch.empty_like(L)
    loss = torch.empty(1, device=dev())
    dl = torch.empty_like(L)
    hip.call("vlfb_sigmoid_ce", hip.ptr(L), gp(labels), hip.ptr(prob), hip.ptr(loss), hip.ptr(dl), rows, cout, scale)
    assert rel_err(prob, torch.sigmoid(logits_ref)) < 1e-5
    assert abs(loss.item() - loss_ref.item()) < 1e-5 * abs(loss_ref.item())
    DX = torch.empty(rows, cin, device=dev(), dtype=dtype)
    DW = torch.empty(cout, cin, device=dev())
    DB = torch.empty(cout, device=dev())
    hip.call("vlfb_fc_bwd", hip.ptr(X), code, gp(w), hip.ptr(dl), hip.ptr(DX), hip.ptr(DW), hip.ptr(DB), rows, cin, cout, 0)
    # dx ~ 1e-5 lies in the fp16 subnormal range here (the engine scales the loss gradient by 1024 for that reason)
    assert rel_err(DX.float(), gx) < (5e-3 if dtype == torch.float16 else TOL[dtype])
    assert rel_err(DW, gw) < 1e-4 and rel_err(DB, gb) < 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("D", [512, 40])
def test_fbo_attention_core(dtype, D):
    """D = 512: chip-wide kernels (dot / mix / row fix-up); D = 40: per-row kernels"""
    gen = torch.Generator().manual_seed(31)
    code = hip.dtype_code(dtype)
    R, K = 5, 300
    theta = q(torch.randn(R, D, generator=gen), dtype)
    phi = q(torch.randn(R, K, D, generator=gen), dtype)
    g = q(torch.randn(R, K, D, generator=gen), dtype)
    scale = D ** -0.5
    td, pd, gd = (t.double().requires_grad_(True) for t in (theta, phi, g))
    aff = torch.einsum("rd,rkd->rk", td, pd) * scale
    p_ref = torch.softmax(aff, dim=1)
    t_ref = torch.einsum("rk,rkd->rd", p_ref, gd)
    P = torch.empty(R, K, device=dev())
    T = torch.empty(R, D, device=dev(), dtype=dtype)
    TH, PH, G = gpu(theta, dtype), gpu(phi, dtype), gpu(g, dtype)
    hip.call("vlfb_fbo_attn_fwd", hip.ptr(TH), hip.ptr(PH), hip.ptr(G), hip.ptr(P), hip.ptr(T), code, R, K, D, D, scale)
    assert rel_err(P, p_ref) < 1e-5
    assert rel_err(T.float(), t_ref) < TOL[dtype]
    dt = q(torch.randn(R, D, generator=gen), dtype)
    gth, gph, gg = torch.autograd.grad(t_ref, (td, pd, gd), dt.double())
    DTH = torch.empty(R, D, device=dev(), dtype=dtype)
    DPH = torch.empty(R, K, D, device=dev(), dtype=dtype)
    DG = torch.empty(R, K, D, device=dev(), dtype=dtype)
    DSW = torch.empty(R, K, device=dev())
    hip.call("vlfb_fbo_attn_bwd", gp(dt, dtype), hip.ptr(TH), hip.ptr(PH), hip.ptr(G), hip.ptr(P),
             hip.ptr(DTH), hip.ptr(DPH), hip.ptr(DG), hip.ptr(DSW), code, R, K, D, D, scale)
    assert rel_err(DTH.float(), gth) < TOL[dtype]
    assert rel_err(DPH.float(), gph) < TOL[dtype]
    assert rel_err(DG.float(), gg) < TOL[dtype]


def test_sgd_weight_prep_cast():
    gen = torch.Generator().manual_seed(37)
    n = 10007
    p, g, m = (torch.randn(n, generator=gen) for _ in range(3))
    lr, wd, mu = 0.02, 1e-4, 0.9
    P, G, M = gpu(p.clone()), gpu(g.clone()), gpu(m.clone())
    hip.call("vlfb_sgd_update", hip.ptr(P), hip.ptr(G), hip.ptr(M), n, lr, wd, mu, 1)
    g2 = g.double() + wd * p.double()
    m2 = mu * m.double() + lr * g2
    step = (1 + mu) * m2 - mu * m.double()
    assert rel_err(P, p.double() - step) < 1e-6 and rel_err(M, m2) < 1e-6 and rel_err(G, step) < 1e-6
    cout, taps, cin = 40, 3, 72
    w = torch.randn(cout, taps, cin, generator=gen)
    s = torch.rand(cout, generator=gen) + 0.5
    for dtype in DTYPES:
        code = hip.dtype_code(dtype)
        wf = torch.empty(cout, taps, cin, device=dev(), dtype=dtype)
        wg = torch.empty(cin, taps, cout, device=dev(), dtype=dtype)
        hip.call("vlfb_weight_prep", gp(w), gp(s), hip.ptr(wf), hip.ptr(wg), code, cout, taps, cin)
        ref = (w * s.view(-1, 1, 1)).to(dtype)
        assert torch.equal(wf.cpu(), ref)
        assert torch.equal(wg.cpu(), ref.permute(2, 1, 0).contiguous())
    x = torch.randn(1000, generator=gen)
    xb = torch.empty(1000, device=dev(), dtype=torch.bfloat16)
    hip.call("vlfb_cast", gp(x), hip.F32, hip.ptr(xb), hip.BF16, 1000)
    assert torch.equal(xb.cpu(), x.to(torch.bfloat16))  # round-to-nearest-even, same as torch


@pytest.mark.parametrize("dtype", DTYPES)
def test_roi_align_max_head(dtype):
    """RoIAlign(7x7, 1/16, sampling_ratio 0) + 7x7 max (head_helper.py:88-123): integer decisions
    bit-exact against the oracle, values within tolerance, backward against autograd."""
    from oracle.roi_align import roi_align_loop, roi_align_torch
    gen = torch.Generator().manual_seed(41)
    code = hip.dtype_code(dtype)
    N, Cc, H, W = 2, 32, 14, 14
    feat = q(torch.randn(N, Cc, H, W, generator=gen), dtype)
    rois = torch.tensor([[0, 10, 20, 200, 220], [1, 0, 0, 223, 223], [1, 100, 50, 108, 58],
                         [0, 3.3, 7.7, 15.2, 223], [1, 64, 64, 223, 100], [0, 0, 190, 223, 223]],
                        dtype=torch.float32)
    R = rois.shape[0]
    out_np, dbg_np = roi_align_loop(feat.numpy(), rois.numpy())
    Fg = gpu(feat.permute(0, 2, 3, 1).contiguous(), dtype)  # [N,H,W,C]
    O = torch.empty(R, Cc, device=dev(), dtype=dtype)
    AB = torch.empty(R, Cc, device=dev(), dtype=torch.uint8)
    DBG = torch.empty(R, 7, 7, 8, device=dev(), dtype=torch.int32)
    hip.call("vlfb_roi_align_max_fwd", hip.ptr(Fg), code, gp(rois), hip.ptr(O), hip.ptr(AB), hip.ptr(DBG),
             N, H, W, Cc, R, 7, 1.0 / 16)
    assert np.array_equal(DBG.cpu().numpy(), dbg_np), "RoIAlign integer decisions must be bit-exact"
    if dtype == torch.float32:
        # ... and those of EVERY grid sample (2 x 2 per bin for the largest box here), not only the first of each bin
        from oracle.roi_align import roi_decisions
        ALL = torch.empty(R, 7, 7, 3, 3, 8, device=dev(), dtype=torch.int32)
        hip.call("vlfb_roi_align_decisions", gp(rois), hip.ptr(ALL), H, W, R, 7, 1.0 / 16, 3)
        want = roi_decisions(rois.numpy(), H, W, 7, 1.0 / 16, 3)
        assert want[..., 1].max() >= 2 and np.array_equal(ALL.cpu().numpy(), want)
    flat = torch.from_numpy(out_np).reshape(R, Cc, 49)
    ref_max, ref_arg = flat.max(dim=2)
    assert rel_err(O.float(), q(ref_max, dtype)) < (1e-6 if dtype == torch.float32 else 4e-3)
    if dtype == torch.float32:
        assert torch.equal(AB.cpu().long(), ref_arg)
    # backward
    fd = feat.double().requires_grad_(True)
    pooled = roi_align_torch(fd, rois.numpy()).reshape(R, Cc, 49)
    sel = pooled.gather(2, AB.cpu().long().unsqueeze(2)).squeeze(2)
    dout = q(torch.randn(R, Cc, generator=gen), dtype)
    (gf,) = torch.autograd.grad(sel, (fd,), dout.double())
    DF = torch.zeros(N, H, W, Cc, device=dev(), dtype=torch.float32)
    hip.call("vlfb_roi_align_max_bwd", gp(dout, dtype), code, gp(rois), hip.ptr(AB), hip.ptr(DF),
             N, H, W, Cc, R, 7, 1.0 / 16)
    assert rel_err(DF.permute(0, 3, 1, 2), gf) < 1e-5


@pytest.mark.parametrize("tdt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("M,K,Nc", [(33, 2048, 512), (33, 512, 512), (1, 512, 2048), (64, 1024, 520), (17, 128, 16)])
def test_skinny_row_products_of_the_fbo_head(M, K, Nc, tdt):
    """1x1x1 convs on a handful of rows (lfb_helper.py:170-263 on one row per RoI): the 16-column kernel whose waves split
    K (csrc/vlfb_gemm_skinny.hip) -- FPROP with bias / residual / ReLU, DGRAD-shaped launches with a residual and a mask,
    fp32 output -- against fp64 and against the 128 x 128 tile family (algo = TILE128) on the same operands"""
    code = hip.dtype_code(tdt)
    tol = {torch.bfloat16: 6e-3, torch.float16: 8e-4}[tdt]
    gen = torch.Generator().manual_seed(M * 7 + K + Nc)
    a = q(torch.randn(M, K, generator=gen), tdt)
    w = q(torch.randn(Nc, K, generator=gen) / math.sqrt(K), tdt)
    bias = torch.randn(Nc, generator=gen)
    res = q(torch.randn(M, Nc, generator=gen), tdt)
    msk = q(torch.randn(M, Nc, generator=gen), tdt)
    A, Wd, R, Mk, Bs = gpu(a, tdt), gpu(w, tdt), gpu(res, tdt), gpu(msk, tdt), gpu(bias)
    lin = a.double() @ w.double().t()
    cases = [
        (dict(relu=1, bias_mode=hip.BIAS_COL), dict(bias=Bs, R=R), tdt, torch.relu(lin + bias.double() + res.double())),
        (dict(alpha=0.5), dict(R=R, mask=Mk), tdt, torch.where(msk.double() > 0, 0.5 * lin + res.double(), torch.zeros_like(lin))),
        (dict(), dict(), torch.float32, lin),
    ]
    for dk, rk, odt, want in cases:
        outs = []
        for algo in (hip.ALGO_AUTO, hip.ALGO_TILE128):
            O = torch.full((M, Nc), float("nan"), device=dev(), dtype=odt)
            desc = hip.conv_desc(mode=hip.FPROP, dtype=code, out_dtype=hip.dtype_code(odt), N=1, Tr=1, Hr=1, Wr=M, Ts=1, Hs=1, Ws=M,
                                 Cs=K, Cn=Nc, algo=algo, **dk)
            hip.conv_run(desc, A, Wd, None, O, **rk)
            torch.cuda.synchronize()
            outs.append(O)
        assert not torch.isnan(outs[0].float()).any()
        assert rel_err(outs[0].float(), want) < (tol if odt != torch.float32 else 2e-6 * math.sqrt(K) + 1e-6)
        assert rel_err(outs[0].float(), outs[1].float()) < (tol if odt != torch.float32 else 1e-5)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,Cs,Cn,taps", [(25088, 512, 256, 1), (6272, 1024, 512, 1), (33, 512, 512, 1), (9900, 2048, 512, 1),
                                         (3000, 64, 136, 1), (1568, 128, 128, 3)])
def test_wgrad_with_bias_gradient(M, Cs, Cn, taps, dtype):
    """vlfb_conv_run_wgrad_bias: weight gradient AND bias gradient db = alpha * s * colsum(dY) of a conv that carries a bias
    (nonlocal_helper.py:36-77, lfb_helper.py:175-200) from one pass over dY -- inside the transposed-read kernel on the 16-bit
    paths (single launch and split-K slabs), a column-sum pass behind the launch for the other families (fp32, 256 x 256
    pipelined).  Against fp64; the weight gradient must be bit-identical to the plain WGRAD launch; deterministic."""
    g = torch.Generator().manual_seed(5)
    T = M // 14 // 14 if taps == 3 else 1
    if taps == 3:
        geom = dict(N=1, Tr=T, Hr=14, Wr=14, Ts=T, Hs=14, Ws=14, kt=3, pt=1)
        M = T * 196
    else:
        geom = dict(N=1, Tr=1, Hr=1, Wr=M, Ts=1, Hs=1, Ws=M)
    x = q(torch.randn(M, Cs, generator=g), dtype)
    dy = q(torch.randn(M, Cn, generator=g) * 0.1 + 0.02, dtype)
    s = torch.rand(Cn, generator=g) + 0.5
    alpha = 0.25
    code = hip.dtype_code(dtype)
    kw = dict(mode=hip.WGRAD, dtype=code, out_dtype=hip.F32, Cs=Cs, Cn=Cn, alpha=alpha, **geom)
    d_plain = hip.conv_desc(**kw)
    d_bias = hip.conv_desc(wgrad_bias=1, **kw)
    ws = torch.empty(max(hip.conv_workspace_bytes(d_bias), hip.conv_workspace_bytes(d_plain), 16) // 4, device=dev())
    X, DY, S = gpu(x, dtype), gpu(dy, dtype), gpu(s)
    dw0 = torch.empty(Cn * taps * Cs, device=dev())
    dw1 = torch.full((Cn * taps * Cs,), 7.0, device=dev())
    db = torch.full((Cn,), 7.0, device=dev())
    hip.conv_run(d_plain, X, None, DY, dw0, rowscale=S, workspace=ws)
    hip.conv_run(d_bias, X, None, DY, dw1, rowscale=S, workspace=ws, dbias=db)
    torch.cuda.synchronize()
    assert torch.equal(dw0, dw1), "the weight gradient changed with the fused bias gradient"
    ref = dy.double().sum(0) * s.double() * alpha
    err = rel_err(db, ref)
    print("\n[%s M=%d Cs=%d Cn=%d taps=%d] plan %s: bias gradient rel err %.2e" % (dtype, M, Cs, Cn, taps, hip.conv_plan(d_bias), err))
    assert err < 2e-5, err
    db2 = torch.empty(Cn, device=dev())
    hip.conv_run(d_bias, X, None, DY, dw1, rowscale=S, workspace=ws, dbias=db2)
    torch.cuda.synchronize()
    assert torch.equal(db, db2), "the bias gradient is not deterministic (plan %s)" % hip.conv_plan(d_bias)
