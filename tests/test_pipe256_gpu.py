"""The 256-row phase-pipelined kernels (csrc/vlfb_gemm8.hip, conv desc `algo` = VLFB_ALGO_PIPE256) against
the 128x128 kernels (algo = VLFB_ALGO_TILE128) and against fp64 torch.

Both NT families accumulate k in the same order with the same MFMA, so their outputs must be BIT-IDENTICAL
for every epilogue; that is a far sharper check of the counted-vmcnt / slot-ring pipeline than a tolerance
(one stale LDS row anywhere changes bits).  Shapes are chosen to reach every instance: tile widths 256 / 128,
tile heights 256 / 196, plain rows / gathered FPROP (strided, dilated) / gathered DGRAD, K tails (K % 64 != 0),
ragged last row tile, column counts that do not fill a tile, batched launches, residual + ReLU + bias + mask
epilogues, fp32 outputs.  Each case runs several times: a race would show up as run-to-run differences.
The split TN kernel sums its slabs in another order than the 128x128 one, so it is checked against fp64.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import dev, q, rel_err, to_ncthw, to_nthwc, w_to_kernel

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _hip():
    from vlfb import hip
    hip.lib()
    return hip


def _geom(k, s, p, d):
    return dict(kt=k[0], kh=k[1], kw=k[2], st=s[0], sh=s[1], sw=s[2], pt=p[0], ph=p[1], pw=p[2], dt=d[0], dh=d[1], dw=d[2])


# name: (N, Cin, Cout, T, H, W, k, stride, pad, dil)   -- output rows >= 1024, Cin * 2 bytes % 128 == 0
NT_CASES = {
    "ident_w256_h196": (2, 128, 512, 4, 14, 14, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),    # 1568 rows = 8 x 196 -> 196-row tiles
    "ident_w128": (1, 192, 128, 4, 16, 17, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),         # 1088 rows: ragged, 256-row tiles
    "ident_cols_ragged": (1, 128, 328, 3, 19, 19, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),  # 328 = 256 + 72 columns
    "temporal3": (1, 128, 256, 5, 15, 15, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),
    "spatial3": (2, 64, 256, 2, 23, 23, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),
    "spatial3_dil2": (1, 128, 128, 3, 20, 20, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2)),
    "spatial3_s2_fprop_only": (1, 64, 256, 2, 47, 47, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 1, 1)),
}


@pytest.mark.parametrize("case", sorted(NT_CASES))
def test_nt_pipe256_is_bit_identical_to_tile128_and_matches_fp64(case):
    hip = _hip()
    N, Cin, Cout, T, H, W, k, s, p, d = NT_CASES[case]
    gen = torch.Generator().manual_seed(sum(map(ord, case)))
    x = q(torch.randn(N, Cin, T, H, W, generator=gen), BF)
    w = q(torch.randn(Cout, Cin, *k, generator=gen) / math.sqrt(Cin * k[0] * k[1] * k[2]), BF)
    To, Ho, Wo = [(a + 2 * pp - dd * (kk - 1) - 1) // ss + 1 for a, kk, ss, pp, dd in zip((T, H, W), k, s, p, d)]
    bias = torch.randn(Cout, generator=gen)
    res = q(torch.randn(N, Cout, To, Ho, Wo, generator=gen), BF)
    A = to_nthwc(x).to(dev(), BF)
    Bw = w_to_kernel(w).to(dev(), BF)
    R = to_nthwc(res).to(dev(), BF)
    bg = bias.to(dev())
    y_lin = F.conv3d(x.double(), w.double(), None, s, p, d)
    outs = {}
    for algo in (hip.ALGO_TILE128, hip.ALGO_PIPE256):
        for rep in range(3 if algo == hip.ALGO_PIPE256 else 1):
            O = torch.full((N, To, Ho, Wo, Cout), float("nan"), device=dev(), dtype=BF)
            desc = hip.conv_desc(mode=hip.FPROP, dtype=hip.BF16, out_dtype=hip.BF16, N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T, Hs=H,
                                 Ws=W, Cs=Cin, Cn=Cout, relu=1, bias_mode=hip.BIAS_COL, algo=algo, **_geom(k, s, p, d))
            hip.conv_run(desc, A, Bw, None, O, bias=bg, R=R)
            O32 = torch.full((N, To, Ho, Wo, Cout), float("nan"), device=dev(), dtype=torch.float32)
            desc = hip.conv_desc(mode=hip.FPROP, dtype=hip.BF16, out_dtype=hip.F32, N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T, Hs=H,
                                 Ws=W, Cs=Cin, Cn=Cout, alpha=0.5, algo=algo, **_geom(k, s, p, d))
            hip.conv_run(desc, A, Bw, None, O32)
            torch.cuda.synchronize()
            outs.setdefault(algo, []).append((O.clone(), O32.clone()))
    ref, ref32 = outs[hip.ALGO_TILE128][0]
    for O, O32 in outs[hip.ALGO_PIPE256]:
        assert torch.equal(O.view(torch.int16), ref.view(torch.int16)), "fprop: pipelined kernel differs from the 128x128 kernel"
        assert torch.equal(O32.view(torch.int32), ref32.view(torch.int32)), "fprop fp32-out differs"
    assert rel_err(to_ncthw(ref32), 0.5 * y_lin) < 2e-5
    if case.endswith("fprop_only"):
        return
    # ---- dgrad (unit stride: gathered DGRAD instance; 1x1x1: plain rows) with add + mask epilogue --------
    dy = q(torch.randn(N, Cout, To, Ho, Wo, generator=gen), BF)
    mask_src = q(torch.randn(N, Cin, T, H, W, generator=gen), BF)
    add_src = q(torch.randn(N, Cin, T, H, W, generator=gen), BF)
    G = to_nthwc(dy).to(dev(), BF)
    Wd = w.permute(1, 2, 3, 4, 0).contiguous().to(dev(), BF)
    Rm, Mm = to_nthwc(add_src).to(dev(), BF), to_nthwc(mask_src).to(dev(), BF)
    if Cin < 128:
        return                                    # the pipelined kernel needs >= 128 output columns
    got = {}
    for algo in (hip.ALGO_TILE128, hip.ALGO_PIPE256):
        DX = torch.full((N, T, H, W, Cin), float("nan"), device=dev(), dtype=BF)
        desc = hip.conv_desc(mode=hip.DGRAD, dtype=hip.BF16, out_dtype=hip.BF16, N=N, Tr=T, Hr=H, Wr=W, Ts=To, Hs=Ho, Ws=Wo,
                             Cs=Cout, Cn=Cin, algo=algo, **_geom(k, s, p, d))
        hip.conv_run(desc, G, Wd, None, DX, R=Rm, mask=Mm)
        torch.cuda.synchronize()
        got[algo] = DX
    assert torch.equal(got[hip.ALGO_PIPE256].view(torch.int16), got[hip.ALGO_TILE128].view(torch.int16)), "dgrad differs"
    xd = x.double().requires_grad_(True)
    gx, = torch.autograd.grad(F.conv3d(xd, w.double(), None, s, p, d), xd, dy.double())
    dx_ref = torch.where(mask_src.double() > 0, gx + add_src.double(), torch.zeros_like(gx))
    assert rel_err(to_ncthw(got[hip.ALGO_PIPE256].float()), dx_ref) < 1e-2


@pytest.mark.parametrize("L1,L2,Ci,B", [(1200, 328, 128, 3), (1030, 784, 256, 2)])
def test_batched_attention_products_pipe256(L1, L2, Ci, B):
    """S = theta.phi^T (fp32 out), Y = P.g (K = L2 is not a multiple of 64: zero-filled k tail) and the TN
    product dPhi = dS^T.theta, batched, bit-identical between the kernel families"""
    hip = _hip()
    gen = torch.Generator().manual_seed(L1 + L2)
    th = q(torch.randn(B, L1, Ci, generator=gen), BF).to(dev(), BF)
    ph = q(torch.randn(B, L2, Ci, generator=gen), BF).to(dev(), BF)
    P = q(torch.rand(B, L1, L2, generator=gen), BF).to(dev(), BF)
    gT = q(torch.randn(B, Ci, L2, generator=gen), BF).to(dev(), BF)
    res = {}
    for algo in (hip.ALGO_TILE128, hip.ALGO_PIPE256):
        gemm = lambda **kw: hip.conv_desc(mode=hip.FPROP, dtype=hip.BF16, N=1, Tr=1, Hr=1, Wr=L1, Ts=1, Hs=1, Ws=L1, batch=B,
                                          algo=algo, **kw)
        S = torch.full((B, L1, L2), float("nan"), device=dev(), dtype=torch.float32)
        hip.conv_run(gemm(out_dtype=hip.F32, Cs=Ci, Cn=L2, a_bstride=L1 * Ci, b_bstride=L2 * Ci, o_bstride=L1 * L2), th, ph, None, S)
        Y = torch.full((B, L1, Ci), float("nan"), device=dev(), dtype=BF)
        hip.conv_run(gemm(out_dtype=hip.BF16, Cs=L2, Cn=Ci, a_bstride=L1 * L2, b_bstride=Ci * L2, o_bstride=L1 * Ci), P, gT, None, Y)
        dphi = torch.full((B, L2, Ci), float("nan"), device=dev(), dtype=BF)
        d_tn = hip.conv_desc(mode=hip.WGRAD, dtype=hip.BF16, out_dtype=hip.BF16, N=1, Tr=1, Hr=1, Wr=L1, Ts=1, Hs=1, Ws=L1,
                             Cs=Ci, Cn=L2, batch=B, a_bstride=L1 * Ci, p_bstride=L1 * L2, o_bstride=L2 * Ci, splits=1, algo=algo)
        hip.conv_run(d_tn, th, None, P, dphi)
        torch.cuda.synchronize()
        res[algo] = (S, Y, dphi)
    for a, b, what in zip(res[hip.ALGO_TILE128], res[hip.ALGO_PIPE256], ("S", "Y", "dPhi")):
        assert torch.equal(a.view(torch.int32 if a.dtype == torch.float32 else torch.int16),
                           b.view(torch.int32 if b.dtype == torch.float32 else torch.int16)), what
    S, Y, dphi = res[hip.ALGO_PIPE256]
    assert rel_err(S, torch.bmm(th.double().cpu(), ph.double().cpu().transpose(1, 2))) < 2e-5
    assert rel_err(Y.float(), torch.bmm(P.double().cpu(), gT.double().cpu().transpose(1, 2))) < 1e-2
    assert rel_err(dphi.float(), torch.bmm(P.double().cpu().transpose(1, 2), th.double().cpu())) < 1e-2


@pytest.mark.parametrize("M,Cin,Cout", [(5000, 256, 640), (12544, 512, 2048)])
def test_split_wgrad_pipe256_matches_fp64(M, Cin, Cout):
    """wgrad of a 1x1x1 conv through the 256x256 TN kernel (fp32 slabs + reduce) against fp64, with the
    frozen-affine row scale, and against the 128x128 kernel to fp32 summation-order noise"""
    hip = _hip()
    gen = torch.Generator().manual_seed(M)
    x = q(torch.randn(M, Cin, generator=gen), BF)
    g = q(torch.randn(M, Cout, generator=gen), BF)
    scale = torch.rand(Cout, generator=gen) + 0.5
    X, G, Sc = x.to(dev(), BF), g.to(dev(), BF), scale.to(dev())
    ref = (g.double().t() @ x.double()) * scale.double()[:, None]
    out = {}
    for algo in (hip.ALGO_TILE128, hip.ALGO_PIPE256):
        d = hip.conv_desc(mode=hip.WGRAD, dtype=hip.BF16, out_dtype=hip.F32, N=1, Tr=1, Hr=1, Wr=M, Ts=1, Hs=1, Ws=M,
                          Cs=Cin, Cn=Cout, algo=algo)
        ws = torch.empty(max(hip.conv_workspace_bytes(d) // 4, 4), device=dev(), dtype=torch.float32)
        O = torch.full((Cout, Cin), float("nan"), device=dev(), dtype=torch.float32)
        hip.conv_run(d, X, None, G, O, rowscale=Sc, workspace=ws)
        torch.cuda.synchronize()
        out[algo] = O
        assert rel_err(O, ref) < 2e-5, algo
    assert rel_err(out[hip.ALGO_PIPE256], out[hip.ALGO_TILE128]) < 1e-5


def test_pipe256_rejects_problems_it_cannot_run():
    hip = _hip()
    d = hip.conv_desc(mode=hip.FPROP, dtype=hip.F32, out_dtype=hip.F32, N=1, Tr=1, Hr=1, Wr=2048, Ts=1, Hs=1, Ws=2048,
                      Cs=128, Cn=256, algo=hip.ALGO_PIPE256)
    t = torch.zeros(2048 * 256, device=dev())
    with pytest.raises(hip.VlfbError):
        hip.conv_run(d, t, t, None, t)
