"""The whole-row WGRAD kernel (csrc/vlfb_wgrad_rows.hip: 64 -> 64 channels, unit stride, rows of <= 64 positions --
what the library picks for the res2 3x3 / 3x1x1 weight gradients) against fp64 torch and against the generic
128-column TN kernel (algo = VLFB_ALGO_TILE128).  Rows of 8 ... 56 positions (partial last k-step), paddings on every
side, both tap families, bf16 and fp16, with the frozen-affine row scale and accumulation."""
import math

import pytest
import torch
import torch.nn.functional as F

from gpu_util import dev, q, rel_err, to_nthwc, w_to_kernel

pytestmark = pytest.mark.gpu

# name: (N, T, H, W, k, pad)
CASES = {
    "spatial3_w56": (1, 2, 9, 56, (1, 3, 3), (0, 1, 1)),
    "spatial3_w24": (2, 3, 7, 24, (1, 3, 3), (0, 1, 1)),
    "spatial3_w8": (1, 2, 5, 8, (1, 3, 3), (0, 1, 1)),
    "temporal3_w40": (2, 5, 6, 40, (3, 1, 1), (1, 0, 0)),
    "temporal3_w16": (1, 4, 3, 16, (3, 1, 1), (1, 0, 0)),
    "fat_temporal3_w56": (1, 4, 3, 56, (3, 1, 1), (1, 0, 0)),      # 256 -> 64 channels: the fat-input kernel
    "fat_temporal3_w24": (2, 3, 5, 24, (3, 1, 1), (1, 0, 0)),
}


@pytest.mark.parametrize("tdt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("case", sorted(CASES))
def test_whole_row_wgrad_matches_fp64_and_the_generic_kernel(case, tdt):
    from vlfb import hip
    hip.lib()
    hdt = hip.BF16 if tdt == torch.bfloat16 else hip.F16
    N, T, H, W, k, p = CASES[case]
    C = 64
    Cin = 256 if case.startswith("fat") else 64
    gen = torch.Generator().manual_seed(sum(map(ord, case)))
    x = q(torch.randn(N, Cin, T, H, W, generator=gen), tdt)
    dy = q(torch.randn(N, C, T, H, W, generator=gen), tdt)
    scale = torch.rand(C, generator=gen) + 0.5
    wd = torch.zeros(C, Cin, *k, dtype=torch.float64, requires_grad=True)
    (gw,) = torch.autograd.grad(F.conv3d(x.double(), wd, None, 1, p), wd, dy.double())
    ref = w_to_kernel(gw * scale.double().view(-1, 1, 1, 1, 1))
    X, G, Sc = to_nthwc(x).to(dev(), tdt), to_nthwc(dy).to(dev(), tdt), scale.to(dev())
    geom = dict(kt=k[0], kh=k[1], kw=k[2], pt=p[0], ph=p[1], pw=p[2])
    out = {}
    for algo in (hip.ALGO_AUTO, hip.ALGO_TILE128):
        for rep in range(2):
            d = hip.conv_desc(mode=hip.WGRAD, dtype=hdt, out_dtype=hip.F32, N=N, Tr=T, Hr=H, Wr=W, Ts=T, Hs=H, Ws=W,
                              Cs=Cin, Cn=C, algo=algo, **geom)
            ws = torch.empty(max(hip.conv_workspace_bytes(d) // 4, 4), device=dev(), dtype=torch.float32)
            DW = torch.full((C,) + tuple(k) + (Cin,), float("nan"), device=dev(), dtype=torch.float32)
            hip.conv_run(d, X, None, G, DW, rowscale=Sc, workspace=ws)
            torch.cuda.synchronize()
            out.setdefault(algo, []).append(DW)
    a0, a1 = out[hip.ALGO_AUTO]
    assert torch.equal(a0, a1), "run-to-run difference"
    assert rel_err(a0, ref) < 2e-5
    assert rel_err(a0, out[hip.ALGO_TILE128][0]) < 1e-5
    # accumulate on top of an existing gradient
    base = torch.randn(C, *k, Cin, generator=gen)
    DW = base.clone().to(dev())
    d = hip.conv_desc(mode=hip.WGRAD, dtype=hdt, out_dtype=hip.F32, N=N, Tr=T, Hr=H, Wr=W, Ts=T, Hs=H, Ws=W,
                      Cs=Cin, Cn=C, accumulate=1, alpha=0.5, **geom)
    ws = torch.empty(max(hip.conv_workspace_bytes(d) // 4, 4), device=dev(), dtype=torch.float32)
    hip.conv_run(d, X, None, G, DW, workspace=ws)
    assert rel_err(DW, base.double() + 0.5 * w_to_kernel(gw)) < 2e-5
