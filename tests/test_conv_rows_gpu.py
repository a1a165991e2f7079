"""1x3x3 convolutions with 64 -> 64 channels (res2 branch2b, resnet_helper.py:53-63) run as a direct convolution
with the weights resident in LDS and the input rows rolling through a ring (csrc/vlfb_conv_rows.hip) -- FPROP and
the unit-stride DGRAD.  k order, MFMA and epilogue order are those of the tiled kernel (algo = TILE128), so the
outputs must be BIT-IDENTICAL: one wrong ring slot, halo pixel, swizzle key or mirrored tap changes bits.  Cases:
rows narrower than one fragment up to 62 positions (the widest the row pitch holds), frame heights that are not a
multiple of the four-row step (cut last step, one-step frames), several frames per workgroup walk, every epilogue
(bias + residual + ReLU; alpha alone; residual + mask), bf16 and fp16.  Three runs each: a race between the ring DMA
and the fragment reads would show up as run-to-run differences."""
import math

import pytest
import torch
import torch.nn.functional as F

from gpu_util import dev, q, rel_err, to_ncthw, to_nthwc, w_to_kernel

pytestmark = pytest.mark.gpu

S, D = (1, 1, 1), (1, 1, 1)
SPATIAL = ((1, 3, 3), (0, 1, 1))       # res2 branch2b
TEMPORAL = ((3, 1, 1), (1, 0, 0))      # res2_0 branch2a: the same walk with t in the role of h
# name: (N, T, H, W, (kernel, pad))
CASES = {
    "res2_row": (1, 2, 9, 56, SPATIAL),          # 56-wide rows (three and a half fragments), cut last step
    "widest": (1, 1, 6, 62, SPATIAL),            # the widest row the 64-pixel pitch holds
    "narrow": (2, 3, 4, 7, SPATIAL),             # less than one fragment per row, exactly one step per frame
    "tall": (1, 1, 23, 14, SPATIAL),             # six steps: the ring wraps twice
    "one_row": (3, 2, 1, 20, SPATIAL),           # frames of a single row
    "many_frames": (2, 150, 5, 16, SPATIAL),     # 300 frames: several frames per workgroup
    "temporal_res2_row": (1, 9, 3, 56, TEMPORAL),      # t walks, cut last step
    "temporal_long": (2, 23, 2, 14, TEMPORAL),         # six steps along t
    "temporal_many_slabs": (3, 5, 110, 16, TEMPORAL),  # 330 (n, h) slabs: several per workgroup
    "temporal_single_frame": (1, 1, 4, 20, TEMPORAL),  # T = 1: both temporal neighbours are padding
}


@pytest.mark.parametrize("tdt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("case", sorted(CASES))
def test_direct_rows_kernel_is_bit_identical_to_tile128_and_matches_fp64(case, tdt):
    from vlfb import hip
    hip.lib()
    hdt = hip.BF16 if tdt == torch.bfloat16 else hip.F16
    tol = 1e-2 if tdt == torch.bfloat16 else 2e-3
    if tdt == torch.float16 and case not in ("res2_row", "tall", "temporal_long"):
        pytest.skip("fp16 instances are the same template: a subset is enough")
    N, T, H, W, (K, P) = CASES[case]
    GEOM = dict(kt=K[0], kh=K[1], kw=K[2], st=1, sh=1, sw=1, pt=P[0], ph=P[1], pw=P[2], dt=1, dh=1, dw=1)
    C = 64
    gen = torch.Generator().manual_seed(sum(map(ord, case)))
    x = q(torch.randn(N, C, T, H, W, generator=gen), tdt)
    w = q(torch.randn(C, C, *K, generator=gen) / math.sqrt(C * K[0] * K[1] * K[2]), tdt)
    bias = torch.randn(C, generator=gen)
    res = q(torch.randn(N, C, T, H, W, generator=gen), tdt)
    A = to_nthwc(x).to(dev(), tdt)
    Bw = w_to_kernel(w).to(dev(), tdt)
    R = to_nthwc(res).to(dev(), tdt)
    bg = bias.to(dev())
    y_ref = torch.relu(F.conv3d(x.double(), w.double(), bias.double(), S, P, D) + res.double())
    outs = {}
    for algo in (hip.ALGO_TILE128, hip.ALGO_AUTO):
        for rep in range(3 if algo == hip.ALGO_AUTO else 1):
            O = torch.full((N, T, H, W, C), float("nan"), device=dev(), dtype=tdt)
            desc = hip.conv_desc(mode=hip.FPROP, dtype=hdt, out_dtype=hdt, N=N, Tr=T, Hr=H, Wr=W, Ts=T, Hs=H, Ws=W, Cs=C, Cn=C,
                                 relu=1, bias_mode=hip.BIAS_COL, algo=algo, **GEOM)
            hip.conv_run(desc, A, Bw, None, O, bias=bg, R=R)
            O2 = torch.full((N, T, H, W, C), float("nan"), device=dev(), dtype=tdt)      # alpha alone, as the engine's plain convs
            desc = hip.conv_desc(mode=hip.FPROP, dtype=hdt, out_dtype=hdt, N=N, Tr=T, Hr=H, Wr=W, Ts=T, Hs=H, Ws=W, Cs=C, Cn=C,
                                 alpha=0.75, algo=algo, **GEOM)
            hip.conv_run(desc, A, Bw, None, O2)
            torch.cuda.synchronize()
            outs.setdefault(algo, []).append((O.clone(), O2.clone()))
    ref, ref2 = outs[hip.ALGO_TILE128][0]
    assert not torch.isnan(ref.float()).any()
    for O, O2 in outs[hip.ALGO_AUTO]:
        assert torch.equal(O.view(torch.int16), ref.view(torch.int16)), "fprop differs from the tiled kernel"
        assert torch.equal(O2.view(torch.int16), ref2.view(torch.int16)), "fprop (alpha alone) differs"
    assert rel_err(to_ncthw(ref.float()), y_ref) < tol
    # ---- dgrad with residual-add + mask epilogue, and with the mask alone (res2 branch2b) -----------------------
    dy = q(torch.randn(N, C, T, H, W, generator=gen), tdt)
    mask_src = q(torch.randn(N, C, T, H, W, generator=gen), tdt)
    add_src = q(torch.randn(N, C, T, H, W, generator=gen), tdt)
    G = to_nthwc(dy).to(dev(), tdt)
    Wd = w.permute(1, 2, 3, 4, 0).contiguous().to(dev(), tdt)
    Rm, Mm = to_nthwc(add_src).to(dev(), tdt), to_nthwc(mask_src).to(dev(), tdt)
    got = {}
    for algo in (hip.ALGO_TILE128, hip.ALGO_AUTO):
        for rep in range(3 if algo == hip.ALGO_AUTO else 1):
            DX = torch.full((N, T, H, W, C), float("nan"), device=dev(), dtype=tdt)
            desc = hip.conv_desc(mode=hip.DGRAD, dtype=hdt, out_dtype=hdt, N=N, Tr=T, Hr=H, Wr=W, Ts=T, Hs=H, Ws=W, Cs=C, Cn=C,
                                 algo=algo, **GEOM)
            hip.conv_run(desc, G, Wd, None, DX, R=Rm, mask=Mm)
            DX2 = torch.full((N, T, H, W, C), float("nan"), device=dev(), dtype=tdt)
            desc = hip.conv_desc(mode=hip.DGRAD, dtype=hdt, out_dtype=hdt, N=N, Tr=T, Hr=H, Wr=W, Ts=T, Hs=H, Ws=W, Cs=C, Cn=C,
                                 alpha=0.5, algo=algo, **GEOM)
            hip.conv_run(desc, G, Wd, None, DX2, mask=Mm)
            torch.cuda.synchronize()
            got.setdefault(algo, []).append((DX, DX2))
    for DX, DX2 in got[hip.ALGO_AUTO]:
        assert torch.equal(DX.view(torch.int16), got[hip.ALGO_TILE128][0][0].view(torch.int16)), "dgrad differs"
        assert torch.equal(DX2.view(torch.int16), got[hip.ALGO_TILE128][0][1].view(torch.int16)), "dgrad (mask, alpha) differs"
    xd = x.double().requires_grad_(True)
    gx, = torch.autograd.grad(F.conv3d(xd, w.double(), None, S, P, D), xd, dy.double())
    dx_ref = torch.where(mask_src.double() > 0, gx + add_src.double(), torch.zeros_like(gx))
    assert rel_err(to_ncthw(got[hip.ALGO_AUTO][0][0].float()), dx_ref) < tol
