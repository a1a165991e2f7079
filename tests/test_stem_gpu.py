"""The packed stem FPROP (conv1, resnet_video.py:169-179) runs as a direct convolution on the matrix cores
(csrc/vlfb_stem.hip: whole output rows per wave, raw input rows staged in LDS) when the output rows are 112 wide.
Same k order, same MFMA, same epilogue order as the 128x64 tiled kernel (algo = TILE128): the outputs must be
BIT-IDENTICAL, also with row blocks cut by the frame height, temporal padding on both sides, bias + ReLU + alpha."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from vlfb import hip   # noqa: E402


def _run(dtype, N, T, H, algo, relu, alpha, bias):
    W, Cout = 224, 64
    gen = torch.Generator().manual_seed(17)
    code = hip.dtype_code(dtype)
    x = torch.randn(N, 3, T, H, W, generator=gen).to(dtype)
    w = (torch.randn(Cout, 3, 5, 7, 7, generator=gen) * 0.05).to(dtype)
    WP = W + 8
    X4 = torch.empty(N, T, H, WP, 4, device="cuda", dtype=dtype)
    xs = x.float().cuda().contiguous()
    hip.call("vlfb_ncthw_to_nthwc_wpad", hip.ptr(xs), hip.ptr(X4), code, N, 3, T * H, W, 4, 4, WP)
    wp = torch.zeros(Cout, 5, 7, 8, 4)
    wp[:, :, :, :7, :3] = w.float().permute(0, 2, 3, 4, 1)
    Bw = wp.to(dtype).cuda()
    Ho, Wo = (H + 6 - 7) // 2 + 1, 112
    O = torch.full((N, T, Ho, Wo, Cout), float("nan"), device="cuda", dtype=dtype)
    b = (torch.randn(Cout, generator=gen) * 0.3).cuda() if bias else None
    desc = hip.conv_desc(mode=hip.FPROP, dtype=code, out_dtype=code, N=N, Tr=T, Hr=Ho, Wr=Wo, Ts=T, Hs=H, Ws=WP, Cs=4,
                         Cn=Cout, pack_w=8, kt=5, kh=7, kw=7, st=1, sh=2, sw=2, pt=2, ph=3, pw=3 - 4, dt=1, dh=1, dw=1,
                         relu=int(relu), alpha=alpha, bias_mode=hip.BIAS_COL if bias else hip.BIAS_NONE, algo=algo)
    hip.conv_run(desc, X4, Bw, None, O, bias=b)
    torch.cuda.synchronize()
    return O, x, w, b


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(2, 5, 50), (1, 3, 224), (3, 2, 17)], ids=["cut_last_block", "full_frame", "two_blocks"])
def test_direct_stem_is_bit_identical_to_the_tiled_kernel(dtype, shape):
    N, T, H = shape
    a, x, w, b = _run(dtype, N, T, H, hip.ALGO_AUTO, True, 0.75, True)
    t, _, _, _ = _run(dtype, N, T, H, hip.ALGO_TILE128, True, 0.75, True)
    assert not torch.isnan(a.float()).any()
    assert torch.equal(a.view(torch.int16), t.view(torch.int16))
    assert float(a.float().abs().sum()) > 0 and float((a.float() == 0).float().mean()) > 0.2     # the ReLU did cut
    # and against the convolution itself
    ref = torch.nn.functional.conv3d(x.double(), w.double(), None, (1, 2, 2), (2, 3, 3)) * 0.75 + b.double().cpu().view(1, -1, 1, 1, 1)
    ref = torch.relu(ref).permute(0, 2, 3, 4, 1)
    err = float((a.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < (6e-3 if dtype == torch.bfloat16 else 1e-3), err


def test_direct_stem_without_bias_and_relu():
    a, _, _, _ = _run(torch.bfloat16, 1, 4, 30, hip.ALGO_AUTO, False, 1.0, False)
    t, _, _, _ = _run(torch.bfloat16, 1, 4, 30, hip.ALGO_TILE128, False, 1.0, False)
    assert torch.equal(a.view(torch.int16), t.view(torch.int16)) and float((a.float() < 0).float().mean()) > 0.3
