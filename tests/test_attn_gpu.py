"""The fused non-local attention kernels (csrc/vlfb_attn.hip): scores + row softmax, and the backward
dS = scale * P o (dP - rowsum(dP o P)) with dP = dY g^T computed on the fly, against fp64 torch -- on the key
counts of the model (784: 32 x 224^2 clips, 1024: test crop 256), ragged query tiles and both 16-bit types."""
import numpy as np
import pytest
import torch

from gpu_util import dev, q, rel_err

pytestmark = pytest.mark.gpu

CASES = [(2, 200, 784, 256), (1, 128, 1024, 512), (3, 64, 520, 64), (1, 3136, 784, 256)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,L1,L2,Ci", CASES)
def test_fused_scores_softmax_forward_and_backward(B, L1, L2, Ci, dtype):
    from vlfb import hip
    hip.lib()
    code = hip.dtype_code(dtype)
    assert hip.lib().vlfb_attn_scores_supported(code, L1, L2, Ci) & hip.ATTN_CAN_RUN
    gen = torch.Generator().manual_seed(L1 * 7 + L2)
    theta = q(torch.randn(B, L1, Ci, generator=gen), dtype)
    phi = q(torch.randn(B, L2, Ci, generator=gen), dtype)
    scale = Ci ** -0.5
    th, ph = theta.to(dev(), dtype), phi.to(dev(), dtype)
    P = torch.full((B, L1, L2), float("nan"), device=dev(), dtype=dtype)
    hip.call("vlfb_attn_scores_fwd", hip.ptr(th), hip.ptr(ph), hip.ptr(P), code, B, L1, L2, Ci, scale)
    torch.cuda.synchronize()
    ref = torch.softmax(scale * torch.bmm(theta.double(), phi.double().transpose(1, 2)), dim=2)
    tol = 6e-3 if dtype == torch.bfloat16 else 8e-4
    assert rel_err(P.float(), ref) < tol
    assert float((P.float().sum(dim=2) - 1).abs().max()) < (2e-2 if dtype == torch.bfloat16 else 3e-3)
    # the composed path (GEMM -> fp32 scores -> softmax kernel) gives the same probabilities up to the last bit or two
    S = torch.empty(B, L1, L2, device=dev(), dtype=torch.float32)
    d = hip.conv_desc(mode=hip.FPROP, dtype=code, out_dtype=hip.F32, N=1, Tr=1, Hr=1, Wr=L1, Ts=1, Hs=1, Ws=L1, batch=B,
                      Cs=Ci, Cn=L2, a_bstride=L1 * Ci, b_bstride=L2 * Ci, o_bstride=L1 * L2)
    hip.conv_run(d, th, ph, None, S)
    P2 = torch.empty_like(P)
    hip.call("vlfb_softmax_fwd", hip.ptr(S), hip.ptr(P2), code, B * L1, L2, scale)
    torch.cuda.synchronize()
    assert rel_err(P.float(), P2.float()) < (2e-3 if dtype == torch.bfloat16 else 3e-4)
    # ---- backward ---------------------------------------------------------------------------------------------
    dY = q(torch.randn(B, L1, Ci, generator=gen), dtype)
    g = q(torch.randn(B, L2, Ci, generator=gen), dtype)
    dS = torch.full((B, L1, L2), float("nan"), device=dev(), dtype=dtype)
    dYd, gd = dY.to(dev(), dtype), g.to(dev(), dtype)       # named: a temporary would be freed before the launch runs
    hip.call("vlfb_attn_scores_bwd", hip.ptr(dYd), hip.ptr(gd), hip.ptr(P), hip.ptr(dS), code, B, L1, L2, Ci, scale)
    torch.cuda.synchronize()
    Pd = P.double().cpu()                                   # the probabilities the kernel saw
    dP = torch.bmm(dY.double(), g.double().transpose(1, 2))
    want = scale * Pd * (dP - (dP * Pd).sum(dim=2, keepdim=True))
    assert rel_err(dS.float(), want) < (6e-3 if dtype == torch.bfloat16 else 8e-4)


def test_unsupported_shapes_are_refused():
    from vlfb import hip
    hip.lib()
    assert hip.lib().vlfb_attn_scores_supported(hip.F32, 3136, 784, 256) == 0        # parity path keeps fp32 scores
    assert hip.lib().vlfb_attn_scores_supported(hip.BF16, 6272, 1568, 512) == 0      # 64-frame clips: 1568 keys
    assert hip.lib().vlfb_attn_scores_supported(hip.BF16, 784, 196, 256) == 0
    t = torch.zeros(64 * 1568, device=dev(), dtype=torch.bfloat16)
    with pytest.raises(hip.VlfbError):
        hip.call("vlfb_attn_scores_fwd", hip.ptr(t), hip.ptr(t), hip.ptr(t), hip.BF16, 1, 64, 1568, 64, 1.0)
