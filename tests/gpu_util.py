"""Helpers shared by the GPU kernel tests: layout conversion and error metrics."""
import torch

from vlfb import hip

DTYPES = [torch.float32, torch.bfloat16, torch.float16]
# relative-L2 bars: the fp32 path (exact-fp32 MFMA) is held to the north-star 1e-3 with margin;
# the bf16 path rounds every stored tensor to 8 mantissa bits (2^-9 = 2e-3 per rounding), fp16 to 11 (2^-12).
TOL = {torch.float32: 2e-5, torch.bfloat16: 1e-2, torch.float16: 1.5e-3}


def dev():
    return torch.device("cuda:0")


def rel_err(got, ref):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    den = ref.norm().item()
    return (got - ref).norm().item() / (den if den > 0 else 1.0)


def q(t, dtype):
    """round a fp32 CPU tensor through `dtype` (so the reference sees what the kernel sees)"""
    return t.to(dtype).to(torch.float32)


def to_nthwc(x):  # (N,C,T,H,W) -> (N,T,H,W,C)
    return x.permute(0, 2, 3, 4, 1).contiguous()


def to_ncthw(x):  # (N,T,H,W,C) -> (N,C,T,H,W)
    return x.permute(0, 4, 1, 2, 3).contiguous()


def w_to_kernel(w):  # (Cout,Cin,kt,kh,kw) -> (Cout, kt,kh,kw, Cin)
    return w.permute(0, 2, 3, 4, 1).contiguous()


def w_from_kernel(wk):  # (Cout,kt,kh,kw,Cin) -> (Cout,Cin,kt,kh,kw)
    return wk.permute(0, 4, 1, 2, 3).contiguous()
