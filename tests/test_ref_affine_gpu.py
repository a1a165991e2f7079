"""AffineNd pinned against REFERENCE-BUILT code: oracle/_ref/libref_affine_nd.so is the reference's own
caffe2_customized_ops/video/affine_nd_op.cu (kernels :31-58 and both RunOnDevice bodies :61-107)
compiled unmodified by oracle/build_ref.py.  Checked here:
  * vlfb_affine_nd_fwd / vlfb_affine_nd_bwd (the drop-in operator of include/vlfb.h) == the reference
    operator, BIT-EXACT (fp32, same fma contraction),
  * the CPU oracle's `_affine` restatement == the reference operator to 1 ulp (torch-CPU rounds the
    product before the add, the GPU build fuses it), and its fp64 form to fp32 round-off.
The library is test infrastructure: only tests/ load it."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_affine_nd.so")

SHAPES = [(2, 64, 8 * 14 * 14), (1, 256, 4 * 7 * 7), (3, 2048, 1), (2, 3, 5), (1, 1, 4099), (4, 512, 16 * 7 * 7 + 3)]


def _ref():
    assert os.path.exists(REF_SO), ("%s is missing: run `python oracle/build_ref.py` where /root/reference is "
                                    "mounted (build() does) -- the prebuilt library travels to the GPU box" % REF_SO)
    lib = C.CDLL(REF_SO)
    P, LL = C.c_void_p, C.c_longlong
    lib.ref_affine_nd_fwd.argtypes = [P, P, P, P, LL, LL, LL, P]
    lib.ref_affine_nd_bwd.argtypes = [P, P, P, LL, LL, LL, P]
    lib.ref_affine_nd_fwd.restype = lib.ref_affine_nd_bwd.restype = C.c_int
    return lib


def test_reference_library_exports_both_entry_points():
    """CPU-side: the reference-built checker exists and exports what the GPU tests bind (no compute)"""
    if not os.path.exists(REF_SO) and not os.path.exists("/root/reference"):
        pytest.skip("neither the prebuilt library nor /root/reference is available here")
    lib = _ref()
    assert lib.ref_affine_nd_fwd and lib.ref_affine_nd_bwd


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES)
def test_affine_nd_matches_the_reference_built_operator_bit_exactly(shape):
    from vlfb import hip
    from oracle import model as om
    n, c, inner = shape
    g = torch.Generator().manual_seed(1234 + n * 7 + c)
    x = torch.randn(n, c, inner, generator=g)
    s = torch.rand(c, generator=g) + 0.5
    b = torch.randn(c, generator=g) * 0.1
    xd, sd, bd = x.cuda(), s.cuda(), b.cuda()
    ref = _ref()
    stream = torch.cuda.current_stream().cuda_stream
    y_ref = torch.full_like(xd, float("nan"))
    assert ref.ref_affine_nd_fwd(xd.data_ptr(), sd.data_ptr(), bd.data_ptr(), y_ref.data_ptr(), n, c, inner, stream) == 0
    y = torch.full_like(xd, float("nan"))
    hip.call("vlfb_affine_nd_fwd", hip.ptr(xd), hip.ptr(sd), hip.ptr(bd), hip.ptr(y), n, c, inner)
    dx_ref = torch.full_like(xd, float("nan"))
    assert ref.ref_affine_nd_bwd(xd.data_ptr(), sd.data_ptr(), dx_ref.data_ptr(), n, c, inner, stream) == 0
    dx = torch.full_like(xd, float("nan"))
    hip.call("vlfb_affine_nd_bwd", hip.ptr(xd), hip.ptr(sd), hip.ptr(dx), n, c, inner)
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref), "vlfb_affine_nd_fwd differs from the reference-built AffineNdOp"
    assert torch.equal(dx, dx_ref), "vlfb_affine_nd_bwd differs from the reference-built AffineNdGradientOp"
    # the oracle's restatement (what every model-level parity test leans on)
    P = {"p_s": s, "p_b": b}
    y_or = om._affine(x.view(n, c, inner, 1, 1), P, "p").view(n, c, inner)
    ulp = np.spacing(np.abs(y_ref.cpu().numpy()).astype(np.float32)) + np.spacing(np.abs(b.numpy()).astype(np.float32))[None, :, None]
    assert np.all(np.abs(y_or.numpy() - y_ref.cpu().numpy()) <= ulp), "fp32 oracle _affine is more than 1 ulp away"
    P64 = {"p_s": s.double(), "p_b": b.double()}
    y64 = om._affine(x.double().view(n, c, inner, 1, 1), P64, "p").view(n, c, inner)
    err = (y64 - y_ref.cpu().double()).abs().max() / y64.abs().max()
    assert float(err) < 1e-6
