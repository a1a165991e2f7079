// TEST INFRASTRUCTURE (oracle/_ref recipe, see oracle/build_ref.py).  C entry points around the
// REFERENCE's own AffineNd operator: this translation unit #includes
// /root/reference/caffe2_customized_ops/video/affine_nd_op.cu where it lies (its two kernels,
// affine_nd_op.cu:31-58, and both RunOnDevice() bodies, :61-107, compile unmodified against the
// mini-Caffe2 in oracle/ref_shim/) and drives AffineNdOp / AffineNdGradientOp on caller-owned device
// buffers.  Built with hipcc's default floating-point contraction, i.e. the same a*b+c -> fma fusion
// nvcc applies to the reference by default.
#include "video/affine_nd_op.cu"

extern "C" int ref_affine_nd_fwd(const float* x, const float* scale, const float* bias, float* y,
                                 long long n, long long c, long long inner, void* stream) {
  try {
    caffe2::OperatorDef def;
    caffe2::AffineNdOp<float, caffe2::CUDAContext> op(def, nullptr);
    op.context_.set_stream((hipStream_t)stream);
    op.inputs = {caffe2::Tensor((void*)x, {n, c, inner}), caffe2::Tensor((void*)scale, {c}),
                 caffe2::Tensor((void*)bias, {c})};
    op.outputs = {caffe2::Tensor((void*)y, {n, c, inner})};
    if (!op.RunOnDevice()) return 2;
    return hipGetLastError() == hipSuccess ? 0 : 3;
  } catch (...) { return 1; }
}

extern "C" int ref_affine_nd_bwd(const float* dy, const float* scale, float* dx,
                                 long long n, long long c, long long inner, void* stream) {
  try {
    caffe2::OperatorDef def;
    caffe2::AffineNdGradientOp<float, caffe2::CUDAContext> op(def, nullptr);
    op.context_.set_stream((hipStream_t)stream);
    op.inputs = {caffe2::Tensor((void*)scale, {c}), caffe2::Tensor((void*)dy, {n, c, inner})};
    op.outputs = {caffe2::Tensor((void*)dx, {n, c, inner})};
    if (!op.RunOnDevice()) return 2;
    return hipGetLastError() == hipSuccess ? 0 : 3;
  } catch (...) { return 1; }
}
