// TEST INFRASTRUCTURE (oracle/_ref recipe).  A just-big-enough stand-in for the Caffe2 headers the
// reference's caffe2_customized_ops/video/affine_nd_op.{h,cu} include, so that THOSE FILES compile
// unmodified with hipcc from where they lie under /root/reference (nothing of the reference is copied
// here).  Only what the two RunOnDevice() bodies and the kernels touch is modelled: a tensor that wraps
// a device pointer + dims, an operator base with Input()/Output(), a context that carries the stream,
// and the launch-geometry helpers with Caffe2's published values (caffe2/core/common_gpu.h:
// CAFFE_CUDA_NUM_THREADS = 128 in the early-2019 tree, CAFFE_MAXIMUM_NUM_BLOCKS = 4096,
// CUDA_1D_KERNEL_LOOP = grid-stride loop).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>
#include <stdexcept>

namespace caffe2 {

constexpr int CAFFE_CUDA_NUM_THREADS = 128;
constexpr int CAFFE_MAXIMUM_NUM_BLOCKS = 4096;
inline int CAFFE_GET_BLOCKS(const int N) {
  int b = (N + CAFFE_CUDA_NUM_THREADS - 1) / CAFFE_CUDA_NUM_THREADS;
  return b < 1 ? 1 : (b > CAFFE_MAXIMUM_NUM_BLOCKS ? CAFFE_MAXIMUM_NUM_BLOCKS : b);
}
#define CUDA_1D_KERNEL_LOOP(i, n) \
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += blockDim.x * gridDim.x)

#define CAFFE_NOT_IMPLEMENTED throw std::runtime_error("CAFFE_NOT_IMPLEMENTED")
#define USE_OPERATOR_CONTEXT_FUNCTIONS using Operator<Context>::context_
#define REGISTER_CUDA_OPERATOR(name, ...) static_assert(true, "operator registry is not modelled")

class Tensor {
 public:
  Tensor() : p_(nullptr) {}
  Tensor(void* p, std::vector<int64_t> dims) : p_(p), dims_(std::move(dims)) {}
  int64_t size() const { int64_t n = 1; for (auto d : dims_) n *= d; return n; }
  int dim32(int i) const { return (int)dims_.at(i); }
  void ResizeLike(const Tensor& o) {
    if (p_ == nullptr) throw std::runtime_error("shim tensor has no storage");
    dims_ = o.dims_;
  }
  template <typename T> const T* data() const { return static_cast<const T*>(p_); }
  template <typename T> T* mutable_data() { return static_cast<T*>(p_); }
 private:
  void* p_;
  std::vector<int64_t> dims_;
};

struct OperatorDef {};
struct Workspace {};

class CUDAContext {
 public:
  hipStream_t cuda_stream() const { return s_; }
  void set_stream(hipStream_t s) { s_ = s; }
 private:
  hipStream_t s_ = nullptr;
};
class CPUContext {};

template <class Context>
class Operator {
 public:
  Operator(const OperatorDef&, Workspace*) {}
  virtual ~Operator() {}
  virtual bool RunOnDevice() = 0;
  const Tensor& Input(int i) { return inputs.at(i); }
  Tensor* Output(int i) { return &outputs.at(i); }
  std::vector<Tensor> inputs, outputs;
  Context context_;
};

}  // namespace caffe2
