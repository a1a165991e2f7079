// SpatialBN over channels-last rows (x[rows][C], rows = N*T*H*W) -- the graphs with MODEL.USE_AFFINE False /
// NONLOCAL.USE_BN True (model_builder_video.py:176-197 Conv3dBN, resnet_video.py:185-188, nonlocal_helper.py:146-155).
// The operator itself is Caffe2's (caffe2/operators/spatial_batch_norm_op.{h,cc}, absent from /root/reference); what is
// restated here is its published algorithm:
//   train:  mu = mean_rows(x), var = mean_rows((x - mu)^2)  (biased), inv_std = 1 / sqrt(var + eps)
//           y = (x - mu) * inv_std * gamma + beta
//           running_mean = momentum * running_mean + (1 - momentum) * mu
//           running_var  = momentum * running_var  + (1 - momentum) * var * rows / (rows - 1)      (unbiased)
//   test:   y = (x - running_mean) / sqrt(running_var + eps) * gamma + beta
//   grad:   dbeta = sum_rows(dy), dgamma = sum_rows(dy * xhat), dx = gamma * inv_std * (dy - dbeta / rows - xhat * dgamma / rows)
// HBM-bound, three passes forward (moments, apply) and backward (two sums, apply).  All reductions are two-stage and
// ordered (slab partials, then one thread per channel over the slabs): no atomics, the same bits every run.  The moments
// are taken about a per-channel pivot (the first row) so that sum((x - p)^2) - sum(x - p)^2 / rows does not cancel when
// |mean| >> std.
#include "vlfb_common.h"

namespace vlfb {
namespace {

constexpr int kCL = 8;                  // 16-byte chunk lanes per block
constexpr int kRL = 256 / kCL;          // row lanes

// partial[slab][0 | 1][C]: sums of (a) d = x - pivot, d^2 (MOMENTS) or (b) dy, dy * xhat (GRADS)
template <typename T, bool GRADS>
__global__ void bn_partial_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ mean,
                                  const float* __restrict__ inv_std, long long rows, int C, float* __restrict__ partial) {
  constexpr int V = Vec16<T>::N;
  __shared__ float red[2][kRL][kCL * 8 + 1];
  const int cl = threadIdx.x % kCL, rl = threadIdx.x / kCL;
  const int c0 = (blockIdx.x * kCL + cl) * V;
  const long long per = (rows + gridDim.y - 1) / gridDim.y;
  const long long r0 = (long long)blockIdx.y * per, r1 = min(rows, r0 + per);
  float a0[V], a1[V], piv[V], sc[V];
#pragma unroll
  for (int k = 0; k < V; ++k) { a0[k] = a1[k] = 0.f; piv[k] = 0.f; sc[k] = 1.f; }
  if (c0 < C) {
    if (GRADS) {
#pragma unroll
      for (int k = 0; k < V; ++k) { piv[k] = mean[c0 + k]; sc[k] = inv_std[c0 + k]; }
    } else {
      Vec16<T>::load(x + c0, piv);          // row 0: the pivot of every slab
    }
    for (long long r = r0 + rl; r < r1; r += kRL) {
      float v[V];
      Vec16<T>::load(x + r * C + c0, v);
      if (GRADS) {
        float g[V];
        Vec16<T>::load(dy + r * C + c0, g);
#pragma unroll
        for (int k = 0; k < V; ++k) { a0[k] += g[k]; a1[k] += g[k] * ((v[k] - piv[k]) * sc[k]); }
      } else {
#pragma unroll
        for (int k = 0; k < V; ++k) { const float d = v[k] - piv[k]; a0[k] += d; a1[k] += d * d; }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < V; ++k) { red[0][rl][cl * 8 + k] = a0[k]; red[1][rl][cl * 8 + k] = a1[k]; }
  __syncthreads();
  if (rl < 2 * V && c0 < C) {             // row lanes 0 .. V-1 fold sum 0 of channel rl, V .. 2V-1 sum 1
    const int which = rl / V, k = rl % V;
    float s = 0.f;
    for (int j = 0; j < kRL; ++j) s += red[which][j][cl * 8 + k];
    partial[((long long)blockIdx.y * 2 + which) * C + c0 + k] = s;
  }
}

// one thread per channel: fold the slabs in order, then
//   TRAIN: mean / inv_std (saved), running statistics, and the fused apply coefficients a = gamma * inv_std, b = beta - mean * a
template <typename T>
__global__ void bn_finish_train_kernel(const float* __restrict__ partial, int slabs, const T* __restrict__ x, long long rows,
                                       int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                       float* __restrict__ running_mean, float* __restrict__ running_var,
                                       float* __restrict__ save_mean, float* __restrict__ save_inv_std,
                                       float* __restrict__ coef, float eps, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int j = 0; j < slabs; ++j) { s += partial[((long long)j * 2) * C + c]; q += partial[((long long)j * 2 + 1) * C + c]; }
  const double piv = (double)Elem<T>::ld(x + c);
  const double n = (double)rows;
  const double md = s / n;                                   // mean of (x - pivot)
  double var = q / n - md * md;
  if (var < 0.0) var = 0.0;
  const float mu = (float)(piv + md);
  const float istd = (float)(1.0 / sqrt(var + (double)eps));
  save_mean[c] = mu;
  save_inv_std[c] = istd;
  const double unbiased = rows > 1 ? var * n / (n - 1.0) : var;
  running_mean[c] = momentum * running_mean[c] + (1.f - momentum) * mu;
  running_var[c] = momentum * running_var[c] + (1.f - momentum) * (float)unbiased;
  const float a = gamma[c] * istd;
  coef[c] = a;
  coef[C + c] = beta[c] - mu * a;
}

__global__ void bn_fold_test_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                    float* __restrict__ coef, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float a = gamma[c] / sqrtf(running_var[c] + eps);
  coef[c] = a;
  coef[C + c] = beta[c] - running_mean[c] * a;
}

// GRADS: dbeta, dgamma out; coefficients of dx = ca * dy + cb * x + cc with
//   ca = gamma * inv_std, cb = -ca * inv_std * dgamma / rows, cc = -ca * dbeta / rows - cb * mean
__global__ void bn_finish_grad_kernel(const float* __restrict__ partial, int slabs, long long rows, int C,
                                      const float* __restrict__ gamma, const float* __restrict__ mean,
                                      const float* __restrict__ inv_std, float* __restrict__ dgamma,
                                      float* __restrict__ dbeta, float* __restrict__ coef, float gscale) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double sb = 0.0, sg = 0.0;
  for (int j = 0; j < slabs; ++j) { sb += partial[((long long)j * 2) * C + c]; sg += partial[((long long)j * 2 + 1) * C + c]; }
  if (dbeta) dbeta[c] = (float)sb * gscale;
  if (dgamma) dgamma[c] = (float)sg * gscale;
  const double n = (double)rows;
  const double ca = (double)gamma[c] * inv_std[c];
  const double cb = -ca * inv_std[c] * sg / n;
  coef[c] = (float)ca;
  coef[C + c] = (float)cb;
  coef[2 * C + c] = (float)(-ca * sb / n - cb * mean[c]);
}

// y = a[c] * x + b[c] (forward: u = nullptr) or dx = ca[c] * dy + cb[c] * x + cc[c] (backward: u = dy)
template <typename T, bool BWD>
__global__ void bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ u, T* __restrict__ y,
                                const float* __restrict__ coef, long long rows, int C) {
  constexpr int V = Vec16<T>::N;
  const long long chunks = rows * (C / V);
  const int cpr = C / V;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % cpr) * V;
    float v[V], o[V];
    Vec16<T>::load(x + i * V, v);
    if (BWD) {
      float g[V];
      Vec16<T>::load(u + i * V, g);
#pragma unroll
      for (int k = 0; k < V; ++k) o[k] = coef[c0 + k] * g[k] + coef[C + c0 + k] * v[k] + coef[2 * C + c0 + k];
    } else {
#pragma unroll
      for (int k = 0; k < V; ++k) o[k] = coef[c0 + k] * v[k] + coef[C + c0 + k];
    }
    Vec16<T>::store(y + i * V, o);
  }
}

int slabs_for(int64_t rows, int64_t C, int v) {
  const int cblocks = (int)((C / v + kCL - 1) / kCL);
  int64_t slabs = (rows + 1023) / 1024;
  const int64_t want = (2048 + cblocks - 1) / cblocks;
  if (slabs > want) slabs = want;
  if (slabs < 1) slabs = 1;
  return (int)slabs;
}

int check(int dtype, int64_t rows, int64_t C, const void* ws, int64_t ws_bytes, const char* who) {
  VLFB_REQUIRE(dtype_ok(dtype), "%s: bad dtype", who);
  const int v = dtype == VLFB_F32 ? 4 : 8;
  VLFB_REQUIRE(rows > 0 && C > 0 && C % v == 0, "%s: rows > 0 and C a multiple of %d expected", who, v);
  VLFB_REQUIRE(C <= INT32_MAX && rows * C < (1ll << 40), "%s: tensor too large", who);
  VLFB_REQUIRE(ws && ws_bytes >= vlfb_bn_workspace_bytes(dtype, rows, C), "%s: workspace too small (vlfb_bn_workspace_bytes)", who);
  return VLFB_OK;
}

}  // namespace
}  // namespace vlfb

using namespace vlfb;

extern "C" int64_t vlfb_bn_workspace_bytes(int dtype, int64_t rows, int64_t C) {
  if (!dtype_ok(dtype) || rows <= 0 || C <= 0) return -1;
  const int v = dtype == VLFB_F32 ? 4 : 8;
  return ((int64_t)slabs_for(rows, C, v) * 2 + 3) * C * 4;        // slab partials + three coefficient rows
}

extern "C" int vlfb_bn_fwd(const void* x, void* y, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float* save_mean, float* save_inv_std, void* workspace,
                           int64_t workspace_bytes, int dtype, int64_t rows, int64_t C, float eps, float momentum,
                           int is_test, vlfb_stream_t stream) {
  VLFB_REQUIRE(x && y && gamma && beta && running_mean && running_var, "bn_fwd: null argument");
  VLFB_REQUIRE(is_test || (save_mean && save_inv_std), "bn_fwd: training needs save_mean / save_inv_std");
  if (int rc = check(dtype, rows, C, workspace, workspace_bytes, "bn_fwd")) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int v = dtype == VLFB_F32 ? 4 : 8;
  const int slabs = slabs_for(rows, C, v);
  float* partial = (float*)workspace;
  float* coef = partial + (int64_t)slabs * 2 * C;
  const dim3 cgrid((unsigned)((C + 255) / 256));
  if (is_test) {
    hipLaunchKernelGGL(bn_fold_test_kernel, cgrid, dim3(256), 0, s, (int)C, gamma, beta, running_mean, running_var, coef, eps);
  } else {
    const dim3 grid((unsigned)((C / v + kCL - 1) / kCL), (unsigned)slabs);
    if (dtype == VLFB_F32) {
      hipLaunchKernelGGL((bn_partial_kernel<float, false>), grid, dim3(256), 0, s, (const float*)x, nullptr, nullptr, nullptr, (long long)rows, (int)C, partial);
      hipLaunchKernelGGL(bn_finish_train_kernel<float>, cgrid, dim3(256), 0, s, partial, slabs, (const float*)x, (long long)rows, (int)C, gamma, beta, running_mean, running_var, save_mean, save_inv_std, coef, eps, momentum);
    } else {
      VLFB_WITH_T16(dtype, hipLaunchKernelGGL((bn_partial_kernel<T16, false>), grid, dim3(256), 0, s, (const T16*)x, nullptr, nullptr, nullptr, (long long)rows, (int)C, partial);
                    hipLaunchKernelGGL(bn_finish_train_kernel<T16>, cgrid, dim3(256), 0, s, partial, slabs, (const T16*)x, (long long)rows, (int)C, gamma, beta, running_mean, running_var, save_mean, save_inv_std, coef, eps, momentum));
    }
  }
  const int agrid = grid_for(rows * (C / v), 256);
  if (dtype == VLFB_F32)
    hipLaunchKernelGGL((bn_apply_kernel<float, false>), dim3(agrid), dim3(256), 0, s, (const float*)x, nullptr, (float*)y, coef, (long long)rows, (int)C);
  else
    VLFB_WITH_T16(dtype, hipLaunchKernelGGL((bn_apply_kernel<T16, false>), dim3(agrid), dim3(256), 0, s, (const T16*)x, nullptr, (T16*)y, coef, (long long)rows, (int)C));
  return check_launch("bn_fwd");
}

extern "C" int vlfb_bn_bwd(const void* dy, const void* x, const float* gamma, const float* save_mean,
                           const float* save_inv_std, void* dx, float* dgamma, float* dbeta, void* workspace,
                           int64_t workspace_bytes, int dtype, int64_t rows, int64_t C, float grad_scale,
                           vlfb_stream_t stream) {
  VLFB_REQUIRE(dy && x && gamma && save_mean && save_inv_std, "bn_bwd: null argument");
  VLFB_REQUIRE(dx || dgamma || dbeta, "bn_bwd: nothing to compute");
  if (int rc = check(dtype, rows, C, workspace, workspace_bytes, "bn_bwd")) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int v = dtype == VLFB_F32 ? 4 : 8;
  const int slabs = slabs_for(rows, C, v);
  float* partial = (float*)workspace;
  float* coef = partial + (int64_t)slabs * 2 * C;
  const dim3 grid((unsigned)((C / v + kCL - 1) / kCL), (unsigned)slabs);
  const dim3 cgrid((unsigned)((C + 255) / 256));
  if (dtype == VLFB_F32)
    hipLaunchKernelGGL((bn_partial_kernel<float, true>), grid, dim3(256), 0, s, (const float*)x, (const float*)dy, save_mean, save_inv_std, (long long)rows, (int)C, partial);
  else
    VLFB_WITH_T16(dtype, hipLaunchKernelGGL((bn_partial_kernel<T16, true>), grid, dim3(256), 0, s, (const T16*)x, (const T16*)dy, save_mean, save_inv_std, (long long)rows, (int)C, partial));
  hipLaunchKernelGGL(bn_finish_grad_kernel, cgrid, dim3(256), 0, s, partial, slabs, (long long)rows, (int)C, gamma, save_mean, save_inv_std, dgamma, dbeta, coef, grad_scale);
  if (dx) {
    const int agrid = grid_for(rows * (C / v), 256);
    if (dtype == VLFB_F32)
      hipLaunchKernelGGL((bn_apply_kernel<float, true>), dim3(agrid), dim3(256), 0, s, (const float*)x, (const float*)dy, (float*)dx, coef, (long long)rows, (int)C);
    else
      VLFB_WITH_T16(dtype, hipLaunchKernelGGL((bn_apply_kernel<T16, true>), dim3(agrid), dim3(256), 0, s, (const T16*)x, (const T16*)dy, (T16*)dx, coef, (long long)rows, (int)C));
  }
  return check_launch("bn_bwd");
}
