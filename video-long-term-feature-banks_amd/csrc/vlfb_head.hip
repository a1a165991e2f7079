// Head-side operators: LayerNorm, Dropout, FC, sigmoid cross-entropy, the one-query FBO-NL
// attention core, and the fused solver step.  Shapes here are tiny (R RoIs x 512 / 2560), so these
// are latency-bound; each is one launch with wave-level reductions.
#include "vlfb_common.h"
#include <math.h>

namespace vlfb {
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// block-wide sum for blockDim.x == 256 (4 waves); `red` is 4 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ---- LayerNorm (no affine), one wave per row ------------------------------------------------
template <typename T>
__global__ void layernorm_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                     float* __restrict__ rstd, long long rows, int cols, float eps) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long r = wave; r < rows; r += nwaves) {
    const T* xr = x + r * cols;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += Elem<T>::ld(xr + c);
    const float mean = wave_sum(s) / (float)cols;
    float v = 0.f;
    for (int c = lane; c < cols; c += 64) { float d = Elem<T>::ld(xr + c) - mean; v += d * d; }
    const float var = wave_sum(v) / (float)cols;
    const float rs = 1.0f / sqrtf(var + eps);
    for (int c = lane; c < cols; c += 64) Elem<T>::st(y + r * cols + c, (Elem<T>::ld(xr + c) - mean) * rs);
    if (lane == 0) rstd[r] = rs;
  }
}
// dx = rstd * (dy - mean(dy) - y * mean(dy*y))
template <typename T>
__global__ void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                     const float* __restrict__ rstd, T* __restrict__ dx,
                                     long long rows, int cols) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long r = wave; r < rows; r += nwaves) {
    const T* gr = dy + r * cols;
    const T* yr = y + r * cols;
    float a = 0.f, b = 0.f;
    for (int c = lane; c < cols; c += 64) {
      float g = Elem<T>::ld(gr + c);
      a += g;
      b += g * Elem<T>::ld(yr + c);
    }
    a = wave_sum(a) / (float)cols;
    b = wave_sum(b) / (float)cols;
    const float rs = rstd[r];
    for (int c = lane; c < cols; c += 64)
      Elem<T>::st(dx + r * cols + c, rs * (Elem<T>::ld(gr + c) - a - Elem<T>::ld(yr + c) * b));
  }
}

// ---- Dropout ------------------------------------------------------------------------------------
// u(seed, i): two rounds of the murmur3 finaliser over (i, seed) -> 24-bit uniform in [0,1).
// oracle/vlfb_oracle/rng.py implements the same function in numpy.
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ float dropout_uniform(uint64_t seed, uint64_t i) {
  uint32_t lo = (uint32_t)i, hi = (uint32_t)(i >> 32);
  uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
  uint32_t h = mix32(lo ^ s0);
  h = mix32(h + 0x9e3779b9u + (hi ^ s1));
  return (float)(h >> 8) * (1.0f / 16777216.0f);
}
template <typename T>
__global__ void dropout_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                   uint8_t* __restrict__ mask, long long rows, long long inner,
                                   long long ch, float ratio, unsigned long long seed,
                                   const unsigned long long* __restrict__ seed_dev) {
  if (seed_dev) seed = *seed_dev;     // captured step: the per-iteration seed lives in device memory
  const long long total = rows * inner * ch;
  const float keep_scale = 1.0f / (1.0f - ratio);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    // storage index i = (r*inner + k)*ch + c ; reference index = (r*ch + c)*inner + k
    const long long c = i % ch;
    const long long rk = i / ch;
    const long long k = rk % inner, r = rk / inner;
    const unsigned long long ref = (unsigned long long)((r * ch + c) * inner + k);
    const bool keep = dropout_uniform(seed, ref) >= ratio;
    mask[i] = keep ? 1 : 0;
    Elem<T>::st(y + i, keep ? Elem<T>::ld(x + i) * keep_scale : 0.f);
  }
}
template <typename T>
__global__ void dropout_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ mask,
                                   T* __restrict__ dx, long long n, float ratio) {
  const float keep_scale = 1.0f / (1.0f - ratio);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    Elem<T>::st(dx + i, mask[i] ? Elem<T>::ld(dy + i) * keep_scale : 0.f);
}

// ---- FC ---------------------------------------------------------------------------------------
// one wave per (row, class) pair
template <typename T>
__global__ void fc_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                              const float* __restrict__ b, float* __restrict__ logits,
                              long long rows, int cin, int cout) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  const long long total = rows * cout;
  for (long long i = wave; i < total; i += nwaves) {
    const long long r = i / cout;
    const int k = (int)(i - r * cout);
    const T* xr = x + r * cin;
    const float* wk = w + (long long)k * cin;
    float s = 0.f;
    for (int c = lane; c < cin; c += 64) s += Elem<T>::ld(xr + c) * wk[c];
    s = wave_sum(s);
    if (lane == 0) logits[i] = s + (b ? b[k] : 0.f);
  }
}
template <typename T>
__global__ void fc_bwd_dx_kernel(const float* __restrict__ w, const float* __restrict__ dl,
                                 T* __restrict__ dx, long long rows, int cin, int cout) {
  const long long total = rows * cin;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cin;
    const int c = (int)(i - r * cin);
    float s = 0.f;
    for (int k = 0; k < cout; ++k) s += dl[r * cout + k] * w[(long long)k * cin + c];
    Elem<T>::st(dx + i, s);
  }
}
template <typename T>
__global__ void fc_bwd_dw_kernel(const T* __restrict__ x, const float* __restrict__ dl,
                                 float* __restrict__ dw, float* __restrict__ db, long long rows,
                                 int cin, int cout, int accumulate) {
  const long long total = (long long)cout * cin;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i / cin);
    const int c = (int)(i - (long long)k * cin);
    float s = 0.f;
    for (long long r = 0; r < rows; ++r) s += dl[r * cout + k] * Elem<T>::ld(x + r * cin + c);
    dw[i] = accumulate ? dw[i] + s : s;
    if (c == 0 && db) {
      float t = 0.f;
      for (long long r = 0; r < rows; ++r) t += dl[r * cout + k];
      db[k] = accumulate ? db[k] + t : t;
    }
  }
}

// ---- Sigmoid + SigmoidCrossEntropyLoss (Detectron module semantics) ----------
__global__ void sigmoid_ce_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                  float* __restrict__ prob, float* __restrict__ loss,
                                  float* __restrict__ dlogits, long long n, float scale) {
  __shared__ float red[4];
  float cnt = 0.f, ls = 0.f;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) {
    const float x = logits[i];
    const int t = labels ? labels[i] : 0;
    if (prob) prob[i] = 1.0f / (1.0f + expf(-x));
    if (labels && t >= 0) {
      cnt += 1.f;
      const float pos = x >= 0.f ? 1.f : 0.f;
      ls += -x * ((float)t - pos) + logf(1.0f + expf(x - 2.0f * x * pos));
    }
  }
  if (!labels) return;
  cnt = block_sum(cnt, red);
  ls = block_sum(ls, red);
  const float normalizer = fmaxf(cnt, 1e-5f);
  if (threadIdx.x == 0 && loss) loss[0] = scale * ls / normalizer;
  if (dlogits) {
    for (long long i = threadIdx.x; i < n; i += blockDim.x) {
      const int t = labels[i];
      const float p = 1.0f / (1.0f + expf(-logits[i]));
      dlogits[i] = t >= 0 ? scale * (p - (float)t) / normalizer : 0.f;
    }
  }
}

// ---- Softmax / SoftmaxWithLoss (single-label heads: EPIC-Kitchens verb / noun, resnet_video.py:339-347) ----
// Caffe2 SoftmaxWithLoss with integer labels and no weights: P = softmax(logits) over the class axis,
// loss = scale * sum_r -log(max(P[r][label_r], 1e-20)) / rows, dlogits = scale * (P - onehot) / rows.
// One workgroup; a wave per row (rows = clips per GPU, cols = 125 / 352 classes).
__global__ void softmax_ce_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                  float* __restrict__ prob, float* __restrict__ loss,
                                  float* __restrict__ dlogits, int rows, int cols, float scale) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ls = 0.f;
  for (int r = wave; r < rows; r += 4) {
    const float* x = logits + (long long)r * cols;
    float m = -INFINITY;
    for (int c = lane; c < cols; c += 64) m = fmaxf(m, x[c]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float z = 0.f;
    for (int c = lane; c < cols; c += 64) z += expf(x[c] - m);
    for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o);
    const float inv = 1.0f / z;
    const int t = labels ? labels[r] : -1;
    // a label outside [0, cols) is an error upstream (Caffe2's SoftmaxWithLoss enforces the range): poison the loss so that
    // the NaN guard of the training loop (utils.misc.check_nan_losses) stops on it instead of training on a silent zero
    if (labels && (unsigned)t >= (unsigned)cols && lane == 0) ls = __builtin_nanf("");
    for (int c = lane; c < cols; c += 64) {
      const float p = expf(x[c] - m) * inv;
      if (prob) prob[(long long)r * cols + c] = p;
      if (dlogits) dlogits[(long long)r * cols + c] = scale * (p - (c == t ? 1.f : 0.f)) / (float)rows;
      if (c == t) ls += -logf(fmaxf(p, 1e-20f));
    }
  }
  if (!labels || !loss) return;
  for (int o = 32; o > 0; o >>= 1) ls += __shfl_xor(ls, o);
  ls = block_sum(lane == 0 ? ls : 0.f, red);
  if (threadIdx.x == 0) loss[0] = scale * ls / (float)rows;
}

// ---- FBO-NL attention core, one query per row (lfb_helper.py:170-263) ----------------------------
// one block (256 threads) per row r
template <typename T>
__global__ void fbo_attn_fwd_kernel(const T* __restrict__ theta, const T* __restrict__ phi,
                                    const T* __restrict__ g, float* __restrict__ p, T* __restrict__ t,
                                    int K, int D, long long ld, float scale, const float* __restrict__ owner = nullptr, int ostride = 0) {
  extern __shared__ float sm[];  // [K] logits/probs + 4 reduction slots
  float* s = sm;
  float* red = sm + K;
  const int r = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* th = theta + (long long)r * D;
  const long long kvr = owner ? (long long)owner[(long long)r * ostride] : (long long)r;      // the bank this row attends to
  const T* ph = phi + kvr * K * ld;
  const T* gg = g + kvr * K * ld;
  for (int k = wave; k < K; k += 4) {
    float a = 0.f;
    for (int d = lane; d < D; d += 64) a += Elem<T>::ld(th + d) * Elem<T>::ld(ph + (long long)k * ld + d);
    a = wave_sum(a);
    if (lane == 0) s[k] = a * scale;
  }
  __syncthreads();
  float m = -INFINITY;
  for (int k = threadIdx.x; k < K; k += blockDim.x) m = fmaxf(m, s[k]);
  m = block_max(m, red);
  float sum = 0.f;
  for (int k = threadIdx.x; k < K; k += blockDim.x) sum += expf(s[k] - m);
  sum = block_sum(sum, red);
  const float inv = 1.0f / sum;
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const float pk = expf(s[k] - m) * inv;
    s[k] = pk;
    p[(long long)r * K + k] = pk;
  }
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    // 8 independent partial sums keep 8 loads in flight (the serial chain was latency-bound)
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= K; k += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a8[u] += s[k + u] * Elem<T>::ld(gg + (long long)(k + u) * ld + d);
    }
    float a = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    for (; k < K; ++k) a += s[k] * Elem<T>::ld(gg + (long long)k * ld + d);
    Elem<T>::st(t + (long long)r * D + d, a);
  }
}
template <typename T>
__global__ void fbo_attn_bwd_kernel(const T* __restrict__ dt, const T* __restrict__ theta,
                                    const T* __restrict__ phi, const T* __restrict__ g,
                                    const float* __restrict__ p, T* __restrict__ dtheta,
                                    float* __restrict__ ds_out, int K, int D,
                                    long long ld, float scale) {
  extern __shared__ float sm[];  // [K] ds + 4
  float* ds = sm;
  float* red = sm + K;
  const int r = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* dtr = dt + (long long)r * D;
  const T* th = theta + (long long)r * D;
  const T* ph = phi + (long long)r * K * ld;
  const T* gg = g + (long long)r * K * ld;
  const float* pr = p + (long long)r * K;
  // dp[k] = <dt, g[k]>
  for (int k = wave; k < K; k += 4) {
    float a = 0.f;
    for (int d = lane; d < D; d += 64) a += Elem<T>::ld(dtr + d) * Elem<T>::ld(gg + (long long)k * ld + d);
    a = wave_sum(a);
    if (lane == 0) ds[k] = a;
  }
  __syncthreads();
  float dot = 0.f;
  for (int k = threadIdx.x; k < K; k += blockDim.x) dot += ds[k] * pr[k];
  dot = block_sum(dot, red);
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += blockDim.x) ds[k] = scale * pr[k] * (ds[k] - dot);
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= K; k += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a8[u] += ds[k + u] * Elem<T>::ld(ph + (long long)(k + u) * ld + d);
    }
    float a = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    for (; k < K; ++k) a += ds[k] * Elem<T>::ld(ph + (long long)k * ld + d);
    Elem<T>::st(dtheta + (long long)r * D + d, a);
  }
  // hand ds to the elementwise kernel (dphi / dg are written by the whole chip, not by R blocks)
  for (int k = threadIdx.x; k < K; k += blockDim.x) ds_out[(long long)r * K + k] = ds[k];
}
// dphi[r][k][:] = ds[r][k] * theta[r][:],  dg[r][k][:] = p[r][k] * dt[r][:]  (16 bytes per lane)
template <typename T>
__global__ void fbo_attn_bwd_kv_kernel(const T* __restrict__ dt, const T* __restrict__ theta,
                                       const float* __restrict__ p, const float* __restrict__ ds,
                                       T* __restrict__ dphi, T* __restrict__ dg, long long R, int K,
                                       int D, long long ld) {
  constexpr int V = Vec16<T>::N;
  const int dch = D / V;
  const long long total = R * K * dch;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int dc = (int)(i % dch);
    const long long rk = i / dch;
    const long long r = rk / K;
    float th[V], dtv[V], a[V], b[V];
    Vec16<T>::load(theta + r * D + dc * V, th);
    Vec16<T>::load(dt + r * D + dc * V, dtv);
    const float dsv = ds[rk], pv = p[rk];
#pragma unroll
    for (int e = 0; e < V; ++e) { a[e] = dsv * th[e]; b[e] = pv * dtv[e]; }
    Vec16<T>::store(dphi + rk * ld + dc * V, a);
    Vec16<T>::store(dg + rk * ld + dc * V, b);
  }
}

// ---- chip-wide variants (D a multiple of 8 lanes x 16 bytes): the per-row kernels above run R (= #RoIs,
// a few dozen) workgroups and are latency-bound (204 us for 24 RoIs x 300 bank features x 512) -------
// out[r][k] = scale * <q[r], kv[r][k]> : one wave per (r, k), 16 bytes per lane per pass
template <typename T>
__global__ void fbo_dot_kernel(const T* __restrict__ q, const T* __restrict__ kv, float* __restrict__ out,
                               long long RK, int K, int D, long long ld, float scale, const float* __restrict__ owner = nullptr, int ostride = 0) {
  constexpr int V = Vec16<T>::N;
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wave >= RK) return;
  const long long r = wave / K;
  const T* qr = q + r * D;
  const long long kvrow = owner ? (long long)owner[r * ostride] * K + (wave - r * K) : wave;
  const T* kr = kv + kvrow * ld;
  float a = 0.f;
  for (int d = lane * V; d < D; d += 64 * V) {
    float x[V], y[V];
    Vec16<T>::load(qr + d, x);
    Vec16<T>::load(kr + d, y);
#pragma unroll
    for (int e = 0; e < V; ++e) a += x[e] * y[e];
  }
  a = wave_sum(a);
  if (lane == 0) out[wave] = a * scale;
}
// weights of one row into LDS.  BWD = false: w = softmax(s).  BWD = true: s holds dp = <dt, g[k]>,
// w = scale * p * (dp - <p, dp>).
template <bool BWD>
__device__ __forceinline__ void fbo_row_weights(const float* __restrict__ sr, const float* __restrict__ pr,
                                                float* w, float* red, int K, float scale) {
  if (!BWD) {
    float m = -INFINITY;
    for (int k = threadIdx.x; k < K; k += blockDim.x) m = fmaxf(m, sr[k]);
    m = block_max(m, red);
    float sum = 0.f;
    for (int k = threadIdx.x; k < K; k += blockDim.x) sum += expf(sr[k] - m);
    sum = block_sum(sum, red);
    const float inv = 1.0f / sum;
    for (int k = threadIdx.x; k < K; k += blockDim.x) w[k] = expf(sr[k] - m) * inv;
  } else {
    float dot = 0.f;
    for (int k = threadIdx.x; k < K; k += blockDim.x) dot += sr[k] * pr[k];
    dot = block_sum(dot, red);
    for (int k = threadIdx.x; k < K; k += blockDim.x) w[k] = scale * pr[k] * (sr[k] - dot);
  }
  __syncthreads();
}
// out[r][d] = sum_k w[k] * kv[r][k][d] for this workgroup's 8 x V channels, w recomputed from s by
// every workgroup of the row (s is only READ here: the D/(8V) workgroups of a row run concurrently).
// grid = (R, D / (8 V)); 256 threads = 8 channel lanes x 32 key groups, folded through LDS.
template <typename T, bool BWD>
__global__ void fbo_mix_kernel(const float* __restrict__ s, const float* __restrict__ pin,
                               const T* __restrict__ kv, T* __restrict__ out,
                               int K, int D, long long ld, float scale, const float* __restrict__ owner = nullptr, int ostride = 0) {
  constexpr int V = Vec16<T>::N;
  extern __shared__ float sm[];          // [K] weights, [4] reduction slots, [32][8 V] partials
  float* w = sm;
  float* red = sm + K;
  float* part = red + 4;
  const int r = blockIdx.x, dblk = blockIdx.y;
  fbo_row_weights<BWD>(s + (long long)r * K, BWD ? pin + (long long)r * K : nullptr, w, red, K, scale);
  const int dl = threadIdx.x & 7, kg = threadIdx.x >> 3;
  const int d0 = (dblk * 8 + dl) * V;
  const T* base = kv + (owner ? (long long)owner[(long long)r * ostride] : (long long)r) * K * ld + d0;
  float a[V];
#pragma unroll
  for (int e = 0; e < V; ++e) a[e] = 0.f;
  for (int k = kg; k < K; k += 32) {
    float x[V];
    Vec16<T>::load(base + (long long)k * ld, x);
    const float wk = w[k];
#pragma unroll
    for (int e = 0; e < V; ++e) a[e] += wk * x[e];
  }
#pragma unroll
  for (int e = 0; e < V; ++e) part[(kg * 8 + dl) * V + e] = a[e];
  __syncthreads();
  if (threadIdx.x < 8 * V) {            // one thread per output channel of the block
    float t = 0.f;
    for (int j = 0; j < 32; ++j) t += part[j * 8 * V + threadIdx.x];
    Elem<T>::st(out + (long long)r * D + dblk * 8 * V + threadIdx.x, t);
  }
}

// afterwards, one workgroup per row turns s into the weights in place (p = softmax(logits) / ds from dp)
template <bool BWD>
__global__ void fbo_rowfix_kernel(float* __restrict__ s, const float* __restrict__ pin, int K, float scale) {
  extern __shared__ float sm[];          // [K] weights, [4] reduction slots
  const int r = blockIdx.x;
  float* sr = s + (long long)r * K;
  fbo_row_weights<BWD>(sr, BWD ? pin + (long long)r * K : nullptr, sm, sm + K, K, scale);
  for (int k = threadIdx.x; k < K; k += blockDim.x) sr[k] = sm[k];
}

// ---- solver -------------------------------------------------------------------------------------
__global__ void sgd_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                           long long n, float lr, float wd, float mu, int nesterov,
                           const float* __restrict__ lr_dev) {
  if (lr_dev) lr = *lr_dev;           // captured step: the learning rate of the iteration lives in device memory
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float pi = p[i];
    const float gi = g[i] + wd * pi;           // WeightedSum([g,1,p,wd]) model_builder_video.py:376-383
    const float mo = m[i];
    const float mn = mu * mo + lr * gi;        // MomentumSGDUpdate
    const float step = nesterov ? (1.f + mu) * mn - mu * mo : mn;
    g[i] = step;                               // Caffe2 writes the adjusted gradient back
    m[i] = mn;
    p[i] = pi - step;
  }
}
__global__ void scale_kernel(float* x, long long n, float s) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    x[i] *= s;
}

}  // namespace
}  // namespace vlfb

using namespace vlfb;

#define DISPATCH_T(dtype, NAME, ...)                                             \
  if ((dtype) == VLFB_F32) { NAME<float> __VA_ARGS__; }                          \
  else if ((dtype) == VLFB_BF16) { NAME<bf16_t> __VA_ARGS__; }                   \
  else if ((dtype) == VLFB_F16) { NAME<f16_t> __VA_ARGS__; }                     \
  else return set_error(VLFB_ERR_ARG, "bad dtype %d", (int)(dtype));

extern "C" int vlfb_layernorm_fwd(const void* x, void* y, float* rstd, int dtype, int64_t rows,
                                  int64_t cols, float eps, vlfb_stream_t stream) {
  VLFB_REQUIRE(x && y && rstd && rows > 0 && cols > 0, "layernorm_fwd: bad args");
  int grid = grid_for(rows * 64, 256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == VLFB_F32)
    hipLaunchKernelGGL(layernorm_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, rstd, (long long)rows, (int)cols, eps);
  else if (is16(dtype))
    VLFB_WITH_T16(dtype, hipLaunchKernelGGL(layernorm_fwd_kernel<T16>, dim3(grid), dim3(256), 0, s, (const T16*)x, (T16*)y, rstd, (long long)rows, (int)cols, eps));
  else return set_error(VLFB_ERR_ARG, "layernorm_fwd: bad dtype");
  return check_launch("layernorm_fwd");
}
extern "C" int vlfb_layernorm_bwd(const void* dy, const void* y, const float* rstd, void* dx,
                                  int dtype, int64_t rows, int64_t cols, vlfb_stream_t stream) {
  VLFB_REQUIRE(dy && y && rstd && dx && rows > 0 && cols > 0, "layernorm_bwd: bad args");
  int grid = grid_for(rows * 64, 256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == VLFB_F32)
    hipLaunchKernelGGL(layernorm_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dy, (const float*)y, rstd, (float*)dx, (long long)rows, (int)cols);
  else if (is16(dtype))
    VLFB_WITH_T16(dtype, hipLaunchKernelGGL(layernorm_bwd_kernel<T16>, dim3(grid), dim3(256), 0, s, (const T16*)dy, (const T16*)y, rstd, (T16*)dx, (long long)rows, (int)cols));
  else return set_error(VLFB_ERR_ARG, "layernorm_bwd: bad dtype");
  return check_launch("layernorm_bwd");
}
static int dropout_fwd_any(const void* x, void* y, uint8_t* mask, int dtype, int64_t rows, int64_t inner,
                           int64_t ch, float ratio, uint64_t seed, const uint64_t* seed_dev,
                           vlfb_stream_t stream) {
  VLFB_REQUIRE(x && y && mask && rows > 0 && inner > 0 && ch > 0, "dropout_fwd: bad args");
  VLFB_REQUIRE(ratio >= 0.f && ratio < 1.f, "dropout_fwd: ratio must be in [0,1)");
  int grid = grid_for(rows * inner * ch, 256);
  hipStream_t s = (hipStream_t)stream;
  const unsigned long long* sd = (const unsigned long long*)seed_dev;
  if (dtype == VLFB_F32)
    hipLaunchKernelGGL(dropout_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, mask, (long long)rows, (long long)inner, (long long)ch, ratio, (unsigned long long)seed, sd);
  else if (is16(dtype))
    VLFB_WITH_T16(dtype, hipLaunchKernelGGL(dropout_fwd_kernel<T16>, dim3(grid), dim3(256), 0, s, (const T16*)x, (T16*)y, mask, (long long)rows, (long long)inner, (long long)ch, ratio, (unsigned long long)seed, sd));
  else return set_error(VLFB_ERR_ARG, "dropout_fwd: bad dtype");
  return check_launch("dropout_fwd");
}
extern "C" int vlfb_dropout_fwd(const void* x, void* y, uint8_t* mask, int dtype, int64_t rows,
                                int64_t inner, int64_t ch, float ratio, uint64_t seed,
                                vlfb_stream_t stream) {
  return dropout_fwd_any(x, y, mask, dtype, rows, inner, ch, ratio, seed, nullptr, stream);
}
extern "C" int vlfb_dropout_fwd_dev(const void* x, void* y, uint8_t* mask, int dtype, int64_t rows,
                                    int64_t inner, int64_t ch, float ratio, const uint64_t* seed_dev,
                                    vlfb_stream_t stream) {
  VLFB_REQUIRE(seed_dev, "dropout_fwd_dev: null seed pointer");
  return dropout_fwd_any(x, y, mask, dtype, rows, inner, ch, ratio, 0, seed_dev, stream);
}
extern "C" int vlfb_dropout_bwd(const void* dy, const uint8_t* mask, void* dx, int dtype, int64_t n,
                                float ratio, vlfb_stream_t stream) {
  VLFB_REQUIRE(dy && mask && dx && n > 0, "dropout_bwd: bad args");
  int grid = grid_for(n, 256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == VLFB_F32)
    hipLaunchKernelGGL(dropout_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dy, mask, (float*)dx, (long long)n, ratio);
  else if (is16(dtype))
    VLFB_WITH_T16(dtype, hipLaunchKernelGGL(dropout_bwd_kernel<T16>, dim3(grid), dim3(256), 0, s, (const T16*)dy, mask, (T16*)dx, (long long)n, ratio));
  else return set_error(VLFB_ERR_ARG, "dropout_bwd: bad dtype");
  return check_launch("dropout_bwd");
}

extern "C" int vlfb_fc_fwd(const void* x, int dtype, const float* w, const float* b, float* logits,
                           int64_t rows, int64_t cin, int64_t cout, vlfb_stream_t stream) {
  VLFB_REQUIRE(x && w && logits && rows > 0 && cin > 0 && cout > 0, "fc_fwd: bad args");
  int grid = grid_for(rows * cout * 64, 256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == VLFB_F32)
    hipLaunchKernelGGL(fc_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, w, b, logits, (long long)rows, (int)cin, (int)cout);
  else if (is16(dtype))
    VLFB_WITH_T16(dtype, hipLaunchKernelGGL(fc_fwd_kernel<T16>, dim3(grid), dim3(256), 0, s, (const T16*)x, w, b, logits, (long long)rows, (int)cin, (int)cout));
  else return set_error(VLFB_ERR_ARG, "fc_fwd: bad dtype");
  return check_launch("fc_fwd");
}
extern "C" int vlfb_fc_bwd(const void* x, int dtype, const float* w, const float* dlogits, void* dx,
                           float* dw, float* db, int64_t rows, int64_t cin, int64_t cout,
                           int accumulate, vlfb_stream_t stream) {
  VLFB_REQUIRE(x && w && dlogits && rows > 0 && cin > 0 && cout > 0, "fc_bwd: bad args");
  hipStream_t s = (hipStream_t)stream;
  if (!dtype_ok(dtype)) return set_error(VLFB_ERR_ARG, "fc_bwd: bad dtype");
  if (dx) {
    int grid = grid_for(rows * cin, 256);
    if (dtype == VLFB_F32)
      hipLaunchKernelGGL(fc_bwd_dx_kernel<float>, dim3(grid), dim3(256), 0, s, w, dlogits, (float*)dx, (long long)rows, (int)cin, (int)cout);
    else
      VLFB_WITH_T16(dtype, hipLaunchKernelGGL(fc_bwd_dx_kernel<T16>, dim3(grid), dim3(256), 0, s, w, dlogits, (T16*)dx, (long long)rows, (int)cin, (int)cout));
  }
  if (dw) {
    int grid = grid_for(cout * cin, 256);
    if (dtype == VLFB_F32)
      hipLaunchKernelGGL(fc_bwd_dw_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, dlogits, dw, db, (long long)rows, (int)cin, (int)cout, accumulate);
    else
      VLFB_WITH_T16(dtype, hipLaunchKernelGGL(fc_bwd_dw_kernel<T16>, dim3(grid), dim3(256), 0, s, (const T16*)x, dlogits, dw, db, (long long)rows, (int)cin, (int)cout, accumulate));
  }
  return check_launch("fc_bwd");
}
extern "C" int vlfb_sigmoid_ce(const float* logits, const int32_t* labels, float* prob, float* loss,
                               float* dlogits, int64_t rows, int64_t cols, float scale,
                               vlfb_stream_t stream) {
  VLFB_REQUIRE(logits && rows > 0 && cols > 0, "sigmoid_ce: bad args");
  VLFB_REQUIRE(labels || (!loss && !dlogits), "sigmoid_ce: loss/dlogits need labels");
  hipLaunchKernelGGL(sigmoid_ce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, labels,
                     prob, loss, dlogits, (long long)(rows * cols), scale);
  return check_launch("sigmoid_ce");
}

extern "C" int vlfb_softmax_ce(const float* logits, const int32_t* labels, float* prob, float* loss,
                               float* dlogits, int64_t rows, int64_t cols, float scale, vlfb_stream_t stream) {
  VLFB_REQUIRE(logits && rows > 0 && cols > 0 && rows < (1 << 24) && cols < (1 << 24), "softmax_ce: bad args");
  VLFB_REQUIRE(labels || (!loss && !dlogits), "softmax_ce: loss/dlogits need labels");
  hipLaunchKernelGGL(softmax_ce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, labels, prob, loss,
                     dlogits, (int)rows, (int)cols, scale);
  return check_launch("softmax_ce");
}

static int fbo_attn_fwd_impl(const void* theta, const void* phi, const void* g, float* p, void* t,
                             int dtype, int64_t r, int64_t k, int64_t d, int64_t ld, float scale,
                             const float* owner, int ostride, vlfb_stream_t stream) {
  VLFB_REQUIRE(theta && phi && g && p && t && r > 0 && k > 0 && d > 0 && ld >= d, "fbo_attn_fwd: bad args");
  VLFB_REQUIRE(k <= 8192, "fbo_attn_fwd: bank too long for the LDS row buffer");
  size_t lds = (size_t)(k + 4) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  {
    const int v = dtype == VLFB_F32 ? 4 : 8;
    if ((dtype == VLFB_F32 || is16(dtype)) && d % (8 * v) == 0 && ld % v == 0) {
      // chip-wide path: logits into p (fp32 [R][K]), then softmax in place + weighted sum of g
      const long long rk = (long long)r * k;
      const unsigned g1 = (unsigned)((rk * 64 + 255) / 256);
      const dim3 g2((unsigned)r, (unsigned)(d / (8 * v)));
      const size_t lds2 = (size_t)(k + 4 + 32 * 8 * v) * sizeof(float);
      if (dtype == VLFB_F32) {
        hipLaunchKernelGGL(fbo_dot_kernel<float>, dim3(g1), dim3(256), 0, s, (const float*)theta, (const float*)phi, p, rk, (int)k, (int)d, (long long)ld, scale, owner, ostride);
        hipLaunchKernelGGL((fbo_mix_kernel<float, false>), g2, dim3(256), lds2, s, (const float*)p, (const float*)nullptr, (const float*)g, (float*)t, (int)k, (int)d, (long long)ld, scale, owner, ostride);
      } else {
        VLFB_WITH_T16(dtype, hipLaunchKernelGGL(fbo_dot_kernel<T16>, dim3(g1), dim3(256), 0, s, (const T16*)theta, (const T16*)phi, p, rk, (int)k, (int)d, (long long)ld, scale, owner, ostride));
        VLFB_WITH_T16(dtype, hipLaunchKernelGGL((fbo_mix_kernel<T16, false>), g2, dim3(256), lds2, s, (const float*)p, (const float*)nullptr, (const T16*)g, (T16*)t, (int)k, (int)d, (long long)ld, scale, owner, ostride));
      }
      hipLaunchKernelGGL((fbo_rowfix_kernel<false>), dim3((unsigned)r), dim3(256), lds, s, p, (const float*)nullptr, (int)k, scale);
      return check_launch("fbo_attn_fwd");
    }
  }
  if (dtype == VLFB_F32)
    hipLaunchKernelGGL(fbo_attn_fwd_kernel<float>, dim3((unsigned)r), dim3(256), lds, s, (const float*)theta, (const float*)phi, (const float*)g, p, (float*)t, (int)k, (int)d, (long long)ld, scale, owner, ostride);
  else if (is16(dtype))
    VLFB_WITH_T16(dtype, hipLaunchKernelGGL(fbo_attn_fwd_kernel<T16>, dim3((unsigned)r), dim3(256), lds, s, (const T16*)theta, (const T16*)phi, (const T16*)g, p, (T16*)t, (int)k, (int)d, (long long)ld, scale, owner, ostride));
  else return set_error(VLFB_ERR_ARG, "fbo_attn_fwd: bad dtype");
  return check_launch("fbo_attn_fwd");
}
extern "C" int vlfb_fbo_attn_fwd(const void* theta, const void* phi, const void* g, float* p, void* t,
                                 int dtype, int64_t r, int64_t k, int64_t d, int64_t ld, float scale,
                                 vlfb_stream_t stream) {
  return fbo_attn_fwd_impl(theta, phi, g, p, t, dtype, r, k, d, ld, scale, nullptr, 0, stream);
}
extern "C" int vlfb_fbo_attn_fwd_shared(const void* theta, const void* phi, const void* g, float* p, void* t,
                                        int dtype, int64_t r, int64_t k, int64_t d, int64_t ld, float scale,
                                        const float* owner, int64_t owner_stride, vlfb_stream_t stream) {
  VLFB_REQUIRE(owner && owner_stride > 0, "fbo_attn_fwd_shared: the owner column is required");
  return fbo_attn_fwd_impl(theta, phi, g, p, t, dtype, r, k, d, ld, scale, owner, (int)owner_stride, stream);
}
extern "C" int vlfb_fbo_attn_bwd(const void* dt, const void* theta, const void* phi, const void* g,
                                 const float* p, void* dtheta, void* dphi, void* dg, float* ds_ws,
                                 int dtype, int64_t r, int64_t k, int64_t d, int64_t ld, float scale,
                                 vlfb_stream_t stream) {
  VLFB_REQUIRE(dt && theta && phi && g && p && dtheta && dphi && dg && ds_ws && r > 0 && k > 0 && d > 0 && ld >= d,
               "fbo_attn_bwd: bad args");
  VLFB_REQUIRE(k <= 8192, "fbo_attn_bwd: bank too long for the LDS row buffer");
  const int v = dtype == VLFB_F32 ? 4 : 8;
  VLFB_REQUIRE(d % v == 0 && ld % v == 0, "fbo_attn_bwd: D and ld must be multiples of %d", v);
  size_t lds = (size_t)(k + 4) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  int grid2 = grid_for(r * k * (d / v), 256);
  if ((dtype == VLFB_F32 || is16(dtype)) && d % (8 * v) == 0) {
    // chip-wide path: dp = <dt, g[k]> into ds_ws, then ds (in place) + dtheta = sum_k ds[k] phi[k]
    const long long rk = (long long)r * k;
    const unsigned g1 = (unsigned)((rk * 64 + 255) / 256);
    const dim3 g2((unsigned)r, (unsigned)(d / (8 * v)));
    const size_t lds2 = (size_t)(k + 4 + 32 * 8 * v) * sizeof(float);
    if (dtype == VLFB_F32) {
      hipLaunchKernelGGL(fbo_dot_kernel<float>, dim3(g1), dim3(256), 0, s, (const float*)dt, (const float*)g, ds_ws, rk, (int)k, (int)d, (long long)ld, 1.0f);
      hipLaunchKernelGGL((fbo_mix_kernel<float, true>), g2, dim3(256), lds2, s, (const float*)ds_ws, p, (const float*)phi, (float*)dtheta, (int)k, (int)d, (long long)ld, scale);
      hipLaunchKernelGGL((fbo_rowfix_kernel<true>), dim3((unsigned)r), dim3(256), lds, s, ds_ws, p, (int)k, scale);
      hipLaunchKernelGGL(fbo_attn_bwd_kv_kernel<float>, dim3(grid2), dim3(256), 0, s, (const float*)dt, (const float*)theta, p, (const float*)ds_ws, (float*)dphi, (float*)dg, (long long)r, (int)k, (int)d, (long long)ld);
    } else {
      VLFB_WITH_T16(dtype, hipLaunchKernelGGL(fbo_dot_kernel<T16>, dim3(g1), dim3(256), 0, s, (const T16*)dt, (const T16*)g, ds_ws, rk, (int)k, (int)d, (long long)ld, 1.0f));
      VLFB_WITH_T16(dtype, hipLaunchKernelGGL((fbo_mix_kernel<T16, true>), g2, dim3(256), lds2, s, (const float*)ds_ws, p, (const T16*)phi, (T16*)dtheta, (int)k, (int)d, (long long)ld, scale));
      hipLaunchKernelGGL((fbo_rowfix_kernel<true>), dim3((unsigned)r), dim3(256), lds, s, ds_ws, p, (int)k, scale);
      VLFB_WITH_T16(dtype, hipLaunchKernelGGL(fbo_attn_bwd_kv_kernel<T16>, dim3(grid2), dim3(256), 0, s, (const T16*)dt, (const T16*)theta, p, (const float*)ds_ws, (T16*)dphi, (T16*)dg, (long long)r, (int)k, (int)d, (long long)ld));
    }
    return check_launch("fbo_attn_bwd");
  }
  if (dtype == VLFB_F32) {
    hipLaunchKernelGGL(fbo_attn_bwd_kernel<float>, dim3((unsigned)r), dim3(256), lds, s, (const float*)dt, (const float*)theta, (const float*)phi, (const float*)g, p, (float*)dtheta, ds_ws, (int)k, (int)d, (long long)ld, scale);
    hipLaunchKernelGGL(fbo_attn_bwd_kv_kernel<float>, dim3(grid2), dim3(256), 0, s, (const float*)dt, (const float*)theta, p, (const float*)ds_ws, (float*)dphi, (float*)dg, (long long)r, (int)k, (int)d, (long long)ld);
  } else if (is16(dtype)) {
    VLFB_WITH_T16(dtype, hipLaunchKernelGGL(fbo_attn_bwd_kernel<T16>, dim3((unsigned)r), dim3(256), lds, s, (const T16*)dt, (const T16*)theta, (const T16*)phi, (const T16*)g, p, (T16*)dtheta, ds_ws, (int)k, (int)d, (long long)ld, scale));
    VLFB_WITH_T16(dtype, hipLaunchKernelGGL(fbo_attn_bwd_kv_kernel<T16>, dim3(grid2), dim3(256), 0, s, (const T16*)dt, (const T16*)theta, p, (const float*)ds_ws, (T16*)dphi, (T16*)dg, (long long)r, (int)k, (int)d, (long long)ld));
  } else return set_error(VLFB_ERR_ARG, "fbo_attn_bwd: bad dtype");
  return check_launch("fbo_attn_bwd");
}

extern "C" int vlfb_sgd_update(float* p, float* g, float* m, int64_t n, float lr, float wd, float mu,
                               int nesterov, vlfb_stream_t stream) {
  VLFB_REQUIRE(p && g && m && n > 0, "sgd_update: bad args");
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m,
                     (long long)n, lr, wd, mu, nesterov, (const float*)nullptr);
  return check_launch("sgd_update");
}
extern "C" int vlfb_sgd_update_dev(float* p, float* g, float* m, int64_t n, const float* lr_dev, float wd,
                                   float mu, int nesterov, vlfb_stream_t stream) {
  VLFB_REQUIRE(p && g && m && n > 0 && lr_dev, "sgd_update_dev: bad args");
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m,
                     (long long)n, 0.f, wd, mu, nesterov, lr_dev);
  return check_launch("sgd_update_dev");
}
// per-iteration scalars of a captured step (learning rate, dropout seeds): the values travel in the launch
// packet, so the host may overwrite its copy at once, and land in device memory in stream order
struct StepScalars { unsigned long long v[8]; };
namespace vlfb { namespace {
__global__ void store_scalars_kernel(unsigned long long* dst, int n, StepScalars s) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = s.v[threadIdx.x];
}
} }
extern "C" int vlfb_store_scalars(uint64_t* dst, int n, const uint64_t* values, vlfb_stream_t stream) {
  VLFB_REQUIRE(dst && values && n > 0 && n <= 8, "store_scalars: 1..8 values");
  StepScalars s;
  for (int i = 0; i < 8; ++i) s.v[i] = i < n ? values[i] : 0ull;
  hipLaunchKernelGGL(store_scalars_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                     (unsigned long long*)dst, n, s);
  return check_launch("store_scalars");
}
extern "C" int vlfb_scale_inplace(float* x, int64_t n, float sc, vlfb_stream_t stream) {
  VLFB_REQUIRE(x && n > 0, "scale_inplace: bad args");
  hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (long long)n, sc);
  return check_launch("scale_inplace");
}
