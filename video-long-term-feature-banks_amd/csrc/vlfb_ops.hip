// HBM-bound operators of the hot path: AffineNd (the reference's only native op), layout movers,
// pools, row softmax, residual add / ReLU, column sums.  All are one-pass, 16-byte-per-lane
// vectorised kernels; the roofline that bounds them is HBM bandwidth.
#include "vlfb_common.h"
#include <type_traits>
#include <math.h>

namespace vlfb {

static thread_local char g_err[512] = "";
int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

namespace {

// ---------------------------------------------------------------------------------------------
// AffineNd: y[n][c][i] = x*s[c] + b[c]   (affine_nd_op.cu:31-44), dx = dy*s[c] (46-58)
// grid.y walks the (n,c) rows so the per-row scale/bias are scalar loads.
// ---------------------------------------------------------------------------------------------
template <bool HAS_BIAS>
__global__ void affine_nd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                 const float* __restrict__ bias, float* __restrict__ y,
                                 long long rows, int C, long long inner) {
  for (long long row = blockIdx.y; row < rows; row += gridDim.y) {
    const int c = (int)(row % C);
    const float s = scale[c];
    const float b = HAS_BIAS ? bias[c] : 0.f;
    const float* xr = x + row * inner;
    float* yr = y + row * inner;
    const bool vec = ((inner & 3) == 0) && ((((uintptr_t)xr | (uintptr_t)yr) & 15) == 0);
    if (vec) {
      const long long n4 = inner >> 2;
      for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
           i += (long long)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(xr)[i];
        v.x = v.x * s + b; v.y = v.y * s + b; v.z = v.z * s + b; v.w = v.w * s + b;
        reinterpret_cast<float4*>(yr)[i] = v;
      }
    } else {
      for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < inner;
           i += (long long)gridDim.x * blockDim.x)
        yr[i] = xr[i] * s + b;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// layout movers
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void ncthw_to_nthwc_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                      long long n, int c, long long thw, int c_pad) {
  const long long total = n * thw;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / thw, pos = i - b * thw;
    const float* s = src + b * c * thw + pos;
    T* d = dst + i * c_pad;
    for (int k = 0; k < c_pad; ++k) Elem<T>::st(d + k, k < c ? s[(long long)k * thw] : 0.f);
  }
}
// the same with zero pixels on both sides of every W row (the stem kernels then never need a
// w-bounds test: [N][rows][wl + W + wr][c_pad])
template <typename T>
__global__ void ncthw_to_nthwc_wpad_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                           long long n, int c, long long rows, int w, int c_pad,
                                           int wl, int wtot) {
  const long long total = n * rows * wtot;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int wp = (int)(i % wtot);
    const long long nr = i / wtot;
    const long long b = nr / rows, r = nr - b * rows;
    const int x = wp - wl;
    const bool in = x >= 0 && x < w;
    const float* sp = src + (b * c * rows + r) * w + x;
    T* d = dst + i * c_pad;
    for (int k = 0; k < c_pad; ++k) Elem<T>::st(d + k, (in && k < c) ? sp[(long long)k * rows * w] : 0.f);
  }
}
template <typename T>
__global__ void nthwc_to_ncthw_kernel(const T* __restrict__ src, float* __restrict__ dst,
                                      long long n, int c, long long thw) {
  const long long total = n * c * thw;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long pos = i % thw;
    const long long bc = i / thw;
    const long long b = bc / c;
    const int k = (int)(bc - b * c);
    dst[i] = Elem<T>::ld(src + (b * thw + pos) * c + k);
  }
}
template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    Elem<D>::st(dst + i, Elem<S>::ld(src + i));
}
template <typename T>
__global__ void transpose2d_kernel(const T* __restrict__ src, T* __restrict__ dst, long long rows,
                                   long long cols) {
  __shared__ T tile[32][33];
  const long long b = blockIdx.z;
  const T* s = src + b * rows * cols;
  T* d = dst + b * rows * cols;
  const long long c0 = (long long)blockIdx.x * 32, r0 = (long long)blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    long long r = r0 + j, c = c0 + threadIdx.x;
    if (r < rows && c < cols) tile[j][threadIdx.x] = s[r * cols + c];
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    long long c = c0 + j, r = r0 + threadIdx.x;
    if (r < rows && c < cols) d[c * rows + r] = tile[threadIdx.x][j];
  }
}

// bf16 term planes of the split-bf16 path: value v -> h = bf16(v), m = bf16(v - h), l = bf16(v - h - m)
// rounded_mul: the fp32-ROUNDED product (hipcc contracts `w * s - h` into an fma otherwise and the remainder terms
// would expand the unrounded product; the planes are defined as the expansion of the fp32 operand value)
__device__ __forceinline__ float rounded_mul(float a, float b) {
  float v = a * b;
  asm volatile("" : "+v"(v));
  return v;
}
struct SplitW {};       // tag type: weight_prep kernels instantiated with it write planes instead of elements
template <int NPL>
__device__ __forceinline__ void store_terms(bf16_t* dst, long long idx, long long plane, float v) {
#pragma unroll
  for (int pl = 0; pl < NPL; ++pl) {
    const bf16_t t = f2bf(v);
    dst[pl * plane + idx] = t;
    v -= bf2f(t);
  }
}
template <typename T> struct WStore {
  __device__ static __forceinline__ void fprop(void* base, long long idx, long long, float v) { Elem<T>::st(reinterpret_cast<T*>(base) + idx, v); }
  __device__ static __forceinline__ void dgrad(void* base, long long idx, long long, float v) { Elem<T>::st(reinterpret_cast<T*>(base) + idx, v); }
};
template <> struct WStore<SplitW> {
  __device__ static __forceinline__ void fprop(void* base, long long idx, long long plane, float v) { store_terms<3>(reinterpret_cast<bf16_t*>(base), idx, plane, v); }
  __device__ static __forceinline__ void dgrad(void* base, long long idx, long long plane, float v) { store_terms<2>(reinterpret_cast<bf16_t*>(base), idx, plane, v); }
};

// "mix" engine (VLFB_MIX): split-bf16 forward, fp16 backward -- the FPROP copy as the three bf16 term planes of VLFB_SPLIT,
// the DGRAD copy as plain fp16 elements [Cin][taps][Cout]
struct MixW {};
template <> struct WStore<MixW> {
  __device__ static __forceinline__ void fprop(void* base, long long idx, long long plane, float v) { store_terms<3>(reinterpret_cast<bf16_t*>(base), idx, plane, v); }
  __device__ static __forceinline__ void dgrad(void* base, long long idx, long long, float v) { reinterpret_cast<unsigned short*>(base)[idx] = f2h(v); }
};

// VLFB_MIX_W2: the DGRAD copy as TWO fp16 terms of (w * s) * VLFB_MIX_W2_SCALE, [Cin][term][taps][Cout] -- the weight of
// a convolution with a doubled outermost tap dimension of dilation 0 (every tap is visited once per term), which the
// plain fp16 DGRAD kernels contract with the fp16 gradient: dX = dY . (Wh + Wl), 22 significant bits of W
struct MixW2 {};
template <> struct WStore<MixW2> : WStore<MixW> {};

// VLFB_MIXH / VLFB_MIXH_W2: the two-plane fp16 forward (VLFB_MATH_F16X3) -- the FPROP copy as the two fp16 terms of
// (w * s) * VLFB_MIX_W2_SCALE, planes [term][Cout][taps][Cin] (the scale keeps the low term of a small weight in the fp16
// normal range; the launch's alpha carries its inverse); the DGRAD copy as VLFB_MIX / VLFB_MIX_W2
struct MixHW {};
struct MixHW2 {};
__device__ __forceinline__ void store_h2(void* base, long long idx, long long plane, float v) {
  unsigned short* o = reinterpret_cast<unsigned short*>(base);
  const float sv = v * VLFB_MIX_W2_SCALE;
  const unsigned short h = f2h(sv);
  o[idx] = h;
  o[plane + idx] = f2h(sv - h2f(h));
}
template <> struct WStore<MixHW> {
  __device__ static __forceinline__ void fprop(void* base, long long idx, long long plane, float v) { store_h2(base, idx, plane, v); }
  __device__ static __forceinline__ void dgrad(void* base, long long idx, long long, float v) { reinterpret_cast<unsigned short*>(base)[idx] = f2h(v); }
};
template <> struct WStore<MixHW2> : WStore<MixHW> {};
// VLFB_MIX_W2I / VLFB_MIXH_W2I: the two-term DGRAD copy interleaved per 64-channel k-tile, [Cin][taps][Cout / 64][term][64]
// (the weight operand of VLFB_MATH_F16W2: gemm_nt_kernel<.., W2I>)
struct MixW2I {};
struct MixHW2I {};
template <> struct WStore<MixW2I> : WStore<MixW> {};
template <> struct WStore<MixHW2I> : WStore<MixHW> {};

template <typename T>
__device__ __forceinline__ void wstore_dgrad(void* base, int ci, int tap, int co, int taps, int cout, long long plane, float v) {
  if constexpr (std::is_same<T, MixW2>::value || std::is_same<T, MixHW2>::value) {
    unsigned short* o = reinterpret_cast<unsigned short*>(base) + ((long long)ci * 2 * taps + tap) * cout + co;
    const float sv = v * VLFB_MIX_W2_SCALE;
    const unsigned short h = f2h(sv);
    o[0] = h;
    o[(long long)taps * cout] = f2h(sv - h2f(h));
  } else if constexpr (std::is_same<T, MixW2I>::value || std::is_same<T, MixHW2I>::value) {
    unsigned short* o = reinterpret_cast<unsigned short*>(base) + (((long long)ci * taps + tap) * cout + (co & ~63)) * 2 + (co & 63);
    const float sv = v * VLFB_MIX_W2_SCALE;
    const unsigned short h = f2h(sv);
    o[0] = h;
    o[64] = f2h(sv - h2f(h));
  } else {
    WStore<T>::dgrad(base, ((long long)ci * taps + tap) * cout + co, plane, v);
  }
}

// w[Cout][taps][Cin] fp32 (+ scale[Cout]) -> fprop copy (same order) in T
template <typename T>
__global__ void weight_prep_fprop_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                         void* __restrict__ out, long long per_cout, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const float s = scale ? scale[i / per_cout] : 1.f;
    WStore<T>::fprop(out, i, total, rounded_mul(w[i], s));
  }
}
// -> dgrad copy [Cin][taps][Cout]; blockIdx.z = tap; 32x32 LDS tile transpose
template <typename T>
__global__ void weight_prep_dgrad_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                         void* __restrict__ out, int cout, int taps, int cin) {
  __shared__ float tile[32][33];
  const int tap = blockIdx.z;
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    int co = co0 + j, ci = ci0 + threadIdx.x;
    if (co < cout && ci < cin)
      tile[j][threadIdx.x] = rounded_mul(w[((long long)co * taps + tap) * cin + ci], scale ? scale[co] : 1.f);
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    int ci = ci0 + j, co = co0 + threadIdx.x;
    if (co < cout && ci < cin)
      wstore_dgrad<T>(out, ci, tap, co, taps, cout, (long long)cout * taps * cin, tile[threadIdx.x][j]);
  }
}

// All convolutions of the model in ONE launch: blockIdx.x walks 32x32 (cout x cin) tiles of every
// (layer, tap); the tile is read once (coalesced along cin) and written twice: fprop order
// [cout][tap][cin] and dgrad order [cin][tap][cout] (transposed through LDS).
template <typename T>
__global__ void weight_prep_batched_kernel(const vlfb_wprep_item* __restrict__ items, int n_items) {
  __shared__ float tile[32][33];
  const int b = blockIdx.x;
  int lo = 0, hi = n_items - 1;          // last item whose first tile is <= b
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].tile_begin <= b) lo = mid; else hi = mid - 1;
  }
  const vlfb_wprep_item it = items[lo];
  const int t = b - it.tile_begin;
  const int tiles_ci = (it.cin + 31) >> 5, tiles_co = (it.cout + 31) >> 5;
  const int ci0 = (t % tiles_ci) << 5;
  const int co0 = ((t / tiles_ci) % tiles_co) << 5;
  const int tap = t / (tiles_ci * tiles_co);
  const float* w = reinterpret_cast<const float*>(it.w);
  const float* scale = reinterpret_cast<const float*>(it.scale);
  void* wf = it.w_fprop;
  void* wd = it.w_dgrad;
  const long long plane = (long long)it.cout * it.taps * it.cin;
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int co = co0 + j, ci = ci0 + threadIdx.x;
    if (co < it.cout && ci < it.cin) {
      const long long idx = ((long long)co * it.taps + tap) * it.cin + ci;
      const float v = rounded_mul(w[idx], scale ? scale[co] : 1.f);     // (no fma contraction into the term remainders)
      tile[j][threadIdx.x] = v;
      if (wf) WStore<T>::fprop(wf, idx, plane, v);
    }
  }
  if (!wd) return;
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int ci = ci0 + j, co = co0 + threadIdx.x;
    if (co < it.cout && ci < it.cin)
      wstore_dgrad<T>(wd, ci, tap, co, it.taps, it.cout, plane, tile[threadIdx.x][j]);
  }
}

// ---------------------------------------------------------------------------------------------
// pools (channels-last).  One thread = one position x one 16-byte channel chunk.
// ---------------------------------------------------------------------------------------------
struct PoolP {
  int N, Ti, Hi, Wi, C, To, Ho, Wo;
  int kt, kh, kw, st, sh, sw, pt, ph, pw;
};

template <typename T, typename IdxT>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                   IdxT* __restrict__ argmax, PoolP p) {
  constexpr int V = Vec16<T>::N;
  const int cchunks = p.C / V;
  const long long total = (long long)p.N * p.To * p.Ho * p.Wo * cchunks;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cchunks);
    long long o = i / cchunks;
    const int wo = (int)(o % p.Wo); long long r = o / p.Wo;
    const int ho = (int)(r % p.Ho); r /= p.Ho;
    const int to = (int)(r % p.To);
    const int n = (int)(r / p.To);
    float best[V];
    int arg[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { best[k] = -INFINITY; arg[k] = 0; }
    for (int a = 0; a < p.kt; ++a) {
      const int ti = to * p.st - p.pt + a;
      if ((unsigned)ti >= (unsigned)p.Ti) continue;
      for (int b = 0; b < p.kh; ++b) {
        const int hi = ho * p.sh - p.ph + b;
        if ((unsigned)hi >= (unsigned)p.Hi) continue;
        for (int c = 0; c < p.kw; ++c) {
          const int wi = wo * p.sw - p.pw + c;
          if ((unsigned)wi >= (unsigned)p.Wi) continue;
          float v[V];
          Vec16<T>::load(x + ((((long long)n * p.Ti + ti) * p.Hi + hi) * p.Wi + wi) * p.C + cc * V, v);
          const int tap = (a * p.kh + b) * p.kw + c;
#pragma unroll
          for (int k = 0; k < V; ++k)
            if (v[k] > best[k]) { best[k] = v[k]; arg[k] = tap; }
        }
      }
    }
    Vec16<T>::store(y + o * p.C + cc * V, best);
    if (argmax && sizeof(IdxT) == 2) {
      IdxT* am = argmax + o * p.C + cc * V;
#pragma unroll
      for (int k = 0; k < V; ++k) am[k] = (IdxT)arg[k];
    } else if (argmax) {
      uint8_t* am = reinterpret_cast<uint8_t*>(argmax) + o * p.C + cc * V;
      if (V == 8) {
        uint2 pk;
        pk.x = arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24);
        pk.y = arg[4 % V] | (arg[5 % V] << 8) | (arg[6 % V] << 16) | (arg[7 % V] << 24);
        *reinterpret_cast<uint2*>(am) = pk;
      } else {
        *reinterpret_cast<uint32_t*>(am) = arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24);
      }
    }
  }
}

// Two-plane fp16 tensors (VLFB_F16PAIR: [2][numel] fp16, value = hi + lo; the forward activations of the "mix" path):
// the window maximum of hi + lo, stored as the selected element's two planes (a copy: the pooled value IS an input value)
struct PairVec {
  __device__ static __forceinline__ void load(const f16_t* hi, long long plane, float (&h)[8], float (&l)[8]) {
    Vec16<f16_t>::load(hi, h);
    Vec16<f16_t>::load(hi + plane, l);
  }
};
template <typename IdxT>
__global__ void maxpool_fwd_pair_kernel(const f16_t* __restrict__ x, f16_t* __restrict__ y, IdxT* __restrict__ argmax, PoolP p) {
  constexpr int V = 8;
  const int cchunks = p.C / V;
  const long long total = (long long)p.N * p.To * p.Ho * p.Wo * cchunks;
  const long long xplane = (long long)p.N * p.Ti * p.Hi * p.Wi * p.C, yplane = (long long)p.N * p.To * p.Ho * p.Wo * p.C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cchunks);
    long long o = i / cchunks;
    const int wo = (int)(o % p.Wo); long long r = o / p.Wo;
    const int ho = (int)(r % p.Ho); r /= p.Ho;
    const int to = (int)(r % p.To);
    const int n = (int)(r / p.To);
    float best[V], bh[V], bl[V];
    int arg[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { best[k] = -INFINITY; bh[k] = bl[k] = 0.f; arg[k] = 0; }
    for (int a = 0; a < p.kt; ++a) {
      const int ti = to * p.st - p.pt + a;
      if ((unsigned)ti >= (unsigned)p.Ti) continue;
      for (int b = 0; b < p.kh; ++b) {
        const int hi = ho * p.sh - p.ph + b;
        if ((unsigned)hi >= (unsigned)p.Hi) continue;
        for (int c = 0; c < p.kw; ++c) {
          const int wi = wo * p.sw - p.pw + c;
          if ((unsigned)wi >= (unsigned)p.Wi) continue;
          float h[V], l[V];
          PairVec::load(x + ((((long long)n * p.Ti + ti) * p.Hi + hi) * p.Wi + wi) * p.C + cc * V, xplane, h, l);
          const int tap = (a * p.kh + b) * p.kw + c;
#pragma unroll
          for (int k = 0; k < V; ++k) {
            const float v = h[k] + l[k];
            if (v > best[k]) { best[k] = v; bh[k] = h[k]; bl[k] = l[k]; arg[k] = tap; }
          }
        }
      }
    }
    Vec16<f16_t>::store(y + o * p.C + cc * V, bh);
    Vec16<f16_t>::store(y + yplane + o * p.C + cc * V, bl);
    if (argmax && sizeof(IdxT) == 2) {
      IdxT* am = argmax + o * p.C + cc * V;
#pragma unroll
      for (int k = 0; k < V; ++k) am[k] = (IdxT)arg[k];
    } else if (argmax) {
      uint2 pk;
      pk.x = arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24);
      pk.y = arg[4] | (arg[5] << 8) | (arg[6] << 16) | (arg[7] << 24);
      *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(argmax) + o * p.C + cc * V) = pk;
    }
  }
}
// average pools over a two-plane fp16 tensor, fp32 output (the head reads res5 through them: head_helper.py:37-40, 92-98)
__global__ void avgpool_fwd_pair_kernel(const f16_t* __restrict__ x, float* __restrict__ y, PoolP p) {
  constexpr int V = 8;
  const int cchunks = p.C / V;
  const long long total = (long long)p.N * p.To * p.Ho * p.Wo * cchunks;
  const long long xplane = (long long)p.N * p.Ti * p.Hi * p.Wi * p.C;
  const float inv = 1.0f / (float)(p.kt * p.kh * p.kw);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cchunks);
    long long o = i / cchunks;
    const int wo = (int)(o % p.Wo); long long r = o / p.Wo;
    const int ho = (int)(r % p.Ho); r /= p.Ho;
    const int to = (int)(r % p.To);
    const int n = (int)(r / p.To);
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    for (int a = 0; a < p.kt; ++a)
      for (int b = 0; b < p.kh; ++b)
        for (int c = 0; c < p.kw; ++c) {
          const int ti = to * p.st + a, hi = ho * p.sh + b, wi = wo * p.sw + c;
          float h[V], l[V];
          PairVec::load(x + ((((long long)n * p.Ti + ti) * p.Hi + hi) * p.Wi + wi) * p.C + cc * V, xplane, h, l);
#pragma unroll
          for (int k = 0; k < V; ++k) acc[k] += h[k] + l[k];
        }
    float* yo = y + o * p.C + cc * V;
    *reinterpret_cast<float4*>(yo) = make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
    *reinterpret_cast<float4*>(yo + 4) = make_float4(acc[4] * inv, acc[5] * inv, acc[6] * inv, acc[7] * inv);
  }
}
__global__ void global_avgpool_pair_kernel(const f16_t* __restrict__ x, float* __restrict__ y, long long rows, int C, long long xplane) {
  constexpr int V = 8;
  __shared__ float red[32][8 * 8 + 1];
  const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int n = blockIdx.y;
  const int c0 = (blockIdx.x * 8 + cl) * V;
  float acc[V];
#pragma unroll
  for (int k = 0; k < V; ++k) acc[k] = 0.f;
  if (c0 < C) {
    const f16_t* base = x + (long long)n * rows * C + c0;
    for (long long r = rl; r < rows; r += 32) {
      float h[V], l[V];
      PairVec::load(base + r * C, xplane, h, l);
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] += h[k] + l[k];
    }
  }
#pragma unroll
  for (int k = 0; k < V; ++k) red[rl][cl * 8 + k] = acc[k];
  __syncthreads();
  if (rl == 0 && c0 < C) {
    const float inv = 1.0f / (float)rows;
    float* yo = y + (long long)n * C + c0;
#pragma unroll
    for (int k = 0; k < V; ++k) {
      float sum = 0.f;
      for (int j = 0; j < 32; ++j) sum += red[j][cl * 8 + k];
      yo[k] = sum * inv;
    }
  }
}
// fp32 -> the two fp16 planes, and back
// (pure streaming passes, often next to a GEMM on the other stream: non-temporal accesses keep them out of the L2 that the
// GEMM's operand panels live in -- stream_ld / stream_st below; VLFB_NT_EPI=0 switches the hint off, as in the GEMM epilogues)
typedef unsigned int u32x4_s __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 stream_ld16(bool nt, const void* p) {
  if (nt) { const u32x4_s t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_s*>(p)); return make_uint4(t.x, t.y, t.z, t.w); }
  return *reinterpret_cast<const uint4*>(p);
}
__device__ __forceinline__ uint2 stream_ld8(bool nt, const void* p) {
  if (nt) { const u32x2_s t = __builtin_nontemporal_load(reinterpret_cast<const u32x2_s*>(p)); return make_uint2(t.x, t.y); }
  return *reinterpret_cast<const uint2*>(p);
}
__device__ __forceinline__ void stream_st16(bool nt, void* p, uint4 v) {
  if (nt) { const u32x4_s t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<u32x4_s*>(p)); }
  else *reinterpret_cast<uint4*>(p) = v;
}
__device__ __forceinline__ void stream_st8(bool nt, void* p, uint2 v) {
  if (nt) { const u32x2_s t = {v.x, v.y}; __builtin_nontemporal_store(t, reinterpret_cast<u32x2_s*>(p)); }
  else *reinterpret_cast<uint2*>(p) = v;
}
static bool stream_nt() {
  static const bool on = !(getenv("VLFB_NT_EPI") && atoi(getenv("VLFB_NT_EPI")) == 0);
  return on;
}
__global__ void pair_split_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, long long n, bool nt) {
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * blockDim.x * 4) {
    const uint4 u = stream_ld16(nt, src + i);
    const float4 v = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
    const unsigned short h0 = f2h(v.x), h1 = f2h(v.y), h2 = f2h(v.z), h3 = f2h(v.w);
    stream_st8(nt, dst + i, make_uint2((uint32_t)h0 | ((uint32_t)h1 << 16), (uint32_t)h2 | ((uint32_t)h3 << 16)));
    stream_st8(nt, dst + n + i, make_uint2((uint32_t)f2h(v.x - h2f(h0)) | ((uint32_t)f2h(v.y - h2f(h1)) << 16),
                                           (uint32_t)f2h(v.z - h2f(h2)) | ((uint32_t)f2h(v.w - h2f(h3)) << 16)));
  }
}
__global__ void pair_join_kernel(const unsigned short* __restrict__ src, float* __restrict__ dst, long long n, bool nt) {
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * blockDim.x * 4) {
    const uint2 h = stream_ld8(nt, src + i), l = stream_ld8(nt, src + n + i);
    const float4 o = make_float4(h2f((unsigned short)(h.x & 0xffffu)) + h2f((unsigned short)(l.x & 0xffffu)), h2f((unsigned short)(h.x >> 16)) + h2f((unsigned short)(l.x >> 16)),
                                                      h2f((unsigned short)(h.y & 0xffffu)) + h2f((unsigned short)(l.y & 0xffffu)), h2f((unsigned short)(h.y >> 16)) + h2f((unsigned short)(l.y >> 16)));
    stream_st16(nt, dst + i, make_uint4(__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)));
  }
}

// gather formulation of the pool backward: every input position sums the output gradients
// of the windows that (a) cover it and (b) for max pooling selected it.
template <typename T, bool IS_MAX, typename IdxT>
__global__ void pool_bwd_kernel(const T* __restrict__ dy, const IdxT* __restrict__ argmax,
                                T* dx, const T* add, const T* __restrict__ mask, PoolP p,
                                float inv_window, const T* __restrict__ ymask = nullptr) {
  constexpr int V = Vec16<T>::N;
  const int cchunks = p.C / V;
  const long long total = (long long)p.N * p.Ti * p.Hi * p.Wi * cchunks;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cchunks);
    long long q = i / cchunks;
    const int wi = (int)(q % p.Wi); long long r = q / p.Wi;
    const int hi = (int)(r % p.Hi); r /= p.Hi;
    const int ti = (int)(r % p.Ti);
    const int n = (int)(r / p.Ti);
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    // windows `to` with to*st - pt <= ti <= to*st - pt + kt - 1
    const int t_lo = max(0, (ti + p.pt - p.kt + p.st) / p.st), t_hi = min(p.To - 1, (ti + p.pt) / p.st);
    const int h_lo = max(0, (hi + p.ph - p.kh + p.sh) / p.sh), h_hi = min(p.Ho - 1, (hi + p.ph) / p.sh);
    const int w_lo = max(0, (wi + p.pw - p.kw + p.sw) / p.sw), w_hi = min(p.Wo - 1, (wi + p.pw) / p.sw);
    for (int to = t_lo; to <= t_hi; ++to) {
      const int a = ti + p.pt - to * p.st;
      if (a < 0 || a >= p.kt) continue;
      for (int ho = h_lo; ho <= h_hi; ++ho) {
        const int b = hi + p.ph - ho * p.sh;
        if (b < 0 || b >= p.kh) continue;
        for (int wo = w_lo; wo <= w_hi; ++wo) {
          const int c = wi + p.pw - wo * p.sw;
          if (c < 0 || c >= p.kw) continue;
          const long long o = (((long long)n * p.To + to) * p.Ho + ho) * p.Wo + wo;
          float g[V];
          Vec16<T>::load(dy + o * p.C + cc * V, g);
          if (IS_MAX && ymask) {
            // max-pool of a ReLU output: the selected element IS the pooled value, so "the input passed its
            // ReLU" can be read from the (window-times smaller) pooled tensor instead of the input
            float yv[V];
            Vec16<T>::load(ymask + o * p.C + cc * V, yv);
#pragma unroll
            for (int k = 0; k < V; ++k) g[k] = yv[k] > 0.f ? g[k] : 0.f;
          }
          if (IS_MAX && sizeof(IdxT) == 2) {
            const int tap = (a * p.kh + b) * p.kw + c;
            const IdxT* am = argmax + o * p.C + cc * V;
#pragma unroll
            for (int k = 0; k < V; ++k)
              if ((int)am[k] == tap) acc[k] += g[k];
          } else if (IS_MAX) {
            const int tap = (a * p.kh + b) * p.kw + c;
            const uint8_t* am = reinterpret_cast<const uint8_t*>(argmax) + o * p.C + cc * V;
            uint32_t w0 = *reinterpret_cast<const uint32_t*>(am);
            uint32_t w1 = V == 8 ? *reinterpret_cast<const uint32_t*>(am + 4) : 0u;
#pragma unroll
            for (int k = 0; k < V; ++k) {
              const int sel = (int)(((k < 4 ? w0 : w1) >> (8 * (k & 3))) & 0xff);
              if (sel == tap) acc[k] += g[k];
            }
          } else {
#pragma unroll
            for (int k = 0; k < V; ++k) acc[k] += g[k] * inv_window;
          }
        }
      }
    }
    const long long off = q * p.C + cc * V;
    if (add) {
      float a2[V];
      Vec16<T>::load(add + off, a2);
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] += a2[k];
    }
    if (mask) {
      float mk[V];
      Vec16<T>::load(mask + off, mk);
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] = mk[k] > 0.f ? acc[k] : 0.f;
    }
    Vec16<T>::store(dx + off, acc);
  }
}

// Max-pool backward for the window shapes the models use (pool1 1x3x3 / 1x2x2, pool2 2x1x1 / 2x1x1, the non-local
// 1x2x2 / 1x2x2 pools), with the window geometry as compile-time constants: the generic kernel above spends its time
// in 64-bit index divisions and data-dependent window loops (pool1 of the 8-clip step: 532 us for 0.57 GB of
// traffic).  One workgroup per input row (n, t, h): the row decode is scalar, the candidate windows of a position
// are at most ceil(k / s) per dimension and unroll completely, and the accumulation order (windows ascending in
// t, h, w) is the one of the generic kernel, so both give identical bits.
// 16 bytes of T as floats
template <typename T>
__device__ __forceinline__ void unpack_vec(const uint4& t, float (&v)[Vec16<T>::N]) {
  if (sizeof(T) == 4) {
    v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y); v[2 % Vec16<T>::N] = __uint_as_float(t.z); v[3 % Vec16<T>::N] = __uint_as_float(t.w);
  } else {
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[(2 * i) % Vec16<T>::N] = Elem<T>::lo(w[i]);
      v[(2 * i + 1) % Vec16<T>::N] = Elem<T>::hi(w[i]);
    }
  }
}

template <typename T, int KT, int KH, int KW, int ST, int SH, int SW, bool YMASK>
__global__ __launch_bounds__(128) void maxpool_bwd_fixed_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ argmax,
                                                                T* dx, const T* add, const T* __restrict__ mask, PoolP p,
                                                                const T* __restrict__ ymask) {
  constexpr int V = Vec16<T>::N;
  constexpr int CT = (KT + ST - 1) / ST, CH = (KH + SH - 1) / SH, CW = (KW + SW - 1) / SW;
  const int cchunks = p.C / V;
  const int row = blockIdx.x;                       // (n * Ti + ti) * Hi + hi
  const int hi = row % p.Hi;
  const int r2 = row / p.Hi;
  const int ti = r2 % p.Ti;
  const int n = r2 / p.Ti;
  const int items = p.Wi * cchunks;
  const int to_hi = (ti + p.pt) / ST, ho_hi = (hi + p.ph) / SH;
  for (int it = threadIdx.x; it < items; it += 128) {
    // item -> (w, channel chunk), w running through one residue class mod SW after the other
    const int wj = it / cchunks, cc = it - wj * cchunks;
    int wi = wj;
    if (SW > 1) {
      // classes 0 .. SW-1 hold ceil / floor (Wi / SW) positions; walk them in order
      int cls = 0, base = 0;
#pragma unroll
      for (int q = 0; q < SW - 1; ++q) {
        const int cnt = (p.Wi - q + SW - 1) / SW;
        if (wj >= base + cnt) { base += cnt; cls = q + 1; }
      }
      wi = (wj - base) * SW + cls;
    }
    const int wo_hi = (wi + p.pw) / SW;
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    // Candidate windows in t and h are workgroup-uniform: those that do not exist are skipped by scalar branches.
    // The w candidates of a position are all loaded, unconditionally and before any of them is used (a window that
    // does not exist reads window 0 and is compared against a tap no arg-max holds): the loads of a position are
    // in flight together instead of one round trip after the other.
#pragma unroll
    for (int jt = CT - 1; jt >= 0; --jt) {
      const int to = to_hi - jt, a = ti + p.pt - to * ST;
      if (to < 0 || to >= p.To || a >= KT) continue;
#pragma unroll
      for (int jh = CH - 1; jh >= 0; --jh) {
        const int ho = ho_hi - jh, b = hi + p.ph - ho * SH;
        if (ho < 0 || ho >= p.Ho || b >= KH) continue;
        const int orow = ((n * p.To + to) * p.Ho + ho) * p.Wo;
        uint4 gq[CW], yq[CW];
        uint2 aq[CW];
        int tapq[CW];
#pragma unroll
        for (int jw = 0; jw < CW; ++jw) {
          const int wo = wo_hi - (CW - 1 - jw), c = wi + p.pw - wo * SW;
          const bool ok = wo >= 0 && wo < p.Wo && c < KW;
          const long long o = ok ? (long long)(orow + wo) * p.C + cc * V : (long long)(cc * V);
          tapq[jw] = ok ? (a * KH + b) * KW + c : 255;
          if (sizeof(T) == 2) {
            gq[jw] = *reinterpret_cast<const uint4*>(dy + o);
            if (YMASK) yq[jw] = *reinterpret_cast<const uint4*>(ymask + o);
            aq[jw] = *reinterpret_cast<const uint2*>(argmax + o);
          } else {
            gq[jw] = *reinterpret_cast<const uint4*>(dy + o);
            if (YMASK) yq[jw] = *reinterpret_cast<const uint4*>(ymask + o);
            aq[jw] = make_uint2(*reinterpret_cast<const uint32_t*>(argmax + o), 0u);
          }
        }
#pragma unroll
        for (int jw = 0; jw < CW; ++jw) {
          float g[V];
          unpack_vec<T>(gq[jw], g);
          if (YMASK) {
            float yv[V];
            unpack_vec<T>(yq[jw], yv);
#pragma unroll
            for (int k = 0; k < V; ++k) g[k] = yv[k] > 0.f ? g[k] : 0.f;
          }
#pragma unroll
          for (int k = 0; k < V; ++k) {
            const int sel = (int)(((k < 4 ? aq[jw].x : aq[jw].y) >> (8 * (k & 3))) & 0xff);
            acc[k] += sel == tapq[jw] ? g[k] : 0.f;
          }
        }
      }
    }
    const long long off = ((long long)row * p.Wi + wi) * p.C + cc * V;
    if (add) {
      float a2[V];
      Vec16<T>::load(add + off, a2);
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] += a2[k];
    }
    if (mask) {
      float mk[V];
      Vec16<T>::load(mask + off, mk);
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] = mk[k] > 0.f ? acc[k] : 0.f;
    }
    Vec16<T>::store(dx + off, acc);
  }
}

// pool1 (1x3x3 windows, stride 1x2x2): overlapping windows, so the row kernel above reads every pooled row for up to
// three input rows (PMC: 762 MB fetched for 257 MB of pooled gradient / arg-max / pooled output at 8 clips).  Here a
// workgroup owns a band of 2R input rows of one frame: the R + 2 pooled rows the band can touch are staged in LDS once
// -- the gradient already masked by the pooled output's sign, the arg-max bytes beside it -- and every input position
// then sums its (at most 2 x 2) candidate windows out of LDS, in the order of the kernels above (windows ascending in h,
// then w), so the bits are the same.  Pooled rows are read (R + 2) / R times per 2R input rows instead of 3 times per 2.
template <typename T, bool YMASK>
__global__ __launch_bounds__(256) void maxpool_bwd_band_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ argmax,
                                                               T* dx, const T* add, const T* __restrict__ mask, PoolP p,
                                                               const T* __restrict__ ymask, int R, int bands) {
  constexpr int V = Vec16<T>::N;                      // elements per 16 bytes
  extern __shared__ __attribute__((aligned(16))) unsigned char band_lds[];
  const int cch = p.C / V;
  const int NR = R + 2;
  const int rowv = p.Wo * cch;                        // 16-byte vectors per pooled row
  uint4* g_l = reinterpret_cast<uint4*>(band_lds);
  unsigned char* t_l = band_lds + (size_t)NR * rowv * 16;
  const int band = blockIdx.x % bands, frame = blockIdx.x / bands;       // frame = n * Ti + ti  (kt = st = 1, pt = 0)
  const int h0 = band * 2 * R;
  const int hb = (h0 + p.ph) / 2 - 1;                 // first pooled row staged
  for (int i = threadIdx.x; i < NR * rowv; i += 256) {
    const int r = i / rowv, ho = hb + r;
    uint4 g = make_uint4(0u, 0u, 0u, 0u);
    uint2 tp = make_uint2(0xffffffffu, 0xffffffffu);
    if (ho >= 0 && ho < p.Ho) {
      const long long o = ((long long)(frame * p.Ho + ho) * rowv + (i - r * rowv)) * V;
      g = *reinterpret_cast<const uint4*>(dy + o);
      tp = V == 8 ? *reinterpret_cast<const uint2*>(argmax + o) : make_uint2(*reinterpret_cast<const uint32_t*>(argmax + o), 0u);
      if (YMASK) {
        const uint4 yq = *reinterpret_cast<const uint4*>(ymask + o);
        float yv[V];
        unpack_vec<T>(yq, yv);
        uint32_t w[4] = {g.x, g.y, g.z, g.w};
        if (sizeof(T) == 4) {
#pragma unroll
          for (int k = 0; k < 4; ++k) w[k] = yv[k % V] > 0.f ? w[k] : 0u;
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            w[k] = (yv[(2 * k) % V] > 0.f ? (w[k] & 0xffffu) : 0u) | (yv[(2 * k + 1) % V] > 0.f ? (w[k] & 0xffff0000u) : 0u);
        }
        g = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
    g_l[i] = g;
    if (V == 8) *reinterpret_cast<uint2*>(t_l + (size_t)i * 8) = tp;
    else *reinterpret_cast<uint32_t*>(t_l + (size_t)i * 4) = tp.x;
  }
  __syncthreads();
  const int rows = min(2 * R, p.Hi - h0);
  const int rowi = p.Wi * cch;                        // items per input row
  for (int it = threadIdx.x; it < rows * rowi; it += 256) {
    const int hr = it / rowi, rem = it - hr * rowi;
    const int wi = rem / cch, cc = rem - wi * cch;
    const int hi = h0 + hr;
    const int ho_hi = (hi + p.ph) / 2, wo_hi = (wi + p.pw) / 2;
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
#pragma unroll
    for (int jh = 1; jh >= 0; --jh) {
      const int ho = ho_hi - jh, b = hi + p.ph - ho * 2;
      const bool okh = ho >= 0 && ho < p.Ho && b < 3;
      const int lr = okh ? ho - hb : 0;
#pragma unroll
      for (int jw = 1; jw >= 0; --jw) {
        const int wo = wo_hi - jw, c = wi + p.pw - wo * 2;
        const bool ok = okh && wo >= 0 && wo < p.Wo && c < 3;
        const int idx = ok ? (lr * p.Wo + wo) * cch + cc : cc;
        const int tap = ok ? b * 3 + c : 255;
        const uint4 gq = g_l[idx];
        uint2 aq;
        if (V == 8) aq = *reinterpret_cast<const uint2*>(t_l + (size_t)idx * 8);
        else aq = make_uint2(*reinterpret_cast<const uint32_t*>(t_l + (size_t)idx * 4), 0u);
        // keep the elements whose arg-max byte equals this tap, on the packed words: bytes of (arg-max ^ tap) that
        // are zero -> 0xff, widened to the element width by a byte permute, AND-ed onto the gradient bits (a dropped
        // element adds +0.0f exactly like the select of the kernels above)
        const uint32_t tapw = (uint32_t)tap * 0x01010101u;
        uint32_t fullb[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const uint32_t t = (q ? aq.y : aq.x) ^ tapw;
          const uint32_t nz = (((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t) & 0x80808080u;     // 0x80 where the byte differs
          fullb[q] = ((nz ^ 0x80808080u) >> 7) * 0xffu;                                   // 0xff where it matches
        }
        uint32_t w[4] = {gq.x, gq.y, gq.z, gq.w};
        if (sizeof(T) == 4) {
#pragma unroll
          for (int k = 0; k < 4; ++k) w[k] &= __builtin_amdgcn_perm(fullb[0], fullb[0], 0x01010101u * (uint32_t)k);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            w[k] &= __builtin_amdgcn_perm(fullb[k >> 1], fullb[k >> 1], (k & 1) ? 0x03030202u : 0x01010000u);
        }
        float g[V];
        unpack_vec<T>(make_uint4(w[0], w[1], w[2], w[3]), g);
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] += g[k];
      }
    }
    const long long off = (((long long)frame * p.Hi + hi) * p.Wi + wi) * p.C + cc * V;
    if (add) {
      float a2[V];
      Vec16<T>::load(add + off, a2);
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] += a2[k];
    }
    if (mask) {
      float mk[V];
      Vec16<T>::load(mask + off, mk);
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] = mk[k] > 0.f ? acc[k] : 0.f;
    }
    Vec16<T>::store(dx + off, acc);
  }
}

// launches the fixed-shape kernel when the window is one of the compiled shapes (one-byte arg-max); false = use the generic one
template <typename T>
bool launch_maxpool_bwd_fixed(const PoolP& p, const void* dy, const void* argmax, void* dx, const void* add, const void* mask,
                              const void* ymask, hipStream_t s) {
  const long long rows = (long long)p.N * p.Ti * p.Hi;
  if (rows >= (1ll << 31) || (long long)p.N * p.To * p.Ho * p.Wo >= (1ll << 31)) return false;
  const dim3 grid((unsigned)rows), block(128);
  // (with a residual operand and a ReLU mask streamed beside it the row kernel is the faster one: 271 vs 357 us)
  if (p.kt == 1 && p.kh == 3 && p.kw == 3 && p.st == 1 && p.sh == 2 && p.sw == 2 && p.pt == 0 && p.To == p.Ti && !add &&
      !mask) {
    // band kernel: as many input-row pairs per workgroup as 64 KB of LDS hold pooled rows for (R + 2 rows staged)
    const long long row_bytes = (long long)p.Wo * p.C * (sizeof(T) + 1);
    // R = 3: measured 226 us at R = 2 and 3, 259 us at R = 4 (two 64 KB workgroups per CU leave the load and the store
    // phases of a CU unbalanced) against 285 us for the row kernel, pool1 of the 8-clip step in isolation
    const int R = (int)std::min<long long>(3, 65536 / row_bytes - 2);
    if (R >= 2) {
      const int bands = (p.Hi + 2 * R - 1) / (2 * R);
      const long long wgs = (long long)p.N * p.Ti * bands;
      const size_t lds = (size_t)(R + 2) * row_bytes;
      if (wgs < (1ll << 31)) {
        if (ymask)
          hipLaunchKernelGGL((maxpool_bwd_band_kernel<T, true>), dim3((unsigned)wgs), dim3(256), lds, s, (const T*)dy,
                             (const uint8_t*)argmax, (T*)dx, (const T*)add, (const T*)mask, p, (const T*)ymask, R, bands);
        else
          hipLaunchKernelGGL((maxpool_bwd_band_kernel<T, false>), dim3((unsigned)wgs), dim3(256), lds, s, (const T*)dy,
                             (const uint8_t*)argmax, (T*)dx, (const T*)add, (const T*)mask, p, (const T*)ymask, R, bands);
        return true;
      }
    }
  }
#define VLFB_POOL_FIXED(KT_, KH_, KW_, ST_, SH_, SW_)                                                                      \
  if (p.kt == KT_ && p.kh == KH_ && p.kw == KW_ && p.st == ST_ && p.sh == SH_ && p.sw == SW_) {                           \
    if (ymask)                                                                                                           \
      hipLaunchKernelGGL((maxpool_bwd_fixed_kernel<T, KT_, KH_, KW_, ST_, SH_, SW_, true>), grid, block, 0, s,           \
                         (const T*)dy, (const uint8_t*)argmax, (T*)dx, (const T*)add, (const T*)mask, p, (const T*)ymask); \
    else                                                                                                                 \
      hipLaunchKernelGGL((maxpool_bwd_fixed_kernel<T, KT_, KH_, KW_, ST_, SH_, SW_, false>), grid, block, 0, s,          \
                         (const T*)dy, (const uint8_t*)argmax, (T*)dx, (const T*)add, (const T*)mask, p, (const T*)ymask); \
    return true;                                                                                                         \
  }
  VLFB_POOL_FIXED(1, 3, 3, 1, 2, 2)
  VLFB_POOL_FIXED(2, 1, 1, 2, 1, 1)
  VLFB_POOL_FIXED(1, 2, 2, 1, 2, 2)
#undef VLFB_POOL_FIXED
  return false;
}

template <typename T>
__global__ void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, PoolP p) {
  constexpr int V = Vec16<T>::N;
  const int cchunks = p.C / V;
  const long long total = (long long)p.N * p.To * p.Ho * p.Wo * cchunks;
  const float inv = 1.0f / (float)(p.kt * p.kh * p.kw);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cchunks);
    long long o = i / cchunks;
    const int wo = (int)(o % p.Wo); long long r = o / p.Wo;
    const int ho = (int)(r % p.Ho); r /= p.Ho;
    const int to = (int)(r % p.To);
    const int n = (int)(r / p.To);
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    for (int a = 0; a < p.kt; ++a)
      for (int b = 0; b < p.kh; ++b)
        for (int c = 0; c < p.kw; ++c) {
          const int ti = to * p.st + a, hi = ho * p.sh + b, wi = wo * p.sw + c;
          float v[V];
          Vec16<T>::load(x + ((((long long)n * p.Ti + ti) * p.Hi + hi) * p.Wi + wi) * p.C + cc * V, v);
#pragma unroll
          for (int k = 0; k < V; ++k) acc[k] += v[k];
        }
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] *= inv;
    Vec16<T>::store(y + o * p.C + cc * V, acc);
  }
}

// global average: y[n][c] = mean over `rows` positions.  block = (8 chunk lanes) x (32 row lanes)
template <typename T>
__global__ void global_avgpool_kernel(const T* __restrict__ x, T* __restrict__ y, long long rows,
                                      int C) {
  constexpr int V = Vec16<T>::N;
  __shared__ float red[32][8 * 8 + 1];
  const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int n = blockIdx.y;
  const int c0 = (blockIdx.x * 8 + cl) * V;
  float acc[V];
#pragma unroll
  for (int k = 0; k < V; ++k) acc[k] = 0.f;
  if (c0 < C) {
    const T* base = x + (long long)n * rows * C + c0;
    for (long long r = rl; r < rows; r += 32) {
      float v[V];
      Vec16<T>::load(base + r * C, v);
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] += v[k];
    }
  }
#pragma unroll
  for (int k = 0; k < V; ++k) red[rl][cl * 8 + k] = acc[k];
  __syncthreads();
  if (rl == 0 && c0 < C) {
    float inv = 1.0f / (float)rows;
    float out[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
      float s = 0.f;
      for (int j = 0; j < 32; ++j) s += red[j][cl * 8 + k];
      out[k] = s * inv;
    }
    Vec16<T>::store(y + (long long)n * C + c0, out);
  }
}

// ---------------------------------------------------------------------------------------------
// row softmax (one wave per row; rows stay L1/L2-resident between the passes)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// 4 consecutive elements of T (16 / 8 bytes, naturally aligned) as floats
template <typename T>
__device__ __forceinline__ void load_elems4(const T* p, float (&v)[4]) {
  if (sizeof(T) == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = Elem<T>::lo(t.x); v[1] = Elem<T>::hi(t.x);
    v[2] = Elem<T>::lo(t.y); v[3] = Elem<T>::hi(t.y);
  }
}

template <typename T>
__global__ void softmax_fwd_kernel(const float* __restrict__ s, T* __restrict__ p, long long rows,
                                   int cols, float scale) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long r = wave; r < rows; r += nwaves) {
    const float* sr = s + r * cols;
    float m = -INFINITY;
    for (int c = lane; c < cols; c += 64) m = fmaxf(m, sr[c] * scale);
    m = wave_max(m);
    float sum = 0.f;
    for (int c = lane; c < cols; c += 64) sum += __expf(sr[c] * scale - m);
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    T* pr = p + r * cols;
    for (int c = lane; c < cols; c += 64) Elem<T>::st(pr + c, __expf(sr[c] * scale - m) * inv);
  }
}
// Rows of at most 64*4*NV columns (cols % 4 == 0): the row is read ONCE with 16-byte loads and stays in
// registers (same arithmetic as above: max, exp, sum, scale), the result leaves as 4 packed elements
// per lane.  The non-local blocks have 784 (train) / 1024 (test crop) / 1568 (64 frames) columns.
template <typename T, int NV>
__global__ void softmax_fwd_rowreg_kernel(const float* __restrict__ s, T* __restrict__ p, long long rows,
                                          int cols, float scale) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  const int nvec = cols >> 2;
  for (long long r = wave; r < rows; r += nwaves) {
    const float4* sr = reinterpret_cast<const float4*>(s + r * cols);
    float4 v[NV];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        float4 t = sr[c];
        t.x *= scale; t.y *= scale; t.z *= scale; t.w *= scale;
        v[i] = t;
        m = fmaxf(m, fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w)));
      }
    }
    m = wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (lane + 64 * i < nvec) {
        v[i].x = __expf(v[i].x - m); v[i].y = __expf(v[i].y - m);
        v[i].z = __expf(v[i].z - m); v[i].w = __expf(v[i].w - m);
        sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      }
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    T* pr = p + r * cols;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        if (sizeof(T) == 4) {
          *reinterpret_cast<float4*>(pr + 4 * c) = make_float4(v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv);
        } else {
          *reinterpret_cast<uint2*>(pr + 4 * c) = make_uint2(Elem<T>::pack2(v[i].x * inv, v[i].y * inv), Elem<T>::pack2(v[i].z * inv, v[i].w * inv));
        }
      }
    }
  }
}
template <typename T>
__global__ void softmax_bwd_kernel(const float* __restrict__ dp, const T* __restrict__ p,
                                   T* __restrict__ ds, long long rows, int cols, float scale) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long r = wave; r < rows; r += nwaves) {
    const float* dr = dp + r * cols;
    const T* pr = p + r * cols;
    float dot = 0.f;
    for (int c = lane; c < cols; c += 64) dot += dr[c] * Elem<T>::ld(pr + c);
    dot = wave_sum(dot);
    T* o = ds + r * cols;
    for (int c = lane; c < cols; c += 64)
      Elem<T>::st(o + c, scale * Elem<T>::ld(pr + c) * (dr[c] - dot));
  }
}

// backward with the row resident in registers (same conditions as softmax_fwd_rowreg_kernel)
template <typename T, int NV, typename TD = T>
__global__ void softmax_bwd_rowreg_kernel(const float* __restrict__ dp, const T* __restrict__ p,
                                          TD* __restrict__ ds, long long rows, int cols, float scale) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  const int nvec = cols >> 2;
  for (long long r = wave; r < rows; r += nwaves) {
    const float4* dr = reinterpret_cast<const float4*>(dp + r * cols);
    const T* pr = p + r * cols;
    float4 d[NV], q[NV];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        d[i] = dr[c];
        float t[4];
        load_elems4<T>(pr + 4 * c, t);
        q[i] = make_float4(t[0], t[1], t[2], t[3]);
        dot += (d[i].x * q[i].x + d[i].y * q[i].y) + (d[i].z * q[i].z + d[i].w * q[i].w);
      }
    }
    dot = wave_sum(dot);
    TD* o = ds + r * cols;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        const float a = scale * q[i].x * (d[i].x - dot), b = scale * q[i].y * (d[i].y - dot);
        const float e = scale * q[i].z * (d[i].z - dot), f = scale * q[i].w * (d[i].w - dot);
        if (sizeof(TD) == 4) *reinterpret_cast<float4*>(o + 4 * c) = make_float4(a, b, e, f);
        else *reinterpret_cast<uint2*>(o + 4 * c) = make_uint2(Elem<TD>::pack2(a, b), Elem<TD>::pack2(e, f));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// small elementwise
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void add_kernel(const T* a, const T* b, T* y, const T* __restrict__ mask, long long n,
                           int relu) {
  constexpr int V = Vec16<T>::N;
  const long long nv = n / V;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv;
       i += (long long)gridDim.x * blockDim.x) {
    float x[V], z[V];
    Vec16<T>::load(a + i * V, x);
    if (b) {
      Vec16<T>::load(b + i * V, z);
#pragma unroll
      for (int k = 0; k < V; ++k) x[k] += z[k];
    }
    if (relu) {
#pragma unroll
      for (int k = 0; k < V; ++k) x[k] = fmaxf(x[k], 0.f);
    }
    if (mask) {
      Vec16<T>::load(mask + i * V, z);
#pragma unroll
      for (int k = 0; k < V; ++k) x[k] = z[k] > 0.f ? x[k] : 0.f;
    }
    Vec16<T>::store(y + i * V, x);
  }
  // tail
  for (long long i = nv * V + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float x = Elem<T>::ld(a + i) + (b ? Elem<T>::ld(b + i) : 0.f);
    if (relu) x = fmaxf(x, 0.f);
    if (mask) x = Elem<T>::ld(mask + i) > 0.f ? x : 0.f;
    Elem<T>::st(y + i, x);
  }
}

// column sums: block = 16-byte channel-chunk lanes x row lanes over one row slab; 16-byte loads,
// LDS reduction over the row lanes, then one fp32 atomic per channel and slab.
// PARTIAL: every row slab leaves its sums in out[slab][cols] instead (a later pass folds the slabs in order: deterministic)
template <typename T, bool PARTIAL = false>
__global__ void colsum_kernel(const T* __restrict__ g, long long rows, int cols, long long ld,
                              float* __restrict__ out) {
  constexpr int V = Vec16<T>::N;
  constexpr int CL = 8;                 // chunk lanes (8 * V channels per block)
  constexpr int RL = 256 / CL;          // row lanes
  __shared__ float red[RL][CL * 8 + 1];
  const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
  const int c0 = (blockIdx.x * CL + cl) * V;
  const long long per = (rows + gridDim.y - 1) / gridDim.y;
  const long long r0 = (long long)blockIdx.y * per, r1 = min(rows, r0 + per);
  float acc[V];
#pragma unroll
  for (int k = 0; k < V; ++k) acc[k] = 0.f;
  if (c0 < cols) {
    for (long long r = r0 + rl; r < r1; r += RL) {
      float v[V];
      Vec16<T>::load(g + r * ld + c0, v);
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] += v[k];
    }
  }
#pragma unroll
  for (int k = 0; k < V; ++k) red[rl][cl * 8 + k] = acc[k];
  __syncthreads();
  if (rl < V && c0 < cols) {            // row lane k sums channel k of this chunk
    float sum = 0.f;
    for (int j = 0; j < RL; ++j) sum += red[j][cl * 8 + rl];
    if (PARTIAL) out[(long long)blockIdx.y * cols + c0 + rl] = sum;
    else atomicAdd(out + c0 + rl, sum);
  }
}

// strided 2-D copy (Concat along channels and its backward): dst[r*ldd + c] = src[r*lds + c]
template <typename T>
__global__ void copy2d_kernel(const T* __restrict__ src, long long lds, T* __restrict__ dst,
                              long long ldd, long long rows, long long cols) {
  const long long total = rows * cols;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cols, c = i - r * cols;
    dst[r * ldd + c] = src[r * lds + c];
  }
}

__global__ void zero_kernel(float* p, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    p[i] = 0.f;
}

PoolP to_poolp(const vlfb_pool_desc* d) {
  PoolP p;
  p.N = d->N; p.Ti = d->Ti; p.Hi = d->Hi; p.Wi = d->Wi; p.C = d->C;
  p.To = d->To; p.Ho = d->Ho; p.Wo = d->Wo;
  p.kt = d->kt; p.kh = d->kh; p.kw = d->kw; p.st = d->st; p.sh = d->sh; p.sw = d->sw;
  p.pt = d->pt; p.ph = d->ph; p.pw = d->pw;
  return p;
}
int check_pool(const vlfb_pool_desc* d) {
  VLFB_REQUIRE(d->dtype == VLFB_F32 || is16(d->dtype) || d->dtype == VLFB_F16PAIR, "pool: bad dtype");
  const int v = d->dtype == VLFB_F32 ? 4 : 8;
  VLFB_REQUIRE(d->C % v == 0, "pool: C=%d must be a multiple of %d", d->C, v);
  VLFB_REQUIRE(d->kt * d->kh * d->kw <= 65535, "pool: window too large for a 16-bit argmax");
  VLFB_REQUIRE(d->N > 0 && d->To > 0 && d->Ho > 0 && d->Wo > 0, "pool: empty output");
  return VLFB_OK;
}

}  // namespace
}  // namespace vlfb

using namespace vlfb;

extern "C" const char* vlfb_last_error(void) { return g_err; }
extern "C" int vlfb_version(void) { return 100; }
extern "C" int vlfb_dtype_size(int dtype) { return dtype == VLFB_F32 ? 4 : is16(dtype) ? 2 : 0; }

extern "C" int vlfb_affine_nd_fwd(const float* x, const float* scale, const float* bias, float* y,
                                  int64_t n, int64_t c, int64_t inner, vlfb_stream_t stream) {
  VLFB_REQUIRE(x && scale && bias && y, "affine_nd_fwd: null pointer");
  VLFB_REQUIRE(n > 0 && c > 0 && inner > 0, "affine_nd_fwd: empty tensor");
  VLFB_REQUIRE(n * c * inner < (1ll << 31), "affine_nd_fwd: numel must stay below 2^31 (affine_nd_op.cu:69)");
  const long long rows = n * c;
  dim3 grid((unsigned)grid_for((inner + 3) / 4, 256, 64), (unsigned)(rows < 16384 ? rows : 16384));
  hipLaunchKernelGGL(affine_nd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, scale, bias,
                     y, rows, (int)c, (long long)inner);
  return check_launch("affine_nd_fwd");
}
extern "C" int vlfb_affine_nd_bwd(const float* dy, const float* scale, float* dx, int64_t n,
                                  int64_t c, int64_t inner, vlfb_stream_t stream) {
  VLFB_REQUIRE(dy && scale && dx, "affine_nd_bwd: null pointer");
  VLFB_REQUIRE(n > 0 && c > 0 && inner > 0, "affine_nd_bwd: empty tensor");
  VLFB_REQUIRE(n * c * inner < (1ll << 31), "affine_nd_bwd: numel must stay below 2^31");
  const long long rows = n * c;
  dim3 grid((unsigned)grid_for((inner + 3) / 4, 256, 64), (unsigned)(rows < 16384 ? rows : 16384));
  hipLaunchKernelGGL(affine_nd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, dy, scale,
                     (const float*)nullptr, dx, rows, (int)c, (long long)inner);
  return check_launch("affine_nd_bwd");
}

extern "C" int vlfb_ncthw_to_nthwc(const float* src, void* dst, int dtype, int64_t n, int64_t c,
                                   int64_t thw, int64_t c_pad, vlfb_stream_t stream) {
  VLFB_REQUIRE(src && dst && c_pad >= c && n > 0 && c > 0 && thw > 0, "ncthw_to_nthwc: bad args");
  int grid = grid_for(n * thw, 256);
  if (dtype == VLFB_F32)
    hipLaunchKernelGGL(ncthw_to_nthwc_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       src, (float*)dst, (long long)n, (int)c, (long long)thw, (int)c_pad);
  else if (is16(dtype))
    VLFB_WITH_T16(dtype, hipLaunchKernelGGL(ncthw_to_nthwc_kernel<T16>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       src, (T16*)dst, (long long)n, (int)c, (long long)thw, (int)c_pad));
  else return set_error(VLFB_ERR_ARG, "ncthw_to_nthwc: bad dtype");
  return check_launch("ncthw_to_nthwc");
}
extern "C" int vlfb_nthwc_to_ncthw(const void* src, float* dst, int dtype, int64_t n, int64_t c,
                                   int64_t thw, vlfb_stream_t stream) {
  VLFB_REQUIRE(src && dst && n > 0 && c > 0 && thw > 0, "nthwc_to_ncthw: bad args");
  int grid = grid_for(n * c * thw, 256);
  if (dtype == VLFB_F32)
    hipLaunchKernelGGL(nthwc_to_ncthw_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const float*)src, dst, (long long)n, (int)c, (long long)thw);
  else if (is16(dtype))
    VLFB_WITH_T16(dtype, hipLaunchKernelGGL(nthwc_to_ncthw_kernel<T16>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const T16*)src, dst, (long long)n, (int)c, (long long)thw));
  else return set_error(VLFB_ERR_ARG, "nthwc_to_ncthw: bad dtype");
  return check_launch("nthwc_to_ncthw");
}
// fp32 -> fp16 copy for the "mix" engine's backward (16 bytes read, 8 written per lane and step; positive values stay
// positive, f2h_pos)
__global__ void half_copy_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, long long n4, long long n, bool nt) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const uint4 u = stream_ld16(nt, reinterpret_cast<const float4*>(src) + i);
    stream_st8(nt, reinterpret_cast<uint2*>(dst) + i, make_uint2(pack_h2_pos(__uint_as_float(u.x), __uint_as_float(u.y)), pack_h2_pos(__uint_as_float(u.z), __uint_as_float(u.w))));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[(n4 << 2) + threadIdx.x] = f2h_pos(src[(n4 << 2) + threadIdx.x]);
}
extern "C" int vlfb_half_copy(const float* src, void* dst, int64_t n, vlfb_stream_t stream) {
  VLFB_REQUIRE(src && dst && n >= 0, "half_copy: bad args");
  if (n == 0) return VLFB_OK;
  hipLaunchKernelGGL(half_copy_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     (unsigned short*)dst, (long long)(n / 4), (long long)n, stream_nt());
  return check_launch("half_copy");
}
extern "C" int vlfb_pair_split(const float* src, void* dst_pair, int64_t n, vlfb_stream_t stream) {
  VLFB_REQUIRE(src && dst_pair && n >= 0 && n % 8 == 0, "pair_split: bad args (n must be a multiple of 8)");
  if (n == 0) return VLFB_OK;
  hipLaunchKernelGGL(pair_split_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, src, (unsigned short*)dst_pair, (long long)n, stream_nt());
  return check_launch("pair_split");
}
extern "C" int vlfb_pair_join(const void* src_pair, float* dst, int64_t n, vlfb_stream_t stream) {
  VLFB_REQUIRE(src_pair && dst && n >= 0 && n % 8 == 0, "pair_join: bad args (n must be a multiple of 8)");
  if (n == 0) return VLFB_OK;
  hipLaunchKernelGGL(pair_join_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)src_pair, dst, (long long)n, stream_nt());
  return check_launch("pair_join");
}
extern "C" int vlfb_cast(const void* src, int sd, void* dst, int dd, int64_t n, vlfb_stream_t stream) {
  VLFB_REQUIRE(src && dst && n >= 0, "cast: bad args");
  if (n == 0) return VLFB_OK;
  int grid = grid_for(n, 256);
  hipStream_t s = (hipStream_t)stream;
  if (sd == VLFB_F32 && is16(dd))
    VLFB_WITH_T16(dd, hipLaunchKernelGGL((cast_kernel<float, T16>), dim3(grid), dim3(256), 0, s, (const float*)src, (T16*)dst, (long long)n));
  else if (is16(sd) && dd == VLFB_F32)
    VLFB_WITH_T16(sd, hipLaunchKernelGGL((cast_kernel<T16, float>), dim3(grid), dim3(256), 0, s, (const T16*)src, (float*)dst, (long long)n));
  else if (sd == VLFB_F32 && dd == VLFB_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), dim3(grid), dim3(256), 0, s, (const float*)src, (float*)dst, (long long)n);
  else if (is16(sd) && sd == dd)
    VLFB_WITH_T16(dd, hipLaunchKernelGGL((cast_kernel<T16, T16>), dim3(grid), dim3(256), 0, s, (const T16*)src, (T16*)dst, (long long)n));
  else return set_error(VLFB_ERR_ARG, "cast: bad dtypes %d -> %d", sd, dd);
  return check_launch("cast");
}
extern "C" int vlfb_transpose2d(const void* src, void* dst, int dtype, int64_t batch, int64_t rows,
                                int64_t cols, vlfb_stream_t stream) {
  VLFB_REQUIRE(src && dst && batch > 0 && rows > 0 && cols > 0, "transpose2d: bad args");
  VLFB_REQUIRE(batch < 65536 && (rows + 31) / 32 < 65536, "transpose2d: grid too large");
  dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32), (unsigned)batch);
  dim3 block(32, 8);
  if (dtype == VLFB_F32)
    hipLaunchKernelGGL(transpose2d_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)src, (float*)dst, (long long)rows, (long long)cols);
  else if (is16(dtype))
    VLFB_WITH_T16(dtype, hipLaunchKernelGGL(transpose2d_kernel<T16>, grid, block, 0, (hipStream_t)stream, (const T16*)src, (T16*)dst, (long long)rows, (long long)cols));
  else return set_error(VLFB_ERR_ARG, "transpose2d: bad dtype");
  return check_launch("transpose2d");
}
extern "C" int vlfb_weight_prep(const float* w, const float* scale, void* w_fprop, void* w_dgrad,
                                int dtype, int64_t cout, int64_t taps, int64_t cin,
                                vlfb_stream_t stream) {
  VLFB_REQUIRE(w && (w_fprop || w_dgrad) && cout > 0 && taps > 0 && cin > 0, "weight_prep: bad args");
  VLFB_REQUIRE(dtype == VLFB_F32 || is16(dtype) || dtype == VLFB_SPLIT || dtype == VLFB_MIX || dtype == VLFB_MIX_W2 || dtype == VLFB_MIXH ||
                   dtype == VLFB_MIXH_W2 || dtype == VLFB_MIX_W2I || dtype == VLFB_MIXH_W2I, "weight_prep: bad dtype");
  VLFB_REQUIRE((dtype != VLFB_MIX_W2I && dtype != VLFB_MIXH_W2I) || !w_dgrad || cout % 64 == 0, "weight_prep: the interleaved two-term DGRAD copy needs Cout %% 64 == 0");
  VLFB_REQUIRE(taps < 65536, "weight_prep: too many taps");
  hipStream_t s = (hipStream_t)stream;
  const long long total = cout * taps * cin;
  if (w_fprop) {
    int grid = grid_for(total, 256);
    if (dtype == VLFB_F32)
      hipLaunchKernelGGL(weight_prep_fprop_kernel<float>, dim3(grid), dim3(256), 0, s, w, scale, w_fprop, (long long)(taps * cin), total);
    else if (dtype == VLFB_SPLIT || dtype == VLFB_MIX || dtype == VLFB_MIX_W2 || dtype == VLFB_MIX_W2I)
      hipLaunchKernelGGL(weight_prep_fprop_kernel<SplitW>, dim3(grid), dim3(256), 0, s, w, scale, w_fprop, (long long)(taps * cin), total);
    else if (dtype == VLFB_MIXH || dtype == VLFB_MIXH_W2 || dtype == VLFB_MIXH_W2I)
      hipLaunchKernelGGL(weight_prep_fprop_kernel<MixHW>, dim3(grid), dim3(256), 0, s, w, scale, w_fprop, (long long)(taps * cin), total);
    else
      VLFB_WITH_T16(dtype, hipLaunchKernelGGL(weight_prep_fprop_kernel<T16>, dim3(grid), dim3(256), 0, s, w, scale, w_fprop, (long long)(taps * cin), total));
  }
  if (w_dgrad) {
    dim3 grid((unsigned)((cin + 31) / 32), (unsigned)((cout + 31) / 32), (unsigned)taps);
    if (dtype == VLFB_F32)
      hipLaunchKernelGGL(weight_prep_dgrad_kernel<float>, grid, dim3(32, 8), 0, s, w, scale, w_dgrad, (int)cout, (int)taps, (int)cin);
    else if (dtype == VLFB_SPLIT)
      hipLaunchKernelGGL(weight_prep_dgrad_kernel<SplitW>, grid, dim3(32, 8), 0, s, w, scale, w_dgrad, (int)cout, (int)taps, (int)cin);
    else if (dtype == VLFB_MIX || dtype == VLFB_MIXH)
      hipLaunchKernelGGL(weight_prep_dgrad_kernel<MixW>, grid, dim3(32, 8), 0, s, w, scale, w_dgrad, (int)cout, (int)taps, (int)cin);
    else if (dtype == VLFB_MIX_W2 || dtype == VLFB_MIXH_W2)
      hipLaunchKernelGGL(weight_prep_dgrad_kernel<MixW2>, grid, dim3(32, 8), 0, s, w, scale, w_dgrad, (int)cout, (int)taps, (int)cin);
    else if (dtype == VLFB_MIX_W2I || dtype == VLFB_MIXH_W2I)
      hipLaunchKernelGGL(weight_prep_dgrad_kernel<MixW2I>, grid, dim3(32, 8), 0, s, w, scale, w_dgrad, (int)cout, (int)taps, (int)cin);
    else
      VLFB_WITH_T16(dtype, hipLaunchKernelGGL(weight_prep_dgrad_kernel<T16>, grid, dim3(32, 8), 0, s, w, scale, w_dgrad, (int)cout, (int)taps, (int)cin));
  }
  return check_launch("weight_prep");
}

extern "C" int vlfb_pool_argmax_bytes(const vlfb_pool_desc* d) {
  return d->kt * d->kh * d->kw <= 255 ? 1 : 2;
}
extern "C" int vlfb_maxpool_fwd(const vlfb_pool_desc* d, const void* x, void* y, void* argmax,
                                vlfb_stream_t stream) {
  int rc = check_pool(d);
  if (rc) return rc;
  VLFB_REQUIRE(x && y, "maxpool_fwd: null pointer");
  PoolP p = to_poolp(d);
  const int v = d->dtype == VLFB_F32 ? 4 : 8;
  const bool wide = vlfb_pool_argmax_bytes(d) == 2;
  int grid = grid_for((long long)p.N * p.To * p.Ho * p.Wo * (p.C / v), 256);
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == VLFB_F16PAIR) {
    if (!wide) hipLaunchKernelGGL((maxpool_fwd_pair_kernel<uint8_t>), dim3(grid), dim3(256), 0, s, (const f16_t*)x, (f16_t*)y, (uint8_t*)argmax, p);
    else hipLaunchKernelGGL((maxpool_fwd_pair_kernel<uint16_t>), dim3(grid), dim3(256), 0, s, (const f16_t*)x, (f16_t*)y, (uint16_t*)argmax, p);
    return check_launch("maxpool_fwd (fp16 planes)");
  }
  if (d->dtype == VLFB_F32 && !wide)
    hipLaunchKernelGGL((maxpool_fwd_kernel<float, uint8_t>), dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, (uint8_t*)argmax, p);
  else if (d->dtype == VLFB_F32)
    hipLaunchKernelGGL((maxpool_fwd_kernel<float, uint16_t>), dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, (uint16_t*)argmax, p);
  else if (!wide)
    VLFB_WITH_T16(d->dtype, hipLaunchKernelGGL((maxpool_fwd_kernel<T16, uint8_t>), dim3(grid), dim3(256), 0, s, (const T16*)x, (T16*)y, (uint8_t*)argmax, p));
  else
    VLFB_WITH_T16(d->dtype, hipLaunchKernelGGL((maxpool_fwd_kernel<T16, uint16_t>), dim3(grid), dim3(256), 0, s, (const T16*)x, (T16*)y, (uint16_t*)argmax, p));
  return check_launch("maxpool_fwd");
}
extern "C" int vlfb_maxpool_bwd(const vlfb_pool_desc* d, const void* dy, const void* argmax,
                                void* dx, const void* add, const void* mask, vlfb_stream_t stream) {
  int rc = check_pool(d);
  if (rc) return rc;
  VLFB_REQUIRE(dy && argmax && dx, "maxpool_bwd: null pointer");
  PoolP p = to_poolp(d);
  const int v = d->dtype == VLFB_F32 ? 4 : 8;
  const bool wide = vlfb_pool_argmax_bytes(d) == 2;
  int grid = grid_for((long long)p.N * p.Ti * p.Hi * p.Wi * (p.C / v), 256);
  hipStream_t s = (hipStream_t)stream;
  if (!wide) {
    bool done;
    if (d->dtype == VLFB_F32) done = launch_maxpool_bwd_fixed<float>(p, dy, argmax, dx, add, mask, nullptr, s);
    else VLFB_WITH_T16(d->dtype, done = launch_maxpool_bwd_fixed<T16>(p, dy, argmax, dx, add, mask, nullptr, s));
    if (done) return check_launch("maxpool_bwd (fixed window)");
  }
  if (d->dtype == VLFB_F32 && !wide)
    hipLaunchKernelGGL((pool_bwd_kernel<float, true, uint8_t>), dim3(grid), dim3(256), 0, s, (const float*)dy, (const uint8_t*)argmax, (float*)dx, (const float*)add, (const float*)mask, p, 1.f);
  else if (d->dtype == VLFB_F32)
    hipLaunchKernelGGL((pool_bwd_kernel<float, true, uint16_t>), dim3(grid), dim3(256), 0, s, (const float*)dy, (const uint16_t*)argmax, (float*)dx, (const float*)add, (const float*)mask, p, 1.f);
  else if (!wide)
    VLFB_WITH_T16(d->dtype, hipLaunchKernelGGL((pool_bwd_kernel<T16, true, uint8_t>), dim3(grid), dim3(256), 0, s, (const T16*)dy, (const uint8_t*)argmax, (T16*)dx, (const T16*)add, (const T16*)mask, p, 1.f));
  else
    VLFB_WITH_T16(d->dtype, hipLaunchKernelGGL((pool_bwd_kernel<T16, true, uint16_t>), dim3(grid), dim3(256), 0, s, (const T16*)dy, (const uint16_t*)argmax, (T16*)dx, (const T16*)add, (const T16*)mask, p, 1.f));
  return check_launch("maxpool_bwd");
}
extern "C" int vlfb_maxpool_relu_bwd(const vlfb_pool_desc* d, const void* dy, const void* argmax, const void* y,
                                     void* dx, vlfb_stream_t stream) {
  int rc = check_pool(d);
  if (rc) return rc;
  VLFB_REQUIRE(dy && argmax && dx && y, "maxpool_relu_bwd: null pointer");
  PoolP p = to_poolp(d);
  const int v = d->dtype == VLFB_F32 ? 4 : 8;
  const bool wide = vlfb_pool_argmax_bytes(d) == 2;
  int grid = grid_for((long long)p.N * p.Ti * p.Hi * p.Wi * (p.C / v), 256);
  hipStream_t s = (hipStream_t)stream;
  if (!wide) {
    bool done;
    if (d->dtype == VLFB_F32) done = launch_maxpool_bwd_fixed<float>(p, dy, argmax, dx, nullptr, nullptr, y, s);
    else VLFB_WITH_T16(d->dtype, done = launch_maxpool_bwd_fixed<T16>(p, dy, argmax, dx, nullptr, nullptr, y, s));
    if (done) return check_launch("maxpool_relu_bwd (fixed window)");
  }
  if (d->dtype == VLFB_F32 && !wide)
    hipLaunchKernelGGL((pool_bwd_kernel<float, true, uint8_t>), dim3(grid), dim3(256), 0, s, (const float*)dy, (const uint8_t*)argmax, (float*)dx, (const float*)nullptr, (const float*)nullptr, p, 1.f, (const float*)y);
  else if (d->dtype == VLFB_F32)
    hipLaunchKernelGGL((pool_bwd_kernel<float, true, uint16_t>), dim3(grid), dim3(256), 0, s, (const float*)dy, (const uint16_t*)argmax, (float*)dx, (const float*)nullptr, (const float*)nullptr, p, 1.f, (const float*)y);
  else if (!wide)
    VLFB_WITH_T16(d->dtype, hipLaunchKernelGGL((pool_bwd_kernel<T16, true, uint8_t>), dim3(grid), dim3(256), 0, s, (const T16*)dy, (const uint8_t*)argmax, (T16*)dx, (const T16*)nullptr, (const T16*)nullptr, p, 1.f, (const T16*)y));
  else
    VLFB_WITH_T16(d->dtype, hipLaunchKernelGGL((pool_bwd_kernel<T16, true, uint16_t>), dim3(grid), dim3(256), 0, s, (const T16*)dy, (const uint16_t*)argmax, (T16*)dx, (const T16*)nullptr, (const T16*)nullptr, p, 1.f, (const T16*)y));
  return check_launch("maxpool_relu_bwd");
}
extern "C" int vlfb_avgpool_fwd(const vlfb_pool_desc* d, const void* x, void* y, vlfb_stream_t stream) {
  int rc = check_pool(d);
  if (rc) return rc;
  VLFB_REQUIRE(x && y, "avgpool_fwd: null pointer");
  VLFB_REQUIRE(d->pt == 0 && d->ph == 0 && d->pw == 0, "avgpool: padding is not supported (none in the reference)");
  PoolP p = to_poolp(d);
  const int v = d->dtype == VLFB_F32 ? 4 : 8;
  hipStream_t s = (hipStream_t)stream;
  const bool global = p.To == 1 && p.Ho == 1 && p.Wo == 1 && p.kt == p.Ti && p.kh == p.Hi && p.kw == p.Wi;
  if (d->dtype == VLFB_F16PAIR) {
    // x: two fp16 planes, y: fp32
    const long long xplane = (long long)p.N * p.Ti * p.Hi * p.Wi * p.C;
    if (global && p.N < 65536)
      hipLaunchKernelGGL(global_avgpool_pair_kernel, dim3((unsigned)((p.C / 8 + 7) / 8), (unsigned)p.N), dim3(256), 0, s, (const f16_t*)x, (float*)y,
                         (long long)p.Ti * p.Hi * p.Wi, p.C, xplane);
    else
      hipLaunchKernelGGL(avgpool_fwd_pair_kernel, dim3(grid_for((long long)p.N * p.To * p.Ho * p.Wo * (p.C / 8), 256)), dim3(256), 0, s, (const f16_t*)x, (float*)y, p);
    return check_launch("avgpool_fwd (fp16 planes)");
  }
  if (global && p.N < 65536) {
    dim3 grid((unsigned)((p.C / v + 7) / 8), (unsigned)p.N);
    const long long rows = (long long)p.Ti * p.Hi * p.Wi;
    if (d->dtype == VLFB_F32)
      hipLaunchKernelGGL(global_avgpool_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (float*)y, rows, p.C);
    else
      VLFB_WITH_T16(d->dtype, hipLaunchKernelGGL(global_avgpool_kernel<T16>, grid, dim3(256), 0, s, (const T16*)x, (T16*)y, rows, p.C));
  } else {
    int grid = grid_for((long long)p.N * p.To * p.Ho * p.Wo * (p.C / v), 256);
    if (d->dtype == VLFB_F32)
      hipLaunchKernelGGL(avgpool_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, p);
    else
      VLFB_WITH_T16(d->dtype, hipLaunchKernelGGL(avgpool_fwd_kernel<T16>, dim3(grid), dim3(256), 0, s, (const T16*)x, (T16*)y, p));
  }
  return check_launch("avgpool_fwd");
}
extern "C" int vlfb_avgpool_bwd(const vlfb_pool_desc* d, const void* dy, void* dx, const void* add,
                                const void* mask, vlfb_stream_t stream) {
  int rc = check_pool(d);
  if (rc) return rc;
  VLFB_REQUIRE(dy && dx, "avgpool_bwd: null pointer");
  VLFB_REQUIRE(d->pt == 0 && d->ph == 0 && d->pw == 0, "avgpool: padding is not supported");
  PoolP p = to_poolp(d);
  const int v = d->dtype == VLFB_F32 ? 4 : 8;
  const float inv = 1.0f / (float)(p.kt * p.kh * p.kw);
  int grid = grid_for((long long)p.N * p.Ti * p.Hi * p.Wi * (p.C / v), 256);
  if (d->dtype == VLFB_F32)
    hipLaunchKernelGGL((pool_bwd_kernel<float, false, uint8_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)dy, (const uint8_t*)nullptr, (float*)dx, (const float*)add, (const float*)mask, p, inv);
  else
    VLFB_WITH_T16(d->dtype, hipLaunchKernelGGL((pool_bwd_kernel<T16, false, uint8_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T16*)dy, (const uint8_t*)nullptr, (T16*)dx, (const T16*)add, (const T16*)mask, p, inv));
  return check_launch("avgpool_bwd");
}

// Average-pool backward from an fp32 pooled gradient into a TWO-TERM 16-bit input gradient: dx = mask > 0 ? sum dy / window : 0,
// hi = T(dx), lo = T(dx - hi).  Where the "mix" path's fp32 head gradient re-enters the 16-bit backward (the temporal / global
// average pool over res5): every position of a channel receives the SAME value, so a one-term rounding is an error common to all
// positions -- and to every backbone gradient behind it (it was their median error).
template <typename T>
__global__ void avgpool_bwd_two_term_kernel(const float* __restrict__ dy, T* __restrict__ dx_hi, T* __restrict__ dx_lo,
                                            const T* __restrict__ mask, PoolP p, float inv_window) {
  constexpr int V = 8;
  const int cchunks = p.C / V;
  const long long total = (long long)p.N * p.Ti * p.Hi * p.Wi * cchunks;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cchunks);
    long long q = i / cchunks;
    const int wi = (int)(q % p.Wi); long long r = q / p.Wi;
    const int hi = (int)(r % p.Hi); r /= p.Hi;
    const int ti = (int)(r % p.Ti);
    const int n = (int)(r / p.Ti);
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    const int t_lo = max(0, (ti - p.kt + p.st) / p.st), t_hi = min(p.To - 1, ti / p.st);
    const int h_lo = max(0, (hi - p.kh + p.sh) / p.sh), h_hi = min(p.Ho - 1, hi / p.sh);
    const int w_lo = max(0, (wi - p.kw + p.sw) / p.sw), w_hi = min(p.Wo - 1, wi / p.sw);
    for (int to = t_lo; to <= t_hi; ++to) {
      if (ti - to * p.st >= p.kt) continue;
      for (int ho = h_lo; ho <= h_hi; ++ho) {
        if (hi - ho * p.sh >= p.kh) continue;
        for (int wo = w_lo; wo <= w_hi; ++wo) {
          if (wi - wo * p.sw >= p.kw) continue;
          const float* g = dy + ((((long long)n * p.To + to) * p.Ho + ho) * p.Wo + wo) * p.C + cc * V;
          const float4 g0 = *reinterpret_cast<const float4*>(g), g1 = *reinterpret_cast<const float4*>(g + 4);
          acc[0] += g0.x * inv_window; acc[1] += g0.y * inv_window; acc[2] += g0.z * inv_window; acc[3] += g0.w * inv_window;
          acc[4] += g1.x * inv_window; acc[5] += g1.y * inv_window; acc[6] += g1.z * inv_window; acc[7] += g1.w * inv_window;
        }
      }
    }
    const long long off = q * p.C + cc * V;
    if (mask) {
      float mk[V];
      Vec16<T>::load(mask + off, mk);
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] = mk[k] > 0.f ? acc[k] : 0.f;
    }
    Vec16<T>::store(dx_hi + off, acc);
    float h[V];
    Vec16<T>::load(dx_hi + off, h);                  // (the rounded values, as stored)
#pragma unroll
    for (int k = 0; k < V; ++k) h[k] = acc[k] - h[k];
    Vec16<T>::store(dx_lo + off, h);
  }
}

extern "C" int vlfb_avgpool_bwd_two_term(const vlfb_pool_desc* d, const float* dy, void* dx_hi, void* dx_lo, const void* mask,
                                         vlfb_stream_t stream) {
  int rc = check_pool(d);
  if (rc) return rc;
  VLFB_REQUIRE(dy && dx_hi && dx_lo, "avgpool_bwd_two_term: null pointer");
  VLFB_REQUIRE(d->pt == 0 && d->ph == 0 && d->pw == 0, "avgpool: padding is not supported");
  VLFB_REQUIRE(d->dtype == VLFB_F16 || d->dtype == VLFB_BF16, "avgpool_bwd_two_term: the input gradient is 16-bit (hi + lo)");
  PoolP p = to_poolp(d);
  VLFB_REQUIRE(p.C % 8 == 0, "avgpool_bwd_two_term: channels must be a multiple of 8");
  const float inv = 1.0f / (float)(p.kt * p.kh * p.kw);
  int grid = grid_for((long long)p.N * p.Ti * p.Hi * p.Wi * (p.C / 8), 256);
  VLFB_WITH_T16(d->dtype, hipLaunchKernelGGL((avgpool_bwd_two_term_kernel<T16>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                             dy, (T16*)dx_hi, (T16*)dx_lo, (const T16*)mask, p, inv));
  return check_launch("avgpool_bwd_two_term");
}

extern "C" int vlfb_softmax_fwd(const float* s, void* p, int dtype, int64_t rows, int64_t cols,
                                float scale, vlfb_stream_t stream) {
  VLFB_REQUIRE(s && p && rows > 0 && cols > 0 && cols < (1ll << 31), "softmax_fwd: bad args");
  int grid = grid_for(rows * 64, 256);
  if (cols % 4 == 0 && cols <= 64 * 4 * 8 && (dtype == VLFB_F32 || is16(dtype))) {
    hipStream_t st = (hipStream_t)stream;
    const bool small = cols <= 64 * 4 * 4;
    if (dtype == VLFB_F32) {
      if (small) hipLaunchKernelGGL((softmax_fwd_rowreg_kernel<float, 4>), dim3(grid), dim3(256), 0, st, s, (float*)p, (long long)rows, (int)cols, scale);
      else hipLaunchKernelGGL((softmax_fwd_rowreg_kernel<float, 8>), dim3(grid), dim3(256), 0, st, s, (float*)p, (long long)rows, (int)cols, scale);
    } else {
      if (small) VLFB_WITH_T16(dtype, hipLaunchKernelGGL((softmax_fwd_rowreg_kernel<T16, 4>), dim3(grid), dim3(256), 0, st, s, (T16*)p, (long long)rows, (int)cols, scale));
      else VLFB_WITH_T16(dtype, hipLaunchKernelGGL((softmax_fwd_rowreg_kernel<T16, 8>), dim3(grid), dim3(256), 0, st, s, (T16*)p, (long long)rows, (int)cols, scale));
    }
    return check_launch("softmax_fwd");
  }
  if (dtype == VLFB_F32)
    hipLaunchKernelGGL(softmax_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, s, (float*)p, (long long)rows, (int)cols, scale);
  else if (is16(dtype))
    VLFB_WITH_T16(dtype, hipLaunchKernelGGL(softmax_fwd_kernel<T16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, s, (T16*)p, (long long)rows, (int)cols, scale));
  else return set_error(VLFB_ERR_ARG, "softmax_fwd: bad dtype");
  return check_launch("softmax_fwd");
}
// probabilities in fp32, ds in a 16-bit type (the "mix" path: the softmax Jacobian cancels the common part of dP, so P
// is read at full precision and only the result is rounded)
extern "C" int vlfb_softmax_bwd_p32(const float* dp, const float* p, void* ds, int ds_dtype, int64_t rows,
                                    int64_t cols, float scale, vlfb_stream_t stream) {
  VLFB_REQUIRE(dp && p && ds && rows > 0 && cols > 0, "softmax_bwd_p32: bad args");
  VLFB_REQUIRE(is16(ds_dtype) && cols % 4 == 0 && cols <= 64 * 4 * 8, "softmax_bwd_p32: 16-bit ds and cols %% 4 == 0, cols <= 2048");
  const int grid = grid_for(rows * 64, 256);
  hipStream_t st = (hipStream_t)stream;
  if (cols <= 64 * 4 * 4)
    VLFB_WITH_T16(ds_dtype, hipLaunchKernelGGL((softmax_bwd_rowreg_kernel<float, 4, T16>), dim3(grid), dim3(256), 0, st, dp, p, (T16*)ds, (long long)rows, (int)cols, scale));
  else
    VLFB_WITH_T16(ds_dtype, hipLaunchKernelGGL((softmax_bwd_rowreg_kernel<float, 8, T16>), dim3(grid), dim3(256), 0, st, dp, p, (T16*)ds, (long long)rows, (int)cols, scale));
  return check_launch("softmax_bwd_p32");
}
extern "C" int vlfb_softmax_bwd(const float* dp, const void* p, void* ds, int dtype, int64_t rows,
                                int64_t cols, float scale, vlfb_stream_t stream) {
  VLFB_REQUIRE(dp && p && ds && rows > 0 && cols > 0 && cols < (1ll << 31), "softmax_bwd: bad args");
  int grid = grid_for(rows * 64, 256);
  if (cols % 4 == 0 && cols <= 64 * 4 * 8 && (dtype == VLFB_F32 || is16(dtype))) {
    hipStream_t st = (hipStream_t)stream;
    const bool small = cols <= 64 * 4 * 4;
    if (dtype == VLFB_F32) {
      if (small) hipLaunchKernelGGL((softmax_bwd_rowreg_kernel<float, 4>), dim3(grid), dim3(256), 0, st, dp, (const float*)p, (float*)ds, (long long)rows, (int)cols, scale);
      else hipLaunchKernelGGL((softmax_bwd_rowreg_kernel<float, 8>), dim3(grid), dim3(256), 0, st, dp, (const float*)p, (float*)ds, (long long)rows, (int)cols, scale);
    } else {
      if (small) VLFB_WITH_T16(dtype, hipLaunchKernelGGL((softmax_bwd_rowreg_kernel<T16, 4>), dim3(grid), dim3(256), 0, st, dp, (const T16*)p, (T16*)ds, (long long)rows, (int)cols, scale));
      else VLFB_WITH_T16(dtype, hipLaunchKernelGGL((softmax_bwd_rowreg_kernel<T16, 8>), dim3(grid), dim3(256), 0, st, dp, (const T16*)p, (T16*)ds, (long long)rows, (int)cols, scale));
    }
    return check_launch("softmax_bwd");
  }
  if (dtype == VLFB_F32)
    hipLaunchKernelGGL(softmax_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, dp, (const float*)p, (float*)ds, (long long)rows, (int)cols, scale);
  else if (is16(dtype))
    VLFB_WITH_T16(dtype, hipLaunchKernelGGL(softmax_bwd_kernel<T16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, dp, (const T16*)p, (T16*)ds, (long long)rows, (int)cols, scale));
  else return set_error(VLFB_ERR_ARG, "softmax_bwd: bad dtype");
  return check_launch("softmax_bwd");
}

extern "C" int vlfb_add(const void* a, const void* b, void* y, const void* mask, int dtype,
                        int64_t n, int relu, vlfb_stream_t stream) {
  VLFB_REQUIRE(a && y && n >= 0, "add: bad args");
  if (n == 0) return VLFB_OK;
  const int v = dtype == VLFB_F32 ? 4 : 8;
  int grid = grid_for((n + v - 1) / v, 256);
  if (dtype == VLFB_F32)
    hipLaunchKernelGGL(add_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)a, (const float*)b, (float*)y, (const float*)mask, (long long)n, relu);
  else if (is16(dtype))
    VLFB_WITH_T16(dtype, hipLaunchKernelGGL(add_kernel<T16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T16*)a, (const T16*)b, (T16*)y, (const T16*)mask, (long long)n, relu));
  else return set_error(VLFB_ERR_ARG, "add: bad dtype");
  return check_launch("add");
}
extern "C" int vlfb_relu_fwd(const void* x, void* y, int dtype, int64_t n, vlfb_stream_t stream) {
  return vlfb_add(x, nullptr, y, nullptr, dtype, n, 1, stream);
}
extern "C" int vlfb_relu_bwd(const void* dy, const void* yv, void* dx, int dtype, int64_t n,
                             vlfb_stream_t stream) {
  VLFB_REQUIRE(yv != nullptr, "relu_bwd: y is required");
  return vlfb_add(dy, nullptr, dx, yv, dtype, n, 0, stream);
}
namespace vlfb {
// row slabs of a column-sum launch over (rows x cols) elements of `dtype` (the grid's y extent)
int colsum_slabs(int dtype, long long rows, long long cols) {
  const int v = dtype == VLFB_F32 ? 4 : 8;
  const int cblocks = (int)((cols / v + 7) / 8);
  int slabs = (int)((rows + 1023) / 1024);
  const int want = (2048 + cblocks - 1) / cblocks;
  if (slabs > want) slabs = want;
  return slabs < 1 ? 1 : slabs;
}
// per-slab column sums into partials[colsum_slabs][cols] (no atomics: the caller folds the slabs in a fixed order)
int colsum_partials(const void* g, int dtype, long long rows, long long cols, long long ld, float* partials, hipStream_t s) {
  const int v = dtype == VLFB_F32 ? 4 : 8;
  VLFB_REQUIRE(g && partials && rows > 0 && cols > 0 && ld >= cols && cols % v == 0 && ld % v == 0 && dtype_ok(dtype), "colsum_partials: bad args");
  dim3 grid((unsigned)((cols / v + 7) / 8), (unsigned)colsum_slabs(dtype, rows, cols));
  if (dtype == VLFB_F32)
    hipLaunchKernelGGL((colsum_kernel<float, true>), grid, dim3(256), 0, s, (const float*)g, rows, (int)cols, ld, partials);
  else
    VLFB_WITH_T16(dtype, hipLaunchKernelGGL((colsum_kernel<T16, true>), grid, dim3(256), 0, s, (const T16*)g, rows, (int)cols, ld, partials));
  return check_launch("colsum (partials)");
}
}  // namespace vlfb

extern "C" int vlfb_colsum(const void* g, int dtype, int64_t rows, int64_t cols, int64_t ld,
                           float* out, int accumulate, vlfb_stream_t stream) {
  VLFB_REQUIRE(g && out && rows > 0 && cols > 0 && ld >= cols, "colsum: bad args");
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate)
    hipLaunchKernelGGL(zero_kernel, dim3(grid_for(cols, 256)), dim3(256), 0, s, out, (long long)cols);
  const int v = dtype == VLFB_F32 ? 4 : 8;
  VLFB_REQUIRE(cols % v == 0 && ld % v == 0, "colsum: cols and ld must be multiples of %d", v);
  const int cblocks = (int)((cols / v + 7) / 8);
  int slabs = (int)((rows + 1023) / 1024);
  const int want = (2048 + cblocks - 1) / cblocks;
  if (slabs > want) slabs = want;
  if (slabs < 1) slabs = 1;
  dim3 grid((unsigned)cblocks, (unsigned)slabs);
  if (dtype == VLFB_F32)
    hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)g, (long long)rows, (int)cols, (long long)ld, out);
  else if (is16(dtype))
    VLFB_WITH_T16(dtype, hipLaunchKernelGGL(colsum_kernel<T16>, grid, dim3(256), 0, s, (const T16*)g, (long long)rows, (int)cols, (long long)ld, out));
  else return set_error(VLFB_ERR_ARG, "colsum: bad dtype");
  return check_launch("colsum");
}

extern "C" int vlfb_copy2d(const void* src, int64_t lds, void* dst, int64_t ldd, int dtype,
                           int64_t rows, int64_t cols, vlfb_stream_t stream) {
  VLFB_REQUIRE(src && dst && rows > 0 && cols > 0 && lds >= cols && ldd >= cols, "copy2d: bad args");
  int grid = grid_for(rows * cols, 256);
  if (dtype == VLFB_F32)
    hipLaunchKernelGGL(copy2d_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)src, (long long)lds, (float*)dst, (long long)ldd, (long long)rows, (long long)cols);
  else if (is16(dtype))
    VLFB_WITH_T16(dtype, hipLaunchKernelGGL(copy2d_kernel<T16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T16*)src, (long long)lds, (T16*)dst, (long long)ldd, (long long)rows, (long long)cols));
  else return set_error(VLFB_ERR_ARG, "copy2d: bad dtype");
  return check_launch("copy2d");
}
extern "C" int vlfb_zero_f32(float* p, int64_t n, vlfb_stream_t stream) {
  VLFB_REQUIRE(p && n >= 0, "zero_f32: bad args");
  if (n == 0) return VLFB_OK;
  hipLaunchKernelGGL(zero_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, p, (long long)n);
  return check_launch("zero_f32");
}

extern "C" int vlfb_weight_prep_batched(const vlfb_wprep_item* items_dev, int n_items, int total_tiles,
                                        int dtype, vlfb_stream_t stream) {
  VLFB_REQUIRE(items_dev && n_items > 0 && total_tiles > 0, "weight_prep_batched: bad args");
  dim3 block(32, 8);
  if (dtype == VLFB_F32)
    hipLaunchKernelGGL(weight_prep_batched_kernel<float>, dim3((unsigned)total_tiles), block, 0, (hipStream_t)stream, items_dev, n_items);
  else if (is16(dtype))
    VLFB_WITH_T16(dtype, hipLaunchKernelGGL(weight_prep_batched_kernel<T16>, dim3((unsigned)total_tiles), block, 0, (hipStream_t)stream, items_dev, n_items));
  else if (dtype == VLFB_SPLIT)
    hipLaunchKernelGGL(weight_prep_batched_kernel<SplitW>, dim3((unsigned)total_tiles), block, 0, (hipStream_t)stream, items_dev, n_items);
  else if (dtype == VLFB_MIX)
    hipLaunchKernelGGL(weight_prep_batched_kernel<MixW>, dim3((unsigned)total_tiles), block, 0, (hipStream_t)stream, items_dev, n_items);
  else if (dtype == VLFB_MIX_W2)
    hipLaunchKernelGGL(weight_prep_batched_kernel<MixW2>, dim3((unsigned)total_tiles), block, 0, (hipStream_t)stream, items_dev, n_items);
  else if (dtype == VLFB_MIXH)
    hipLaunchKernelGGL(weight_prep_batched_kernel<MixHW>, dim3((unsigned)total_tiles), block, 0, (hipStream_t)stream, items_dev, n_items);
  else if (dtype == VLFB_MIXH_W2)
    hipLaunchKernelGGL(weight_prep_batched_kernel<MixHW2>, dim3((unsigned)total_tiles), block, 0, (hipStream_t)stream, items_dev, n_items);
  else if (dtype == VLFB_MIX_W2I)      // (the caller keeps Cout % 64 == 0 for these items: vlfb_weight_prep checks it per conv)
    hipLaunchKernelGGL(weight_prep_batched_kernel<MixW2I>, dim3((unsigned)total_tiles), block, 0, (hipStream_t)stream, items_dev, n_items);
  else if (dtype == VLFB_MIXH_W2I)
    hipLaunchKernelGGL(weight_prep_batched_kernel<MixHW2I>, dim3((unsigned)total_tiles), block, 0, (hipStream_t)stream, items_dev, n_items);
  else return set_error(VLFB_ERR_ARG, "weight_prep_batched: bad dtype");
  return check_launch("weight_prep_batched");
}

extern "C" int vlfb_ncthw_to_nthwc_wpad(const float* src, void* dst, int dtype, int64_t n, int64_t c,
                                        int64_t rows, int64_t w, int64_t c_pad, int64_t wpad_left,
                                        int64_t w_total, vlfb_stream_t stream) {
  VLFB_REQUIRE(src && dst && c_pad >= c && n > 0 && c > 0 && rows > 0 && w > 0 && wpad_left >= 0 &&
               w_total >= wpad_left + w, "ncthw_to_nthwc_wpad: bad args");
  int grid = grid_for(n * rows * w_total, 256);
  if (dtype == VLFB_F32)
    hipLaunchKernelGGL(ncthw_to_nthwc_wpad_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (float*)dst, (long long)n, (int)c, (long long)rows, (int)w, (int)c_pad, (int)wpad_left, (int)w_total);
  else if (is16(dtype))
    VLFB_WITH_T16(dtype, hipLaunchKernelGGL(ncthw_to_nthwc_wpad_kernel<T16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (T16*)dst, (long long)n, (int)c, (long long)rows, (int)w, (int)c_pad, (int)wpad_left, (int)w_total));
  else return set_error(VLFB_ERR_ARG, "ncthw_to_nthwc_wpad: bad dtype");
  return check_launch("ncthw_to_nthwc_wpad");
}
