// Shared device/host helpers for libvlfb_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "vlfb.h"

namespace vlfb {

typedef unsigned short bf16_t;  // raw bf16 storage
struct f16_t { unsigned short v; };   // raw IEEE fp16 storage (a distinct type so that kernels can be instantiated for it)

inline bool is16(int dtype) { return dtype == VLFB_BF16 || dtype == VLFB_F16; }
inline bool dtype_ok(int dtype) { return dtype == VLFB_F32 || is16(dtype); }
// run the statement(s) with T16 = the 16-bit element type `dtype` names
#define VLFB_WITH_T16(dtype, ...)                                               \
  do {                                                                          \
    if ((dtype) == VLFB_F16) { typedef ::vlfb::f16_t T16; __VA_ARGS__; }        \
    else { typedef ::vlfb::bf16_t T16; __VA_ARGS__; }                           \
  } while (0)

// ---- error plumbing -----------------------------------------------------------------------
int set_error(int code, const char* fmt, ...);
inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(VLFB_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return VLFB_OK;
}
#define VLFB_REQUIRE(cond, ...) \
  do { if (!(cond)) return ::vlfb::set_error(VLFB_ERR_ARG, __VA_ARGS__); } while (0)

// ---- scalar conversions -------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// fp32 -> bf16, round-nearest-even, by the gfx950 converter (v_cvt_pk_bf16_f32: one instruction per PAIR; NaNs come
// out quiet).  Finite values and infinities round exactly like the integer recipe u += 0x7fff + ((u >> 16) & 1).
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw;
typedef __attribute__((ext_vector_type(2))) float f32x2_hw;
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  const f32x2_hw v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float h2f(unsigned short v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ unsigned short f2h(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }   // round-nearest-even

// fp32 -> fp16 for the COPIES the "mix" engine's fp16 backward reads as ReLU masks: a positive value below the fp16
// subnormal range must stay positive (sign(copy) == sign(value) is what `mask > 0` tests), so it becomes the smallest
// subnormal instead of +0 (an absolute change of < 6e-8)
__device__ __forceinline__ unsigned short f2h_pos(float f) {
  const unsigned short h = f2h(f);
  return (f > 0.f && (h & 0x7fffu) == 0) ? (unsigned short)1 : h;
}
__device__ __forceinline__ uint32_t pack_h2_pos(float lo, float hi) { return (uint32_t)f2h_pos(lo) | ((uint32_t)f2h_pos(hi) << 16); }

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int EPC = 4;  // elements per 16-byte chunk
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
  // (the packed-pair helpers only exist so that `sizeof(T) == 4 ? ... : ...` code compiles for float)
  __device__ static __forceinline__ uint32_t pack2(float, float) { return 0u; }
  __device__ static __forceinline__ float lo(uint32_t) { return 0.f; }
  __device__ static __forceinline__ float hi(uint32_t) { return 0.f; }
};
template <> struct Elem<bf16_t> {
  static constexpr int EPC = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
  // two elements packed in a 32-bit word (low half first)
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) { return pack_bf2(lo, hi); }
  __device__ static __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
  __device__ static __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
};
template <> struct Elem<f16_t> {
  static constexpr int EPC = 8;
  __device__ static __forceinline__ float ld(const f16_t* p) { return h2f(p->v); }
  __device__ static __forceinline__ void st(f16_t* p, float v) { p->v = f2h(v); }
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) { return (uint32_t)f2h(lo) | ((uint32_t)f2h(hi) << 16); }
  __device__ static __forceinline__ float lo(uint32_t w) { return h2f((unsigned short)(w & 0xffffu)); }
  __device__ static __forceinline__ float hi(uint32_t w) { return h2f((unsigned short)(w >> 16)); }
};

// 16-byte vector of T viewed as floats
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  __device__ static __forceinline__ void load(const float* p, float (&v)[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ static __forceinline__ void store(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <typename T> struct Vec16x16bit {
  static constexpr int N = 8;
  __device__ static __forceinline__ void load(const T* p, float (&v)[8]) {
    uint4 t = *reinterpret_cast<const uint4*>(p);
    uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = Elem<T>::lo(w[i]);
      v[2 * i + 1] = Elem<T>::hi(w[i]);
    }
  }
  __device__ static __forceinline__ void store(T* p, const float (&v)[8]) {
    uint4 t;
    t.x = Elem<T>::pack2(v[0], v[1]); t.y = Elem<T>::pack2(v[2], v[3]);
    t.z = Elem<T>::pack2(v[4], v[5]); t.w = Elem<T>::pack2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = t;
  }
};
template <> struct Vec16<bf16_t> : Vec16x16bit<bf16_t> {};
template <> struct Vec16<f16_t> : Vec16x16bit<f16_t> {};

// vlfb_ops.hip: column sums without atomics -- per-slab partial rows that the caller folds in a fixed order
int colsum_slabs(int dtype, long long rows, long long cols);
int colsum_partials(const void* g, int dtype, long long rows, long long cols, long long ld, float* partials, hipStream_t s);

inline int grid_for(int64_t work_items, int block, int cap = 256 * 16) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

}  // namespace vlfb
