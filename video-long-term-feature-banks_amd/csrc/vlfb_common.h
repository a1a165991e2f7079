// Shared device/host helpers for libvlfb_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "vlfb.h"

namespace vlfb {

typedef unsigned short bf16_t;  // raw bf16 storage

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
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                          // round-nearest-even
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int EPC = 4;  // elements per 16-byte chunk
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int EPC = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
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
template <> struct Vec16<bf16_t> {
  static constexpr int N = 8;
  __device__ static __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    uint4 t = *reinterpret_cast<const uint4*>(p);
    uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ static __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    uint4 t;
    t.x = pack_bf2(v[0], v[1]); t.y = pack_bf2(v[2], v[3]);
    t.z = pack_bf2(v[4], v[5]); t.w = pack_bf2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = t;
  }
};

inline int grid_for(int64_t work_items, int block, int cap = 256 * 16) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

}  // namespace vlfb
