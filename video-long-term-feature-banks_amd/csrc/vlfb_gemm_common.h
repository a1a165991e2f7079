// Shared pieces of the implicit-GEMM kernels (vlfb_gemm.hip: the 128x128 NT / TN kernels and the
// planner; vlfb_gemm8.hip: the 256-row, 8-phase pipelined NT / TN kernels): launch parameters, gather
// arithmetic, the LDS swizzle, MFMA wrappers, epilogue converters and the XCD-aware workgroup remap.
#pragma once
#include "vlfb_common.h"
#include <string.h>
#include <stdlib.h>

namespace vlfb {

// launch parameters of every implicit-GEMM kernel (filled by make_plan in vlfb_gemm.hip)
struct GP {
  const char* A;
  const char* B;
  const char* P;
  char* O;
  const float* bias;
  const float* rowscale;
  const char* R;
  const char* Mask;
  float* ws;
  int M, Ncols, K;
  int Tr, Hr, Wr, Ts, Hs, Ws, Cs;
  int kt, kh, kw;
  float inv_khw, inv_kw, inv_kh;
  int st, sh, sw, pt, ph, pw, dt, dh, dw;
  int lst, lsh, lsw;  // log2 strides (DGRAD)
  int cpt_shift;      // log2(16-byte chunks per tap)
  int lda, ldb, ldo, ldr, ldp;
  long long a_bs, b_bs, o_bs, r_bs, p_bs;
  float alpha;
  int relu, bias_mode, accumulate;
  int tiles_m, tiles_n;
  int splits, kper;
  unsigned a_bytes, b_bytes;   // NT: extent of one batch element of A / B (buffer descriptors)
  int vec_epi;        // NT: LDS-staged, fully coalesced epilogue is legal for this problem
  int epi;            // NT: the fp32 tile is staged through LDS in this many passes (1 or 2)
  // NT DGRAD with stride (1, 2, 2): rows are enumerated parity class by parity class ((h & 1, w & 1), each class in
  // (n, t, h / 2, w / 2) order), so a tile holds one class and only walks the taps that class can reach
  int s2;             // 1: class-major rows
  int s2_mq;          // rows per class (M / 4)
  int s2_tpc;         // row tiles per class
  // split-bf16 math (vlfb_gemm_split.hip): element stride between the bf16 term planes of the weight operand
  long long b_ps;
  // ... of the activation / gradient operands when they arrive pre-split (vlfb_conv_desc.a_planes / p_planes), and the
  // optional term-plane copy of an NT output (o_planes planes at OP, o_ps elements apart, written by the epilogue)
  long long a_ps, p_ps, o_ps;
  char* OP;
  int op_n;
  // WGRAD (gemm_tn_tr_kernel): also the column sums of P over the positions -- the bias gradient of the conv whose weight
  // gradient this launch computes (the column tile 0 workgroups add up the P tiles they stage anyway).  dbias: fp32 [Ncols];
  // with splits > 1 the per-split partial rows go behind the weight slabs in ws and the reduce pass folds them in order.
  float* dbias;
  // NT, 16-bit outputs: TWO-TERM residual / output (the "mix" path's residual-stream gradient, which is re-rounded once
  // per bottleneck block): v = alpha * acc + bias + R + R2; relu; mask; O = T(v), O2 = T(v - T(v)).  Either may be null.
  const char* R2;
  char* O2;
  // split-bf16 NT launches (vlfb_gemm_split.hip) with a TWO-PLANE fp16 output: O / O2 and R / R2 are fp16 planes (hi, lo)
  // instead of fp32 tensors -- where an fp32 operand (the attention output of a non-local block) enters the two-plane forward
  int pair_io;
  // VLFB_MATH_F16X3: the two planes INTERLEAVED in groups of 32 k ([row][k / 32][hi 32 | lo 32]; a_ps = b_ps = 32 elements)
  int pair_il;
  // epilogue traffic -- output rows, residual and mask rows: touched once per launch -- with the non-temporal hint, so that
  // it streams through the L2 instead of evicting the operand panels the other workgroups of the XCD are re-reading
  // (set per launch in conv_run_impl, VLFB_NT_EPI; measured per launch and in the step)
  int nt_epi;
};

// vlfb_gemm8.hip: 256-row phase-pipelined NT kernel.  bm = 256 | 196 (two wave rows of 98), bn = 256 | 128; mode 0 = plain rows, 1 = gathered
// FPROP, 2 = gathered unit-stride DGRAD (taps must span whole 64-element k-tiles); dtype = VLFB_BF16 | VLFB_F16.
int launch_nt8(const GP& gp, int bm, int bn, int mode, int dtype, bool out_f32, unsigned batch, hipStream_t s);
// 256 x 256 phase-pipelined TN kernel (plain rows: wgrad of 1x1x1 convs, attention products); grid as planned
int launch_tn8(const GP& gp, dim3 grid, int dtype, bool out_f32, hipStream_t s);

// vlfb_gemm_s.hip: weight-resident streaming NT kernel (bf16 / f16 in and out, batch 1).  mode as launch_nt8;
// nts_chunk = k-tiles per load chunk the kernel would use for this K (0: K not supported)
int nts_chunk(int mode, long long K);
int launch_nts(const GP& gp, int mode, int dtype, hipStream_t s);

// vlfb_wgrad_rows.hip: whole-row WGRAD of thin 64 -> 64 channel convs (gp.tiles_m = output rows, gp.kper = rows per
// workgroup, gp.ws = slabs); wgrad_rows_ct = column tiles per wave for K gathered columns (0: K not supported)
int wgrad_rows_ct(long long K);
int launch_wgrad_rows(const GP& gp, int splits, size_t lds, int dtype, hipStream_t s);
// fat-input variant: 256 -> 64 channels, 3x1x1 (the gradient rows are shifted, the 512-byte input rows read once)
int launch_wgrad_rows_fat(const GP& gp, int splits, int dtype, hipStream_t s);

// vlfb_gemm_skinny.hip: plain-row NT products with at most 64 rows (the FBO head's 1x1x1 convs on one row per RoI)
bool skinny_nt_ok(const GP& gp, int dtype, long long batch, bool ident);
int launch_skinny_nt(const GP& gp, int dtype, bool out_f32, hipStream_t s);
bool skinny_nt_split_ok(const GP& gp, long long batch, bool ident);
int launch_skinny_nt_split(const GP& gp, hipStream_t s);

// vlfb_stem.hip: direct-convolution FPROP of the packed stem (whole output rows per wave, raw input rows in LDS)
bool stem_fprop_ok(const GP& gp, int pack_w, int dtype, int out_dtype, long long batch);
int launch_stem_fprop(const GP& gp, int dtype, hipStream_t s);
bool stem_fprop_pair_ok(const GP& gp, int pack_w, long long batch);      // (two fp16 planes in and out: VLFB_MATH_F16X3)
int launch_stem_fprop_pair(const GP& gp, hipStream_t s);

// vlfb_conv_rows.hip: direct-convolution FPROP / unit-stride DGRAD of 1x3x3 convs with 64 -> 64 channels (weights
// resident in LDS, input rows rolling through a ring); mode = VLFB_CONV_FPROP | VLFB_CONV_DGRAD
bool conv_rows64_ok(const GP& gp, int mode, int dtype, int out_dtype, long long batch);
int launch_conv_rows64(const GP& gp, int mode, int dtype, hipStream_t s);

// vlfb_gemm_split.hip: split-bf16 math on fp32 storage.  NT: npl = bf16 terms per operand (2 | 3), bn = 128 | 64, kind 0 plain
// rows, 1 gathered FPROP, 2 gathered DGRAD, 3 packed stem FPROP; ut = scalar tap cursor.  TN: 2 terms, tiles 128 | 64.
int launch_nt_split(const GP& gp, int npl, int bn, int kind, bool ut, dim3 grid, size_t lds, hipStream_t s);
int launch_tn_split(const GP& gp, int bp, int bq, bool ident, bool packw, dim3 grid, size_t lds, hipStream_t s);
// NT with the activation operand pre-split into npl bf16 term planes (kind 0 plain rows, 1 / 2 gathered FPROP / DGRAD with
// the scalar tap cursor)
int launch_nt_planes(const GP& gp, int npl, int bn, int kind, dim3 grid, size_t lds, hipStream_t s);
// vlfb_gemm_pair.hip: VLFB_MATH_F16X3 -- both operands as two fp16 planes (GP::a_ps / b_ps apart), plain rows or scalar tap cursor
int launch_nt_pair(const GP& gp, int bn, bool ident, bool pre, bool out_f32, dim3 grid, size_t lds, hipStream_t s);
int launch_nt8_pair(const GP& gp, int bm, int bn, int mode, bool out_f32, hipStream_t s);

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_v;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_v;
typedef __attribute__((ext_vector_type(4))) float f32x4_v;

constexpr int kThreads = 256;
constexpr int kRowBytes = 128;  // one LDS tile row = 128 bytes of K


struct RowC { int n, t, h, w; };

__device__ __forceinline__ RowC decode_row(const GP& p, int m) {
  RowC r;
  int hw = p.Hr * p.Wr;
  int thw = p.Tr * hw;
  r.n = m / thw;
  int rem = m - r.n * thw;
  r.t = rem / hw;
  rem -= r.t * hw;
  r.h = rem / p.Wr;
  r.w = rem - r.h * p.Wr;
  return r;
}
__device__ __forceinline__ void advance_row(const GP& p, RowC& r) {
  if (++r.w == p.Wr) {
    r.w = 0;
    if (++r.h == p.Hr) {
      r.h = 0;
      if (++r.t == p.Tr) { r.t = 0; ++r.n; }
    }
  }
}

struct TapC { int a, b, c, ci; bool ok; };

// kc = global 16-byte chunk index along K
template <typename T, bool PACKW>
__device__ __forceinline__ TapC decode_tap(const GP& p, int kc) {
  constexpr int EPC = Elem<T>::EPC;
  TapC t;
  t.ok = kc * EPC < p.K;
  int tap = kc >> p.cpt_shift;
  int within = kc & ((1 << p.cpt_shift) - 1);
  if (PACKW) {
    // taps enumerate (a, b); the packed (kw, channel) run is the per-tap K extent
    t.a = (int)(((float)tap + 0.5f) * p.inv_kh);
    t.b = tap - t.a * p.kh;
    t.c = within * (EPC / 4);  // first pixel of this chunk (Cs == 4)
    t.ci = 0;
  } else {
    t.a = (int)(((float)tap + 0.5f) * p.inv_khw);
    int rem = tap - t.a * p.kh * p.kw;
    t.b = (int)(((float)rem + 0.5f) * p.inv_kw);
    t.c = rem - t.b * p.kw;
    t.ci = within * EPC;
  }
  return t;
}

// element offset of the source chunk and whether it exists (false = padding). (non-PACKW)
template <bool DGRAD>
__device__ __forceinline__ long long src_offset(const GP& p, const RowC& r, const TapC& t, bool& ok) {
  int ts, hs, ws;
  if (!DGRAD) {
    ts = r.t * p.st - p.pt + t.a * p.dt;
    hs = r.h * p.sh - p.ph + t.b * p.dh;
    ws = r.w * p.sw - p.pw + t.c * p.dw;
    ok = (unsigned)ts < (unsigned)p.Ts && (unsigned)hs < (unsigned)p.Hs && (unsigned)ws < (unsigned)p.Ws;
  } else {
    const int nt = r.t + p.pt - t.a * p.dt;
    const int nh = r.h + p.ph - t.b * p.dh;
    const int nw = r.w + p.pw - t.c * p.dw;
    ok = (nt | nh | nw) >= 0 && ((nt & (p.st - 1)) | (nh & (p.sh - 1)) | (nw & (p.sw - 1))) == 0;
    ts = nt >> p.lst; hs = nh >> p.lsh; ws = nw >> p.lsw;
    ok = ok && ts < p.Ts && hs < p.Hs && ws < p.Ws;
  }
  return ((long long)((r.n * p.Ts + ts) * p.Hs + hs) * p.Ws + ws) * p.lda + t.ci;
}

__device__ __forceinline__ uint4 ld16(const char* base, long long byte_off) {
  return *reinterpret_cast<const uint4*>(base + byte_off);
}
__device__ __forceinline__ uint2 ld8(const char* base, long long byte_off) {
  return *reinterpret_cast<const uint2*>(base + byte_off);
}
// 16 bytes of zeros in HBM: the source of padding / out-of-range chunks of the register-staged
// (fp32 TN) gathers, so that they need no select on the data, only on the address.
__device__ uint4 g_zero16;
__device__ __forceinline__ const char* src_or_zero(const char* base, long long byte_off, bool ok) {
  return ok ? base + byte_off : reinterpret_cast<const char*>(&g_zero16);
}
// Async 16-byte global -> LDS copies (the DMA kernels) go through a buffer descriptor
// (buffer_load_dwordx4 ... offen lds; destination = wave-uniform LDS base + lane * 16): 32-bit byte offset
// per lane, hardware range check -- a lane whose offset is >= num_records writes ZEROS to its LDS
// slot (probed on MI355X, scratch/buf_probe.hip), so padding needs no zero page and no 64-bit address
// arithmetic.  kOOB is the "this chunk is padding" offset; operands are required to be < 2 GiB.
constexpr unsigned kOOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const char* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void bufglds16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16,
                                           (int)voff, (int)soff, 0, 0);
}
// The same DMA, HIDDEN from the compiler (inline asm).  hipcc orders every later LDS read whose address it cannot
// prove distinct -- in practice every ds_read_b64_tr_b16 -- behind ALL pending LDS DMAs of the builtin above by
// emitting s_waitcnt vmcnt(0) in front of it, which drains a multi-stage ring on every k-step (found in the .s of the
// transposed-read kernels; the plain ds_read_b128 loops of the NT kernels are not affected).  With the DMA in asm the
// compiler counts nothing, so completion is the kernel's job: counted `s_waitcnt vmcnt(N)` + `s_barrier` before the
// slot is read, exactly as those kernels already do.  lds_wave_base must be wave-uniform; M0 is written and consumed
// inside the statement (the compiler re-loads M0 before its own uses).
__device__ __forceinline__ unsigned lds_addr_of(const char* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void bufglds16_hidden(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
               :: "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_wave_base) : "memory");
}
__device__ __forceinline__ void bufglds16_hidden(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, const char* lds_wave_base) {
  bufglds16_hidden(rsrc, voff, soff, (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr_of(lds_wave_base)));
}

// Branch-free predicated loads: the load always executes (from offset 0 of the operand when the
// element is padding / out of range) and the result is selected afterwards, so the gather of a
// k-tile is one straight-line run of global loads instead of one basic block per element.
__device__ __forceinline__ uint4 ld16_if(const char* base, long long byte_off, bool ok) {
  return *reinterpret_cast<const uint4*>(src_or_zero(base, byte_off, ok));
}
__device__ __forceinline__ uint2 ld8_if(const char* base, long long byte_off, bool ok) {
  return *reinterpret_cast<const uint2*>(src_or_zero(base, byte_off, ok));
}

// One gathered 16-byte chunk of the activation operand (branch-free).
template <typename T, bool IDENT, bool DGRAD, bool PACKW>
__device__ __forceinline__ uint4 load_act_chunk(const GP& p, const char* base, int m, bool m_ok,
                                                const RowC& r, const TapC& t, int kc) {
  constexpr int EPC = Elem<T>::EPC;
  if (IDENT) {
    return ld16_if(base, ((long long)m * p.lda + (long long)kc * EPC) * (long long)sizeof(T), m_ok && t.ok);
  } else {
    static_assert(!PACKW, "the packed stem is gathered through buffer offsets in the kernels themselves");
    bool ok;
    const long long off = src_offset<DGRAD>(p, r, t, ok);
    return ld16_if(base, off * (long long)sizeof(T), ok && m_ok && t.ok);
  }
}

// LDS tile rows are RB bytes of K (128: 8 chunks, XOR key row & 7; 64: 4 chunks, key (row >> 2) & 3
// -- four 64-byte rows share one 256-byte bank row, so the key must change every 4 rows).
template <int RB>
__device__ __forceinline__ int swz_key(int row) { return RB == 128 ? (row & 7) : ((row >> 2) & 3); }
template <int RB = 128>
__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * RB + ((chunk ^ swz_key<RB>(row)) << 4);
}

// ---- MFMA wrappers --------------------------------------------------------------------------
template <typename T> struct Mma;
// 16-bit element types: the MFMA operand vector (8 k per lane) and the instruction
template <typename T> struct V16;
template <> struct V16<bf16_t> {
  typedef bf16x8_v V;
  __device__ static __forceinline__ f32x4_v mma(V a, V b, f32x4_v c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct V16<f16_t> {
  typedef f16x8_v V;
  __device__ static __forceinline__ f32x4_v mma(V a, V b, f32x4_v c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <typename T> struct Mma16 {
  static constexpr int KSTEPS = 2;   // per 128-byte row (a 64-byte row is one k-step)
  struct Frag { typename V16<T>::V v; };
  template <int RB = 128>
  __device__ static __forceinline__ Frag load(const char* tile, int row, int ks, int g) {
    Frag f;
    f.v = *reinterpret_cast<const typename V16<T>::V*>(tile + lds_off<RB>(row, ks * 4 + g));
    return f;
  }
  __device__ static __forceinline__ f32x4_v mma(const Frag& a, const Frag& b, f32x4_v c) { return V16<T>::mma(a.v, b.v, c); }
};
template <> struct Mma<bf16_t> : Mma16<bf16_t> {};
template <> struct Mma<f16_t> : Mma16<f16_t> {};
template <> struct Mma<float> {
  static constexpr int KSTEPS = 1;
  struct Frag { float v[8]; };
  template <int RB = 128>
  __device__ static __forceinline__ Frag load(const char* tile, int row, int /*ks*/, int g) {
    static_assert(RB == 128, "the fp32 path keeps 128-byte tile rows");
    Frag f;
    float4 lo = *reinterpret_cast<const float4*>(tile + lds_off(row, 2 * g));
    float4 hi = *reinterpret_cast<const float4*>(tile + lds_off(row, 2 * g + 1));
    f.v[0] = lo.x; f.v[1] = lo.y; f.v[2] = lo.z; f.v[3] = lo.w;
    f.v[4] = hi.x; f.v[5] = hi.y; f.v[6] = hi.z; f.v[7] = hi.w;
    return f;
  }
  __device__ static __forceinline__ f32x4_v mma(const Frag& a, const Frag& b, f32x4_v c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
    return c;
  }
};

template <typename T> __device__ __forceinline__ float ld_elem(const char* base, long long idx) {
  return Elem<T>::ld(reinterpret_cast<const T*>(base) + idx);
}

// store 4 consecutive fp32 results as OutT (vector when aligned)
template <typename OutT>
__device__ __forceinline__ void store4(char* base, long long idx, const float (&v)[4], int count, bool vec_ok) {
  OutT* o = reinterpret_cast<OutT*>(base) + idx;
  if (vec_ok && count == 4) {
    if (sizeof(OutT) == 4) {
      *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      *reinterpret_cast<uint2*>(o) = make_uint2(Elem<OutT>::pack2(v[0], v[1]), Elem<OutT>::pack2(v[2], v[3]));
    }
  } else {
    for (int i = 0; i < count; ++i) Elem<OutT>::st(o + i, v[i]);
  }
}

// N consecutive elements of T (N * sizeof(T) = 8 or 16 bytes, naturally aligned) as floats
template <typename T, int N>
__device__ __forceinline__ void load_elems(const T* p, float (&v)[N]) {
  if (sizeof(T) == 4) {
    static_assert(sizeof(T) != 4 || N == 4, "fp32 rows are read 4 at a time");
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2 % N] = t.z; v[3 % N] = t.w;
  } else if (N == 8) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[(2 * i) % N] = Elem<T>::lo(w[i]);
      v[(2 * i + 1) % N] = Elem<T>::hi(w[i]);
    }
  } else {  // 4 bf16 = 8 bytes
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = Elem<T>::lo(t.x); v[1] = Elem<T>::hi(t.x);
    v[2 % N] = Elem<T>::lo(t.y); v[3 % N] = Elem<T>::hi(t.y);
  }
}

// the same from a 16-byte register image (prefetched epilogue operands; only N*sizeof(T) == 16)
template <typename T, int N>
__device__ __forceinline__ void unpack_elems(const uint4& t, float (&v)[N]) {
  if (sizeof(T) == 4) {
    v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y);
    v[2 % N] = __uint_as_float(t.z); v[3 % N] = __uint_as_float(t.w);
  } else {
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[(2 * i) % N] = Elem<T>::lo(w[i]);
      v[(2 * i + 1) % N] = Elem<T>::hi(w[i]);
    }
  }
}

// 16 / 8-byte epilogue accesses, plain or with the non-temporal hint (GP::nt_epi)
typedef unsigned int u32x4_vec __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_vec __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st16_epi(bool nt, void* p, uint4 v) {
  if (nt) { const u32x4_vec t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<u32x4_vec*>(p)); }
  else *reinterpret_cast<uint4*>(p) = v;
}
__device__ __forceinline__ void st16_epi(bool nt, void* p, float4 v) {
  st16_epi(nt, p, make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)));
}
__device__ __forceinline__ void st8_epi(bool nt, void* p, uint2 v) {
  if (nt) { const u32x2_vec t = {v.x, v.y}; __builtin_nontemporal_store(t, reinterpret_cast<u32x2_vec*>(p)); }
  else *reinterpret_cast<uint2*>(p) = v;
}
__device__ __forceinline__ uint4 ld16_epi(bool nt, const void* p) {
  if (nt) { const u32x4_vec t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_vec*>(p)); return make_uint4(t.x, t.y, t.z, t.w); }
  return *reinterpret_cast<const uint4*>(p);
}
// N consecutive elements of T as floats (load_elems), the 16-byte forms through ld16_epi
template <typename T, int N>
__device__ __forceinline__ void load_elems_epi(bool nt, const T* p, float (&v)[N]) {
  if constexpr (N * sizeof(T) == 16) unpack_elems<T, N>(ld16_epi(nt, p), v);
  else load_elems<T, N>(p, v);
}

// XCD-aware remap of a linear workgroup id: consecutive ids on one XCD share operand panels.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int nx = 8;
  int q = nwg / nx, r = nwg % nx;
  int xcd = bid % nx, idx = bid / nx;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// ---- LDS transpose reads of the TN kernels ([position][channel] tiles, ds_read_b64_tr_b16) ----------
typedef __attribute__((ext_vector_type(4))) short s16x4_v;

template <int RS> __device__ __forceinline__ int tr_key(int row) {
  return RS == 256 ? ((row & 3) | (((row >> 3) & 1) << 2)) : (((row >> 1) & 1) | (((row >> 3) & 1) << 1));
}
// fragment: channels c0..c0+15 (lane -> c0 + (l & 15)), positions ks*32 + 8*(l>>4) .. +7
template <int RS, typename V = bf16x8_v>
__device__ __forceinline__ V tr_frag(const char* tile, int c0, int ks, int lane) {
  const int g = lane >> 4, pl = lane & 15;
  const int seg = c0 >> 4;
  const int r0 = ks * 32 + 8 * g + (pl >> 2);
  const int r1 = r0 + 4;
  const s16x4_v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4_v*)(tile + r0 * RS + ((seg ^ tr_key<RS>(r0)) << 5) + ((pl & 3) << 3)));
  const s16x4_v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4_v*)(tile + r1 * RS + ((seg ^ tr_key<RS>(r1)) << 5) + ((pl & 3) << 3)));
  union { struct { s16x4_v a, b; } s; V v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}


}  // namespace
}  // namespace vlfb
