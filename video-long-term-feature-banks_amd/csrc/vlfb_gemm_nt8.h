// The 256-row, phase-pipelined NT kernel (see vlfb_gemm8.hip for the design notes), shared by vlfb_gemm8.hip (16-bit
// instances) and vlfb_gemm_pair.hip (the two-plane fp16 instances, PAIR).
#pragma once
#include "vlfb_gemm_common.h"

namespace vlfb {
namespace {

#define VLFB_BAR() asm volatile("s_barrier" ::: "memory")
#define VLFB_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// scalar cursor of one activation staging stream over the k-tiles of a gathered conv
struct TapCur {
  int a, b, c, tap;
  int ci;      // byte offset inside the tap's channel run
};
template <bool PAIR = false>
__device__ __forceinline__ void tap_advance(const GP& p, TapCur& u) {
  u.ci += PAIR ? 64 : 128;
  if (u.ci >= p.Cs * 2) {
    u.ci = 0;
    ++u.tap;
    if (++u.c == p.kw) { u.c = 0; if (++u.b == p.kh) { u.b = 0; ++u.a; } }
  }
}
template <bool DGRAD>
__device__ __forceinline__ unsigned tap_delta_bytes(const GP& p, const TapCur& u) {
  const int sgn = DGRAD ? -1 : 1;
  const int pix = sgn * ((u.a * p.dt * p.Hs + u.b * p.dh) * p.Ws + u.c * p.dw);
  return (unsigned)(pix * p.lda * 2 + u.ci);
}

// MODE: 0 = plain rows (1x1x1 convs, batched GEMMs), 1 = gathered FPROP, 2 = gathered unit-stride DGRAD.
// KTAIL: K is not a multiple of 64 (attention products with K = 784): chunks past K are zero-filled.
// RV: valid rows per wave row (128, or 98 = half a 14 x 14 frame: the activations of 224^2 clips have
// 196 k positions per frame at every stage, and 8 clips x 16 frames give 256 k tiles of 196 rows --
// whole rounds of workgroups on 256 CUs -- where 256-row tiles leave a quarter of the chip idle).  With
// RV = 98 the seventh fragment of a wave row is partly padding (zero-filled rows, 7/8 of the MFMAs).
// PAIR (fp16; vlfb_conv_desc.math = VLFB_MATH_F16X3): both operands as two fp16 planes (GP::a_ps / b_ps elements apart).  A
// 128-byte LDS row = 32 k of the hi plane | the same 32 k of the lo plane, a k-tile is 32 k, and the two k-steps of a row
// become the three products lo.hi + hi.lo + hi.hi: 24 MFMAs per phase on the same fragment reads and DMA pieces as the 16 of
// the plain form (vlfb_gemm_nt.h has the same mode for the 128-row kernel; the two are bit-identical).
template <typename T, typename OutT, int BN, int MODE, bool KTAIL, int RV, bool PAIR = false>
__global__ __launch_bounds__(512) void gemm_nt8_kernel(const GP p) {
  typedef typename V16<T>::V vec_t;
  static_assert(!PAIR || (MODE != 2 && !KTAIL), "PAIR: plain rows or gathered FPROP, K in whole 32-k tiles");
  constexpr unsigned KTB = PAIR ? 64u : 128u;      // bytes of K (per plane) a k-tile advances
  static_assert(RV == 128 || RV == 98, "rows per wave row");
  constexpr int FM1 = (RV - 64 + 15) / 16;      // fragments of the second quadrant row (4 or 3)
  constexpr int NBH = BN / 128;                  // B half-tiles per k-tile
  constexpr int NHT = 2 + NBH;                   // half-tiles per k-tile (A0, A1, B0[, B1])
  constexpr int NBUF = BN == 256 ? 2 : 3;        // k-tiles in the LDS ring
  constexpr int HT = 128 * 128;                  // bytes of one half-tile slot
  constexpr int BUFSZ = NHT * HT;
  constexpr int FN = BN / 64;                    // 16-channel fragments per wave
  static_assert(BN == 256 || BN == 128, "tile widths");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l15 = lane & 15, g = lane >> 4;

  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = xcd_remap(blockIdx.x, nwg);
  const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
  const int m0 = tile_m * (2 * RV), n0 = tile_n * BN;
  const int z = blockIdx.z;
  const char* Ab = p.A + (long long)z * p.a_bs * 2;
  const char* Bb = p.B + (long long)z * p.b_bs * 2;
  const auto rsA = make_rsrc(Ab, p.a_bytes);
  const auto rsB = make_rsrc(Bb, p.b_bytes);
  const int T_ = PAIR ? (p.K + 31) >> 5 : (p.K + 63) >> 6;                // k-tiles

  // ---- staging assignment: piece (row r0 + 64 i, 16-byte slot tid & 7) of every half-tile -----------
  const int r0 = tid >> 3;                                   // 0..63
  const int ccg = (tid & 7) ^ (r0 & 7);                      // global chunk fetched into LDS slot tid & 7
  // byte offset of that chunk inside a row's k-tile: PAIR = 16 bytes at k-offset (ccg & 3) * 8 of plane ccg >> 2
  const unsigned a_cb = PAIR ? (unsigned)(ccg & 3) * 16u + (unsigned)(ccg >> 2) * (unsigned)p.a_ps * 2u : (unsigned)ccg * 16u;
  const unsigned b_cb = PAIR ? (unsigned)(ccg & 3) * 16u + (unsigned)(ccg >> 2) * (unsigned)p.b_ps * 2u : (unsigned)ccg * 16u;
  unsigned aoff[2][2], amask[2][2], boff[NBH][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + i * RV + h * 64 + r0;               // logical row of LDS row (r0 + 64 i) of slot A_h
      const bool ok = m < p.M && h * 64 + r0 < RV;
      amask[h][i] = 0;
      if (MODE == 0) {
        aoff[h][i] = ok ? (unsigned)(m * p.lda) * 2u + a_cb : kOOB;
      } else {
        RowC r = decode_row(p, ok ? m : 0);
        if (MODE == 1) { r.t = r.t * p.st - p.pt; r.h = r.h * p.sh - p.ph; r.w = r.w * p.sw - p.pw; }
        else { r.t += p.pt; r.h += p.ph; r.w += p.pw; }
        const int pix = ((r.n * p.Ts + r.t) * p.Hs + r.h) * p.Ws + r.w;
        aoff[h][i] = (unsigned)(pix * p.lda) * 2u + a_cb;       // wraps for padding rows (masked below)
        unsigned mk = 0;
        int tap = 0;
        const int sgn = MODE == 2 ? -1 : 1;
        for (int a = 0; a < p.kt; ++a)
          for (int b = 0; b < p.kh; ++b)
            for (int c = 0; c < p.kw; ++c, ++tap) {
              const bool v = (unsigned)(r.t + sgn * a * p.dt) < (unsigned)p.Ts &&
                             (unsigned)(r.h + sgn * b * p.dh) < (unsigned)p.Hs &&
                             (unsigned)(r.w + sgn * c * p.dw) < (unsigned)p.Ws;
              mk |= (v ? 1u : 0u) << tap;
            }
        amask[h][i] = ok ? mk : 0u;
      }
    }
#pragma unroll
  for (int h = 0; h < NBH; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rho = r0 + 64 * i;
      const int n = n0 + (BN == 256 ? (rho >> 5) * 64 + h * 32 + (rho & 31) : rho);
      boff[h][i] = n < p.Ncols ? (unsigned)(n * p.ldb) * 2u + b_cb : kOOB;
    }

  // two activation staging streams (slot A0 runs up to two k-tiles ahead, slot A1 one), one cursor each
  TapCur cur[2];
  int akt[2] = {0, 0};                            // next k-tile of each A stream
#pragma unroll
  for (int h = 0; h < 2; ++h) { cur[h].a = cur[h].b = cur[h].c = cur[h].tap = 0; cur[h].ci = 0; }

  auto stage_a = [&](int h, int buf) {            // stages k-tile akt[h] of slot A_h into ring buffer `buf`
    const int kt = akt[h];
    const bool live = kt < T_;
    char* dst = smem + buf * BUFSZ + h * HT + wave * 1024;
    if (MODE == 0) {
      const bool kok = !KTAIL || (kt * 8 + ccg) * 8 < p.K;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        bufglds16(rsA, (live && kok) ? aoff[h][i] : kOOB, (unsigned)kt * KTB, dst + i * 8192);
    } else {
      const unsigned dbyte = tap_delta_bytes<MODE == 2>(p, cur[h]);
      const int tap = cur[h].tap;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool ok = live && ((amask[h][i] >> tap) & 1u);
        bufglds16(rsA, ok ? aoff[h][i] + dbyte : kOOB, 0, dst + i * 8192);
      }
      tap_advance<PAIR>(p, cur[h]);
    }
    akt[h] = kt + 1;
  };
  int bkt[NBH];
#pragma unroll
  for (int h = 0; h < NBH; ++h) bkt[h] = 0;
  auto stage_b = [&](int h, int buf) {
    const int kt = bkt[h];
    const bool live = kt < T_;
    const bool kok = !KTAIL || (kt * 8 + ccg) * 8 < p.K;
    char* dst = smem + buf * BUFSZ + (2 + h) * HT + wave * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      bufglds16(rsB, (live && kok) ? boff[h][i] : kOOB, (unsigned)kt * KTB, dst + i * 8192);
    bkt[h] = kt + 1;
  };

  // ---- fragment addressing ---------------------------------------------------------------------------
  const int key = l15 & 7;
  const int kof0 = ((0 + g) ^ key) << 4, kof1 = ((4 + g) ^ key) << 4;
  const int ra = (wm * 64 + l15) * 128;           // + ah * HT + ii * 2048
  const int rb = (wn * 32 + l15) * 128;           // + (2 + bh) * HT + jj * 2048

  f32x4_v acc[FN][8];
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = f32x4_v{0.f, 0.f, 0.f, 0.f};
  vec_t xa[4][2], wb[NBH][2][2];

  auto read_a = [&](const char* buf, int ah) {
    const char* s = buf + ah * HT + ra;
#pragma unroll
    for (int ii = 0; ii < (ah ? FM1 : 4); ++ii) {
      xa[ii][0] = *reinterpret_cast<const vec_t*>(s + ii * 2048 + kof0);
      xa[ii][1] = *reinterpret_cast<const vec_t*>(s + ii * 2048 + kof1);
    }
  };
  auto read_b = [&](const char* buf, int bh) {
    const char* s = buf + (2 + bh) * HT + rb;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      wb[bh][jj][0] = *reinterpret_cast<const vec_t*>(s + jj * 2048 + kof0);
      wb[bh][jj][1] = *reinterpret_cast<const vec_t*>(s + jj * 2048 + kof1);
    }
  };
  auto mma_quadrant = [&](int ah, int bh) {
    __builtin_amdgcn_s_setprio(1);
    if constexpr (PAIR) {
      // k-step 0 of a row = the hi plane's 32 k, k-step 1 = the lo plane's: lo.hi + hi.lo + hi.hi, small terms first
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int ii = 0; ii < (ah ? FM1 : 4); ++ii)
            acc[bh * 2 + jj][ah * 4 + ii] = V16<T>::mma(wb[bh][jj][t == 0 ? 1 : 0], xa[ii][t == 1 ? 1 : 0], acc[bh * 2 + jj][ah * 4 + ii]);
    } else {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int ii = 0; ii < (ah ? FM1 : 4); ++ii)
          acc[bh * 2 + jj][ah * 4 + ii] = V16<T>::mma(wb[bh][jj][ks], xa[ii][ks], acc[bh * 2 + jj][ah * 4 + ii]);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  // one phase: the reads were issued by the caller; DMA issue, counted wait, barrier, MFMAs, barrier
#define VLFB_PHASE_TAIL(N_INFLIGHT, AH, BH)           \
  VLFB_VMCNT(N_INFLIGHT);                             \
  VLFB_BAR();                                         \
  __builtin_amdgcn_sched_barrier(0);                  \
  mma_quadrant(AH, BH);                               \
  __builtin_amdgcn_sched_barrier(0);                  \
  VLFB_BAR();                                         \
  __builtin_amdgcn_sched_barrier(0)

  // ---- prologue ------------------------------------------------------------------------------------------
  if (BN == 256) {
    // issue order = steady-state order: A0(0) B0(0) B1(0) A1(0) A0(1) B0(1)
    stage_a(0, 0); stage_b(0, 0); stage_b(NBH - 1, 0); stage_a(1, 0); stage_a(0, 1); stage_b(0, 1);
    VLFB_VMCNT(8);                               // A0(0), B0(0) landed
  } else {
    // [A0 B](0) A1(0) [A0 B](1) A1(1)
    stage_a(0, 0); stage_b(0, 0); stage_a(1, 0); stage_a(0, 1); stage_b(0, 1); stage_a(1, 1);
    VLFB_VMCNT(8);                               // [A0 B](0) landed
  }
  VLFB_BAR();
  if (wm == 1) VLFB_BAR();                       // stagger: wave row 1 runs one barrier interval behind
  __builtin_amdgcn_sched_barrier(0);

  if (BN == 256) {
    int cb = 0;                                   // ring buffer of k-tile kt
    for (int kt = 0; kt < T_; ++kt) {
      const char* bc = smem + cb * BUFSZ;
      const int nb = cb ^ 1;
      // phase 0: quadrant (0,0); refill B1 of the other buffer with k-tile kt+1
      read_a(bc, 0); read_b(bc, 0);
      stage_b(NBH - 1, nb);
      VLFB_PHASE_TAIL(8, 0, 0);
      // phase 1: quadrant (0,1); A1(kt+1)
      read_b(bc, NBH - 1);
      stage_a(1, nb);
      VLFB_PHASE_TAIL(8, 0, NBH - 1);
      // phase 2: quadrant (1,1); slot A0 of this buffer was last read in phase 0: refill with kt+2
      read_a(bc, 1);
      stage_a(0, cb);
      VLFB_PHASE_TAIL(8, 1, NBH - 1);
      // phase 3: quadrant (1,0), fragments already in registers; B0(kt+2)
      stage_b(0, cb);
      VLFB_PHASE_TAIL(8, 1, 0);
      cb = nb;
    }
  } else {
    int cb = 0;
    for (int kt = 0; kt < T_; ++kt) {
      const char* bc = smem + cb * BUFSZ;
      const int sb = cb == 0 ? 2 : cb - 1;        // buffer of k-tile kt+2 (= the one k-tile kt-1 used)
      read_a(bc, 0); read_b(bc, 0);
      stage_a(0, sb); stage_b(0, sb);
      VLFB_PHASE_TAIL(10, 0, 0);
      read_a(bc, 1);
      stage_a(1, sb);
      VLFB_PHASE_TAIL(8, 1, 0);
      cb = cb == 2 ? 0 : cb + 1;
    }
  }
  if (wm == 0) VLFB_BAR();                       // re-align the two wave rows
  VLFB_VMCNT(0);                                 // the zero-fill DMAs of the tail have landed too
  VLFB_BAR();
  __builtin_amdgcn_sched_barrier(0);

  // ---- epilogue: fp32 tile -> LDS (swizzled) -> 16 bytes per lane on whole rows -------------------------
  char* Ob = p.O + (long long)z * p.o_bs * (long long)sizeof(OutT);
  const char* Rb = p.R ? p.R + (long long)z * p.r_bs * 2 : nullptr;
  const char* Mb = p.Mask ? p.Mask + (long long)z * p.r_bs * 2 : nullptr;
  const char* R2b = p.R2 ? p.R2 + (long long)z * p.r_bs * 2 : nullptr;             // low terms (GP::R2 / O2)
  char* O2b = p.O2 ? p.O2 + (long long)z * p.o_bs * 2ll : nullptr;          // (16-bit whatever OutT is: low term / 16-bit copy)
  (void)R2b; (void)O2b;
  constexpr int EPT = 16 / (int)sizeof(OutT);
  constexpr int TPR = BN / EPT;                  // lanes per output row
  constexpr int RPI = 512 / TPR;                 // rows per iteration of the store loop
  constexpr int CPR = BN / 4;                    // 16-byte fp32 chunks per staged row
  constexpr int NPASS = BN == 256 ? 2 : 1;       // 128 KiB of staging per pass
  constexpr int ROWS_PASS = 256 / NPASS;
  const int tc = tid % TPR, tr = tid / TPR;
  const int ncol = n0 + tc * EPT;
  const bool nt = p.nt_epi != 0;                 // epilogue rows with the non-temporal hint (GP::nt_epi)
#pragma unroll
  for (int s = 0; s < NPASS; ++s) {
    if (s > 0) __syncthreads();
#pragma unroll
    for (int ii = 0; ii < 8 / NPASS; ++ii) {
      const int i = s * (8 / NPASS) + ii;
      if (i >= 4 + FM1) continue;                                        // RV = 98: the eighth fragment does not exist
      const int sr = wm * (ROWS_PASS / 2) + ii * 16 + l15;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int ch = (BN == 256) ? wn * 64 + (j >> 1) * 32 + (j & 1) * 16 + g * 4 : wn * 32 + j * 16 + g * 4;
        const int c = ch >> 2;
        *reinterpret_cast<float4*>(smem + ((sr * CPR + (c ^ (sr & 7))) << 4)) =
            make_float4(acc[j][i][0] * p.alpha, acc[j][i][1] * p.alpha, acc[j][i][2] * p.alpha, acc[j][i][3] * p.alpha);
      }
    }
    __syncthreads();
#pragma unroll 2
    for (int it = 0; it < ROWS_PASS / RPI; ++it) {
      const int sr = it * RPI + tr;                                       // staged row
      const int wrow = s * (ROWS_PASS / 2) * (NPASS - 1) + (sr % (ROWS_PASS / 2));   // row inside its wave row
      const int m = m0 + (sr / (ROWS_PASS / 2)) * RV + wrow;
      if (wrow < RV && m < p.M && ncol < p.Ncols) {
        float v[EPT];
#pragma unroll
        for (int q = 0; q < EPT / 4; ++q) {
          const int c = tc * (EPT / 4) + q;
          const float4 t = *reinterpret_cast<const float4*>(smem + ((sr * CPR + (c ^ (sr & 7))) << 4));
          v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
        if (p.bias_mode == VLFB_BIAS_COL) {
#pragma unroll
          for (int e = 0; e < EPT; ++e) v[e] += p.bias[ncol + e];
        } else if (p.bias_mode == VLFB_BIAS_ROW) {
          const float b = p.bias[m];
#pragma unroll
          for (int e = 0; e < EPT; ++e) v[e] += b;
        }
        const long long ridx = (long long)m * p.ldr + ncol;
        if (Rb) {
          float r[EPT];
          load_elems_epi<T, EPT>(nt, reinterpret_cast<const T*>(Rb) + ridx, r);
#pragma unroll
          for (int e = 0; e < EPT; ++e) v[e] += r[e];
        }
        if constexpr (sizeof(OutT) == 2) {
          if (R2b) {
            float r[EPT];
            load_elems_epi<T, EPT>(nt, reinterpret_cast<const T*>(R2b) + ridx, r);
#pragma unroll
            for (int e = 0; e < EPT; ++e) v[e] += r[e];
          }
        }
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < EPT; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (Mb) {
          float r[EPT];
          load_elems_epi<T, EPT>(nt, reinterpret_cast<const T*>(Mb) + ridx, r);
#pragma unroll
          for (int e = 0; e < EPT; ++e) v[e] = r[e] > 0.f ? v[e] : 0.f;
        }
        OutT* o = reinterpret_cast<OutT*>(Ob) + (long long)m * p.ldo + ncol;
        if (sizeof(OutT) == 4) {
          st16_epi(nt, o, make_float4(v[0], v[1], v[2], v[3]));
          if constexpr (PAIR) {       // O2 = the fp16 copy of an fp32 output (vlfb_gemm_nt.h)
            if (O2b) st8_epi(nt, reinterpret_cast<unsigned short*>(O2b) + (long long)m * p.ldo + ncol,
                             make_uint2(pack_h2_pos(v[0], v[1]), pack_h2_pos(v[2 % EPT], v[3 % EPT])));
          } else {                    // ... of a 16-bit launch: the same values rounded to T
            if (O2b) st8_epi(nt, reinterpret_cast<T*>(O2b) + (long long)m * p.ldo + ncol,
                             make_uint2(Elem<T>::pack2(v[0], v[1]), Elem<T>::pack2(v[2 % EPT], v[3 % EPT])));
          }
        } else {
          const uint4 hv = make_uint4(Elem<OutT>::pack2(v[0], v[1]), Elem<OutT>::pack2(v[2 % EPT], v[3 % EPT]),
                                      Elem<OutT>::pack2(v[4 % EPT], v[5 % EPT]), Elem<OutT>::pack2(v[6 % EPT], v[7 % EPT]));
          st16_epi(nt, o, hv);
          if (O2b) {
            float h[EPT];
            unpack_elems<OutT, EPT>(hv, h);
            st16_epi(nt, reinterpret_cast<OutT*>(O2b) + (long long)m * p.ldo + ncol,
                     make_uint4(Elem<OutT>::pack2(v[0] - h[0], v[1] - h[1]), Elem<OutT>::pack2(v[2 % EPT] - h[2 % EPT], v[3 % EPT] - h[3 % EPT]),
                                Elem<OutT>::pack2(v[4 % EPT] - h[4 % EPT], v[5 % EPT] - h[5 % EPT]), Elem<OutT>::pack2(v[6 % EPT] - h[6 % EPT], v[7 % EPT] - h[7 % EPT])));
          }
        }
      }
    }
  }
#undef VLFB_PHASE_TAIL
}

}  // namespace
}  // namespace vlfb
