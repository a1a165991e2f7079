// Split-bf16 implicit GEMM on the bf16 matrix cores with fp32 STORAGE (gfx950): the parity-grade path.
//
// The exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) runs at the fp32 VECTOR rate, 1/16 of the bf16 matrix rate.  Here every
// fp32 operand value x is expanded into bf16 terms  x = h + m (+ l),  h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)
// (round-to-nearest-even, the differences are exact in fp32), and a product is the sum of the bf16 products whose
// weight is above the fp32 rounding level, accumulated by v_mfma_f32_16x16x32_bf16 in fp32:
//   math 6 ("bf16x6", 3 terms per operand, 24+ significand bits):  hh + hm + mh + mm + hl + lh   -- the FORWARD convs
//       and attention products: a forward error above ~2^-21 flips enough ReLU / max-pool decisions to push parameter
//       gradients past 1e-3 (scratch/emu_split.py: 16-bit forward operands -> conv1_w gradient off by 1e-2);
//   math 3 ("bf16x3", 2 terms per operand, 16 bits):  hh + hm + mh  (error ~2^-17 per product, random sign) -- dgrad,
//       wgrad and the backward attention products, where nothing is thresholded (gradients 3e-5 off, same experiment).
// Small terms are issued first; 16 independent accumulators sit between two MFMAs on the same one.
//
// NT kernel (FPROP / DGRAD / batched NT GEMM): the activation operand is read as fp32 (buffer_load ... lds DMA, the
// gathers of vlfb_gemm.hip: identity rows, scalar-tap cursor, per-lane tap decode, packed stem) and split when a
// wave loads its fragments (10 VALU per value pair for three terms, hidden under 96 MFMAs per k-tile and wave); the
// WEIGHT operand arrives PRE-SPLIT as NPL bf16 planes [plane][n][K] (vlfb_weight_prep* with dtype VLFB_SPLIT,
// vlfb_split_planes for the attention operands), so it costs no VALU in the loop at all.
// TN kernel (WGRAD, contract-over-positions products): both operands are activations with the contraction index as
// the slow dimension; the register-staged transposing stager of the fp32 kernel splits each 4x4 block once on its
// way into LDS ([channel][32 positions] rows = 64 bytes of h | 64 bytes of m), the MFMA loop reads bf16 fragments.
//
// LDS bank maps (ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... of
// MI355X_MICROARCH.md): fp32 activation rows of 128 bytes are read as chunk pairs (2g, 2g+1): key128 below; bf16 plane
// rows of 64 bytes as chunk g: key64.  Both conflict-free for those groups (worked out in DESIGN.md 3.1f).
#include "vlfb_gemm_common.h"
#include <type_traits>

namespace vlfb {
namespace {

__device__ __forceinline__ int key128(int row) {
  const int e = (row >> 1) & 7;
  return e ^ (((e + 2) >> 1) & 2);
}
__device__ __forceinline__ int key64(int row) { return (4 - ((row >> 2) & 3)) & 3; }   // 0, 3, 2, 1 per four rows

// fp32 pair -> packed bf16 pair of the leading term, and the exact remainders
__device__ __forceinline__ uint32_t peel(float& a, float& b) {
  const uint32_t hp = pack_bf2(a, b);
  a -= __uint_as_float(hp << 16);
  b -= __uint_as_float(hp & 0xffff0000u);
  return hp;
}
__device__ __forceinline__ uint32_t word_sel(const uint4& v, int i) {
  return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
}
template <int NPL>
struct Terms { bf16x8_v t[NPL]; };

template <int NPL>
__device__ __forceinline__ Terms<NPL> split_frag(const float4 lo, const float4 hi) {
  float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  uint32_t w[NPL][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int pl = 0; pl < NPL - 1; ++pl) w[pl][i] = peel(x[2 * i], x[2 * i + 1]);
    w[NPL - 1][i] = pack_bf2(x[2 * i], x[2 * i + 1]);
  }
  Terms<NPL> r;
#pragma unroll
  for (int pl = 0; pl < NPL; ++pl) {
    const uint4 v = make_uint4(w[pl][0], w[pl][1], w[pl][2], w[pl][3]);
    r.t[pl] = __builtin_bit_cast(bf16x8_v, v);
  }
  return r;
}

// compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N-1 (indices usable in `if constexpr`)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for_impl(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_impl<I + 1, N>(f);
  }
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl<0, N>(f); }

__device__ __forceinline__ f32x4_v mma_bf16(bf16x8_v a, bf16x8_v b, f32x4_v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// =============================================================================================
// Epilogue of the NT kernels (4 waves as 2 x 2, 64 x BN/2 outputs each): fp32 tile through LDS, then 16-byte row
// accesses: v = alpha * acc + bias + R; relu; mask; fp32 store -- and, when the launch asks for them (GP::op_n == 2),
// the first two bf16 terms of v as planes [plane][row][ldo]: the operand format a later split launch reads without
// spending VALU on the expansion (vlfb_conv_desc.o_planes).
// =============================================================================================
// S2 (class-major rows of a (1, 2, 2)-strided DGRAD, see gemm_nt_sp_kernel): tile row m of class (ph, pw) is the input
// position (n, t, 2 h2 + ph, 2 w2 + pw); the residual / mask / output rows are addressed through that map.
__device__ __forceinline__ long long s2_row_pos(const GP& p, int m, int ph, int pw) {
  const int w2n = p.Wr >> 1, h2n = p.Hr >> 1;
  const int w2 = m % w2n;
  int q = m / w2n;
  const int h2 = q % h2n;
  q /= h2n;                                        // n * Tr + t
  return ((long long)q * p.Hr + 2 * h2 + ph) * p.Wr + 2 * w2 + pw;
}

template <int BN, int FN, int FM, bool S2 = false>
__device__ __forceinline__ void nt_epilogue(const GP& p, const f32x4_v (&acc)[FN][FM], char* smem, int m0, int n0, int z, int tid,
                                            int wm, int wn, int l15, int g, int s2_ph = 0, int s2_pw = 0) {
  constexpr int BM = 128, NTHR = 256, WM = FM * 16, WN = FN * 16;      // (the wave tile: 64 x BN/2 as 2 x 2 waves, 32 x BN as 4 x 1)
  const int mrows = S2 ? p.s2_mq : p.M;
  // p.pair_io: O / O2 and R / R2 are fp16 planes (hi, lo) of the values instead of fp32 tensors (GP::pair_io)
  const bool pair = p.pair_io != 0;
  const int oes = pair ? 2 : 4;
  char* Ob = p.O + (long long)z * p.o_bs * oes;
  const char* Rb = p.R ? p.R + (long long)z * p.r_bs * oes : nullptr;
  const char* Mb = p.Mask ? p.Mask + (long long)z * p.r_bs * 4 : nullptr;
  constexpr int TPR = BN / 4, RPP = NTHR / TPR, NPASS = BM / RPP, CPR = BN / 4;
  const int tc = tid % TPR, tr = tid / TPR;
  const int ncol = n0 + tc * 4;
  // The residual / mask rows of ALL passes are requested first, before the accumulators go through LDS: a thin-K launch
  // (the 1x1x1 layers: 16 k-tiles) is otherwise bound by NPASS dependent HBM round trips per workgroup -- measured on
  // res5 branch2c (K = 512, residual + ReLU): 1005 us in the step against 318 us for the same GEMM without a residual.
  float4 rv[NPASS], mv[NPASS];
#pragma unroll
  for (int gp = 0; gp < NPASS; ++gp) {
    const int m = m0 + gp * RPP + tr;
    const bool ok = m < mrows && ncol < p.Ncols;
    const long long mo = S2 ? s2_row_pos(p, ok ? m : 0, s2_ph, s2_pw) : (long long)(ok ? m : 0);
    const long long ridx = mo * p.ldr + (ok ? ncol : 0);
    if (pair) {
      rv[gp] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (Rb && ok) {
        const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(Rb) + ridx);
        const uint2 l = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p.R2 + (long long)z * p.r_bs * 2) + ridx);
        rv[gp] = make_float4(h2f((unsigned short)(h.x & 0xffffu)) + h2f((unsigned short)(l.x & 0xffffu)), h2f((unsigned short)(h.x >> 16)) + h2f((unsigned short)(l.x >> 16)),
                             h2f((unsigned short)(h.y & 0xffffu)) + h2f((unsigned short)(l.y & 0xffffu)), h2f((unsigned short)(h.y >> 16)) + h2f((unsigned short)(l.y >> 16)));
      }
    } else
    rv[gp] = (Rb && ok) ? *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(Rb) + ridx) : make_float4(0.f, 0.f, 0.f, 0.f);
    mv[gp] = (Mb && ok) ? *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(Mb) + ridx) : make_float4(1.f, 1.f, 1.f, 1.f);
  }
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int row = wm * WM + i * 16 + l15;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int c = (wn * WN + j * 16 + g * 4) >> 2;
      *reinterpret_cast<float4*>(smem + ((row * CPR + (c ^ (row & 7))) << 4)) =
          make_float4(acc[j][i][0] * p.alpha, acc[j][i][1] * p.alpha, acc[j][i][2] * p.alpha, acc[j][i][3] * p.alpha);
    }
  }
  __syncthreads();
  const bool planes = p.op_n == 2, half_copy = p.op_n == 1;
  float4 bc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias_mode == VLFB_BIAS_COL && ncol < p.Ncols) bc = *reinterpret_cast<const float4*>(p.bias + ncol);
#pragma unroll
  for (int gp = 0; gp < NPASS; ++gp) {
    const int row = gp * RPP + tr;
    const int m = m0 + row;
    if (m < mrows && ncol < p.Ncols) {
      const long long mo = S2 ? s2_row_pos(p, m, s2_ph, s2_pw) : (long long)m;
      const float4 t = *reinterpret_cast<const float4*>(smem + ((row * CPR + (tc ^ (row & 7))) << 4));
      float v[4] = {t.x + bc.x, t.y + bc.y, t.z + bc.z, t.w + bc.w};
      if (p.bias_mode == VLFB_BIAS_ROW) {
        const float b = p.bias[mo];
        v[0] += b; v[1] += b; v[2] += b; v[3] += b;
      }
      v[0] += rv[gp].x; v[1] += rv[gp].y; v[2] += rv[gp].z; v[3] += rv[gp].w;
      if (p.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      v[0] = mv[gp].x > 0.f ? v[0] : 0.f; v[1] = mv[gp].y > 0.f ? v[1] : 0.f;
      v[2] = mv[gp].z > 0.f ? v[2] : 0.f; v[3] = mv[gp].w > 0.f ? v[3] : 0.f;
      const long long oidx = mo * p.ldo + ncol;
      if (pair) {
        const unsigned short h0 = f2h(v[0]), h1 = f2h(v[1]), h2 = f2h(v[2]), h3 = f2h(v[3]);
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(Ob) + oidx) = make_uint2((uint32_t)h0 | ((uint32_t)h1 << 16), (uint32_t)h2 | ((uint32_t)h3 << 16));
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.O2 + (long long)z * p.o_bs * 2) + oidx) =
            make_uint2((uint32_t)f2h(v[0] - h2f(h0)) | ((uint32_t)f2h(v[1] - h2f(h1)) << 16), (uint32_t)f2h(v[2] - h2f(h2)) | ((uint32_t)f2h(v[3] - h2f(h3)) << 16));
        continue;
      }
      st16_epi(p.nt_epi != 0, reinterpret_cast<float*>(Ob) + oidx, make_float4(v[0], v[1], v[2], v[3]));      // (GP::nt_epi)
      if (planes) {
        bf16_t* op = reinterpret_cast<bf16_t*>(p.OP) + oidx;
        const uint32_t h01 = peel(v[0], v[1]), h23 = peel(v[2], v[3]);
        *reinterpret_cast<uint2*>(op) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(op + p.o_ps) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
      } else if (half_copy) {
        // "mix" engine: the fp16 copy of the output that the fp16 backward reads (WGRAD operand, ReLU mask)
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.OP) + (long long)z * p.o_bs + oidx) = make_uint2(pack_h2_pos(v[0], v[1]), pack_h2_pos(v[2], v[3]));
      }
    }
  }
}

// =============================================================================================
// NT: O[m][n] = sum_k X[m][k] * W[n][k];  X fp32 (gathered), W = NPL bf16 planes, O / R / Mask fp32
// =============================================================================================
// NWN: waves along n.  2 = 2 (m) x 2 (n) waves of 64 x BN/2; 1 = 4 x 1 waves of 32 x BN: every activation fragment is then
// split by ONE wave instead of two (the split is the VALU work of the loop: 2.0 -> 1.0 VALU per MFMA at BN = 128, 4.0 -> 2.0
// at BN = 64), for 1.25x the LDS fragment reads (the bf16 weight planes are read by all four waves).
template <int NPL, int BN, bool IDENT, bool DGRAD, bool PACKW, bool UT, bool S2 = false, int NWN = 2>
__global__ __launch_bounds__(256) void gemm_nt_sp_kernel(const GP p) {
  static_assert(!UT || (!IDENT && !PACKW), "UT is for gathered, unpacked operands");
  static_assert(!S2 || (UT && DGRAD), "S2 is the scalar-cursor DGRAD over the parity classes of a (1, 2, 2)-strided conv");
  static_assert(NWN == 1 || NWN == 2, "wave layouts: 2 x 2 or 4 x 1");
  typedef float T;
  constexpr int BM = 128, NTHR = 256;
  constexpr int EPC = 4;                          // fp32 elements per 16-byte chunk of the activation operand
  constexpr int RPPS_A = NTHR / 8, A_IT = BM / RPPS_A;       // 32 rows per pass, 4 passes
  constexpr int RPPS_B = NTHR / 4, B_ITP = BN / RPPS_B;      // 64 rows per pass; 2 | 1 passes per plane
  constexpr int WM = BM / (4 / NWN), WN = BN / NWN;
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int A_BYTES = BM * 128, BP_BYTES = BN * 64;
  constexpr int BUF = A_BYTES + NPL * BP_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
  const int l15 = lane & 15, g = lane >> 4;

  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = xcd_remap(blockIdx.x, nwg);
  const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
  int m0 = tile_m * BM;
  const int n0 = tile_n * BN;
  const int z = blockIdx.z;
  // S2: DGRAD of a (1, 2, 2)-strided conv.  An input position only meets the taps of its own (h, w) parity, so the rows
  // are enumerated class by class ((n, t, h / 2, w / 2) order inside a class, GP::s2_mq rows and s2_tpc tiles each) and a
  // tile walks the nb x nc taps of its class -- 9 -> 4 / 2 / 2 / 1, 1 -> 1 / 0 / 0 / 0 -- as a unit-stride scalar cursor
  // over the conv OUTPUT grid: position (h2, w2) of the class reads source (h2 + oh - ib, w2 + ow - ic).  A class
  // without taps is an epilogue-only tile (zero accumulators: residual / mask pass through).
  int s2_ph = 0, s2_pw = 0, s2_b0 = 0, s2_c0 = 0, s2_nb = 1, s2_nc = 1;
  if constexpr (S2) {
    const int cls = tile_m / p.s2_tpc;
    m0 = (tile_m - cls * p.s2_tpc) * BM;
    s2_ph = cls >> 1; s2_pw = cls & 1;
    s2_b0 = (s2_ph + p.ph) & 1; s2_c0 = (s2_pw + p.pw) & 1;
    s2_nb = p.kh > s2_b0 ? (p.kh - s2_b0 + 1) >> 1 : 0;
    s2_nc = p.kw > s2_c0 ? (p.kw - s2_c0 + 1) >> 1 : 0;
  }
  const int mrows = S2 ? p.s2_mq : p.M;

  const char* Ab = p.A + (long long)z * p.a_bs * 4;
  const char* Bb = p.B + (long long)z * p.b_bs * 2;

  // ---- staging assignment: activation tile (128-byte fp32 rows), weight plane tiles (64-byte bf16 rows) ----
  const int cc = tid & 7, r0 = tid >> 3;
  const int ccg = cc ^ key128(r0);                 // global 16-byte chunk column this lane fetches (rows r0 + 32 i share the key)
  const int cb = tid & 3, rb0 = tid >> 2;
  const int cbg = cb ^ key64(rb0);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  RowC arow[A_IT];
  bool aok[A_IT];
  int upix[UT ? A_IT : 1];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int m = m0 + r0 + RPPS_A * i;
    aok[i] = m < mrows;
    if constexpr (S2) {
      const int w2n = p.Wr >> 1, h2n = p.Hr >> 1;
      const int mm = aok[i] ? m : 0;
      RowC& r = arow[i];
      const int w2 = mm % w2n;
      int q = mm / w2n;
      const int h2 = q % h2n;
      q /= h2n;
      r.t = q % p.Tr + p.pt; r.n = q / p.Tr;
      r.h = h2 + ((s2_ph + p.ph - s2_b0) >> 1);         // source coordinates at the class's first tap
      r.w = w2 + ((s2_pw + p.pw - s2_c0) >> 1);
    } else if (!IDENT) arow[i] = decode_row(p, aok[i] ? m : 0);
    if (UT) {
      RowC& r = arow[i];
      if constexpr (!S2) {
        if (!DGRAD) { r.t = r.t * p.st - p.pt; r.h = r.h * p.sh - p.ph; r.w = r.w * p.sw - p.pw; }
        else { r.t += p.pt; r.h += p.ph; r.w += p.pw; }
      }
      upix[i] = ((r.n * p.Ts + r.t) * p.Hs + r.h) * p.Ws + r.w;
    }
  }
  unsigned boff[B_ITP], aoff[(IDENT || UT) ? A_IT : 1];
#pragma unroll
  for (int i = 0; i < B_ITP; ++i) {
    const int n = n0 + rb0 + RPPS_B * i;
    boff[i] = n < p.Ncols ? (unsigned)(n * p.ldb + cbg * 8) * 2u : kOOB;
  }
  if (IDENT || UT) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int m = m0 + r0 + RPPS_A * i;
      if (IDENT) aoff[i] = aok[i] ? (unsigned)(m * p.lda + ccg * EPC) * 4u : kOOB;
      else aoff[i] = (unsigned)(upix[i] * p.lda + ccg * EPC) * 4u;
    }
  }
  const unsigned plane_bytes = (unsigned)p.b_ps * 2u;
  int u_a = 0, u_b = 0, u_c = 0, u_ci = 0;       // UT: scalar tap cursor of the next tile to fetch

  const int ktiles = S2 ? p.kt * s2_nb * s2_nc * (p.Cs >> 5) : (p.K + 31) >> 5;

  // The DMA of a k-tile is split into its address phase (prep_tile: per-lane offsets of the ND pieces, tap cursor)
  // and ND issue slots (issue_piece) that the k-loop places between MFMAs.
  constexpr int ND = A_IT + NPL * B_ITP;
  unsigned dvo[ND], dso[ND];                       // voffset / soffset of every piece of the tile being fetched
  auto prep_tile = [&](int kt) {
    const int kc = kt * 8 + ccg;                   // global fp32 chunk index along K
    TapC tap;
    if (IDENT || UT) { tap.ok = kc * EPC < p.K; tap.a = tap.b = tap.c = tap.ci = 0; }
    else tap = decode_tap<T, PACKW>(p, kc);
    const bool kok = kc * EPC < p.K;
    const unsigned kbyte = (unsigned)kt * 128u;
    int ktb = kt;                                  // k-tile of the weight rows (S2: the cursor's tap in the full tap order)
    if constexpr (S2) ktb = ((u_a * p.kh + s2_b0 + 2 * u_b) * p.kw + s2_c0 + 2 * u_c) * (p.Cs >> 5) + (u_ci >> 5);
    if (UT) {
      const int sgn = DGRAD ? -1 : 1;
      const int da = sgn * u_a * p.dt, db = sgn * u_b * p.dh, dc = sgn * u_c * p.dw;
      const unsigned dbyte = (unsigned)(((da * p.Hs + db) * p.Ws + dc) * p.lda + u_ci) * 4u;
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const bool ok = aok[i] && (unsigned)(arow[i].t + da) < (unsigned)p.Ts &&
                        (unsigned)(arow[i].h + db) < (unsigned)p.Hs && (unsigned)(arow[i].w + dc) < (unsigned)p.Ws;
        dvo[i] = ok ? aoff[i] + dbyte : kOOB;
        dso[i] = 0;
      }
      // branch-free cursor advance (scalar selects): the k-loop body has to stay one basic block
      u_ci += 32;
      const int w0 = u_ci >= p.Cs;
      u_ci = w0 ? 0 : u_ci;
      u_c += w0;
      const int w1 = u_c == (S2 ? s2_nc : p.kw);
      u_c = w1 ? 0 : u_c;
      u_b += w1;
      const int w2 = u_b == (S2 ? s2_nb : p.kh);
      u_b = w2 ? 0 : u_b;
      u_a += w2;
    } else if (IDENT) {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) { dvo[i] = kok ? aoff[i] : kOOB; dso[i] = kbyte; }
    } else {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        unsigned off;
        if (PACKW) {
          const int ts = arow[i].t * p.st - p.pt + tap.a * p.dt;
          const int hs = arow[i].h * p.sh - p.ph + tap.b * p.dh;
          const bool ok = aok[i] && tap.ok && (unsigned)ts < (unsigned)p.Ts && (unsigned)hs < (unsigned)p.Hs;
          const int w0 = arow[i].w * p.sw - p.pw + tap.c;
          const int pix = ((arow[i].n * p.Ts + ts) * p.Hs + hs) * p.Ws + w0;
          off = ok ? (unsigned)pix * 16u : kOOB;
        } else {
          bool ok;
          const long long e = src_offset<DGRAD>(p, arow[i], tap, ok);
          off = (ok && aok[i] && tap.ok) ? (unsigned)e * 4u : kOOB;
        }
        dvo[i] = off;
        dso[i] = 0;
      }
    }
    // weight planes: 32 k = 64 bytes per row and plane
    const bool kokb = (ktb * 4 + cbg) * 8 < p.K;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int i = 0; i < B_ITP; ++i) {
        dvo[A_IT + pl * B_ITP + i] = kokb ? boff[i] : kOOB;
        dso[A_IT + pl * B_ITP + i] = (unsigned)ktb * 64u + (unsigned)pl * plane_bytes;
      }
  };
  // past the last tile the fetch still runs (one basic block, see the k-loop) but through descriptors of ZERO
  // records: every lane is out of range and the copies zero-fill a buffer nobody reads -- a scalar select, no lane work
  auto issue_piece = [&](int d, int buf, bool live) {
    char* base = smem + buf * BUF + wave_u * 1024;
    if (d < A_IT) bufglds16(make_rsrc(Ab, live ? p.a_bytes : 0u), dvo[d], dso[d], base + d * (RPPS_A * 128));
    else bufglds16(make_rsrc(Bb, live ? p.b_bytes : 0u), dvo[d], dso[d],
                   base + A_BYTES + ((d - A_IT) / B_ITP) * BP_BYTES + ((d - A_IT) % B_ITP) * (RPPS_B * 64));
  };

  f32x4_v acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b) acc[a][b] = f32x4_v{0.f, 0.f, 0.f, 0.f};

  if (ktiles > 0) {
    prep_tile(0);
#pragma unroll
    for (int d = 0; d < ND; ++d) issue_piece(d, 0, true);
  }
  // MFMA sequence of a k-tile: weight plane a x activation term b, every weight plane against the LEADING activation
  // term first -- those MFMAs need one v_cvt_pk_bf16_f32 per value pair and nothing else; the further activation terms
  // are peeled (5 VALU per pair and term) in the shadow of the MFMAs that precede their first use, and the ND DMA
  // pieces of the next tile are spread over the whole sequence.  Largest products first: the accumulator already holds
  // the sum of the earlier k-tiles, the order inside a tile does not matter for the rounding.
  constexpr int NTERM = NPL == 3 ? 6 : 3;
  constexpr int TA[6] = {0, 1, NPL == 3 ? 2 : 0, 0, 1, 0};
  constexpr int TB[6] = {0, 0, NPL == 3 ? 0 : 1, 1, 1, 2};
  constexpr int PER = FN * FM;                     // MFMAs per term
  constexpr int NMF = NTERM * PER;
  constexpr int G1 = NPL * PER;                    // MFMAs that only need the leading activation term
  constexpr int NPU = 4 * FM;                      // pair units per peeled term
  for (int kt = 0; kt < ktiles; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");     // own DMA landed; bare barrier (see vlfb_gemm.hip)
    // the next tile is fetched unconditionally (past the last tile every lane carries the out-of-range offset and
    // the copies zero-fill a buffer nobody reads), so the loop body is one basic block
    prep_tile(kt + 1);
    const int nbuf = (kt + 1) & 1;
    const bool live = kt + 1 < ktiles;
    const char* xa = smem + (kt & 1) * BUF;
    const char* wb = xa + A_BYTES;
    bf16x8_v wf[NPL][FN];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int row = wn * WN + j * 16 + l15;
        wf[pl][j] = *reinterpret_cast<const bf16x8_v*>(wb + pl * BP_BYTES + row * 64 + ((g ^ key64(row)) << 4));
      }
    float xr[FM][8];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int row = wm * WM + i * 16 + l15;
      const int key = key128(row);
      const float4 lo = *reinterpret_cast<const float4*>(xa + row * 128 + (((2 * g) ^ key) << 4));
      const float4 hi = *reinterpret_cast<const float4*>(xa + row * 128 + (((2 * g + 1) ^ key) << 4));
      xr[i][0] = lo.x; xr[i][1] = lo.y; xr[i][2] = lo.z; xr[i][3] = lo.w;
      xr[i][4] = hi.x; xr[i][5] = hi.y; xr[i][6] = hi.z; xr[i][7] = hi.w;
    }
    uint32_t xw[NPL][FM][4];                       // activation terms, packed bf16 pairs
    // leading term: the packed conversion only (one VALU per pair in front of the first MFMA)
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int q2 = 0; q2 < 4; ++q2) xw[0][i][q2] = pack_bf2(xr[i][2 * q2], xr[i][2 * q2 + 1]);
    __builtin_amdgcn_sched_barrier(0);
    static_for<NMF>([&](auto nc) {
      constexpr int n = decltype(nc)::value;
      constexpr int ts = n / PER, j = (n % PER) / FM, i = n % FM;
      const uint4 xv = make_uint4(xw[TB[ts]][i][0], xw[TB[ts]][i][1], xw[TB[ts]][i][2], xw[TB[ts]][i][3]);
      acc[j][i] = mma_bf16(wf[TA[ts]][j], __builtin_bit_cast(bf16x8_v, xv), acc[j][i]);
      // pair units due after this MFMA: term 1 spread over the first G1 MFMAs, term 2 over the 2 * PER that follow
      static_for<(NPL - 1) * NPU>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int t = 1 + u / NPU, v = u % NPU;
        constexpr int due = t == 1 ? ((v + 1) * G1 + NPU - 1) / NPU : G1 + ((v + 1) * 2 * PER + NPU - 1) / NPU;
        if constexpr (due == n + 1) {
          constexpr int fi = v / 4, q2 = v % 4;
          // remainder after the previous term (exact in fp32), then its leading bf16 pair
          const uint32_t hp = xw[t - 1][fi][q2];
          xr[fi][2 * q2] -= __uint_as_float(hp << 16);
          xr[fi][2 * q2 + 1] -= __uint_as_float(hp & 0xffff0000u);
          xw[t][fi][q2] = pack_bf2(xr[fi][2 * q2], xr[fi][2 * q2 + 1]);
        }
      });
      static_for<ND>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        if constexpr (((d + 1) * NMF) / (ND + 1) == n + 1) issue_piece(d, nbuf, live);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  __syncthreads();

  nt_epilogue<BN, FN, FM, S2>(p, acc, smem, m0, n0, z, tid, wm, wn, l15, g, s2_ph, s2_pw);
}

// =============================================================================================
// NT with the ACTIVATION operand pre-split as well: both operands are NPL bf16 term planes that the DMA drops into LDS
// as they lie -- the k-loop is fragment reads and MFMAs, no VALU (the in-kernel split above costs 2.5 VALU per MFMA in
// the three-MFMA form and every activation fragment is split by both wave columns).  Plain rows or the scalar tap cursor.
// =============================================================================================
template <int NPL, int BN, bool IDENT, bool DGRAD>
__global__ __launch_bounds__(256) void gemm_nt_pl_kernel(const GP p) {
  constexpr int BM = 128, NTHR = 256, NWN = 2;
  constexpr int RPPS = NTHR / 4;                   // 64 tile rows per DMA pass (64-byte rows, 4 chunks)
  constexpr int A_ITP = BM / RPPS, B_ITP = BN / RPPS;
  constexpr int WM = BM / 2, WN = BN / NWN;
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int PLA = BM * 64, PLB = BN * 64;
  constexpr int BUF = NPL * (PLA + PLB);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
  const int l15 = lane & 15, g = lane >> 4;
  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = xcd_remap(blockIdx.x, nwg);
  const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.z;
  const char* Ab = p.A + (long long)z * p.a_bs * 2;
  const char* Bb = p.B + (long long)z * p.b_bs * 2;

  const int cb = tid & 3, rb0 = tid >> 2;
  const int cbg = cb ^ key64(rb0);                 // (rows rb0 + 64 i share the key)
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  RowC arow[A_ITP];
  bool aok[A_ITP];
  unsigned aoff[A_ITP], boff[B_ITP];
#pragma unroll
  for (int i = 0; i < A_ITP; ++i) {
    const int m = m0 + rb0 + RPPS * i;
    aok[i] = m < p.M;
    if (IDENT) {
      aoff[i] = aok[i] ? (unsigned)(m * p.lda + cbg * 8) * 2u : kOOB;
    } else {
      RowC& r = arow[i];
      r = decode_row(p, aok[i] ? m : 0);
      if (!DGRAD) { r.t = r.t * p.st - p.pt; r.h = r.h * p.sh - p.ph; r.w = r.w * p.sw - p.pw; }
      else { r.t += p.pt; r.h += p.ph; r.w += p.pw; }
      aoff[i] = (unsigned)((((r.n * p.Ts + r.t) * p.Hs + r.h) * p.Ws + r.w) * p.lda + cbg * 8) * 2u;   // wraps for padding rows
    }
  }
#pragma unroll
  for (int i = 0; i < B_ITP; ++i) {
    const int n = n0 + rb0 + RPPS * i;
    boff[i] = n < p.Ncols ? (unsigned)(n * p.ldb + cbg * 8) * 2u : kOOB;
  }
  const unsigned a_plane = (unsigned)p.a_ps * 2u, b_plane = (unsigned)p.b_ps * 2u;
  int u_a = 0, u_b = 0, u_c = 0, u_ci = 0;
  const int ktiles = (p.K + 31) >> 5;

  constexpr int ND = NPL * (A_ITP + B_ITP);
  unsigned avo[A_ITP], aso;                        // the activation pieces of the tile being fetched
  bool kokb;
  auto prep_tile = [&](int kt) {
    kokb = (kt * 4 + cbg) * 8 < p.K;
    if (IDENT) {
#pragma unroll
      for (int i = 0; i < A_ITP; ++i) avo[i] = kokb ? aoff[i] : kOOB;
      aso = (unsigned)kt * 64u;
    } else {
      const int sgn = DGRAD ? -1 : 1;
      const int da = sgn * u_a * p.dt, db = sgn * u_b * p.dh, dc = sgn * u_c * p.dw;
      const unsigned dbyte = (unsigned)(((da * p.Hs + db) * p.Ws + dc) * p.lda + u_ci) * 2u;
#pragma unroll
      for (int i = 0; i < A_ITP; ++i) {
        const bool ok = aok[i] && (unsigned)(arow[i].t + da) < (unsigned)p.Ts &&
                        (unsigned)(arow[i].h + db) < (unsigned)p.Hs && (unsigned)(arow[i].w + dc) < (unsigned)p.Ws;
        avo[i] = ok ? aoff[i] + dbyte : kOOB;
      }
      aso = 0;
      u_ci += 32;
      const int w0 = u_ci >= p.Cs;
      u_ci = w0 ? 0 : u_ci; u_c += w0;
      const int w1 = u_c == p.kw;
      u_c = w1 ? 0 : u_c; u_b += w1;
      const int w2 = u_b == p.kh;
      u_b = w2 ? 0 : u_b; u_a += w2;
    }
  };
  // piece d: activation planes first (pl, i), then weight planes; past the last tile through zero-record descriptors
  auto issue_piece = [&](int d, int buf, int kt, bool live) {
    char* base = smem + buf * BUF + wave_u * 1024;
    if (d < NPL * A_ITP) {
      const int pl = d / A_ITP, i = d % A_ITP;
      bufglds16(make_rsrc(Ab, live ? (unsigned)(NPL - 1) * a_plane + p.a_bytes : 0u), avo[i], aso + (unsigned)pl * a_plane,
                base + pl * PLA + i * (RPPS * 64));
    } else {
      const int e = d - NPL * A_ITP, pl = e / B_ITP, i = e % B_ITP;
      bufglds16(make_rsrc(Bb, live ? p.b_bytes : 0u), kokb ? boff[i] : kOOB, (unsigned)kt * 64u + (unsigned)pl * b_plane,
                base + NPL * PLA + pl * PLB + i * (RPPS * 64));
    }
  };

  f32x4_v acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b) acc[a][b] = f32x4_v{0.f, 0.f, 0.f, 0.f};

  if (ktiles > 0) {
    prep_tile(0);
#pragma unroll
    for (int d = 0; d < ND; ++d) issue_piece(d, 0, 0, true);
  }
  constexpr int NTERM = NPL == 3 ? 6 : 3;
  constexpr int TA[6] = {0, 1, NPL == 3 ? 2 : 0, 0, 1, 0};
  constexpr int TB[6] = {0, 0, NPL == 3 ? 0 : 1, 1, 1, 2};
  constexpr int PER = FN * FM, NMF = NTERM * PER;
  int aro[FM], bro[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) { const int row = wm * WM + i * 16 + l15; aro[i] = row * 64 + ((g ^ key64(row)) << 4); }
#pragma unroll
  for (int j = 0; j < FN; ++j) { const int row = wn * WN + j * 16 + l15; bro[j] = row * 64 + ((g ^ key64(row)) << 4); }
  for (int kt = 0; kt < ktiles; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    prep_tile(kt + 1);
    const int nbuf = (kt + 1) & 1;
    const bool live = kt + 1 < ktiles;
    const char* xa = smem + (kt & 1) * BUF;
    const char* wb = xa + NPL * PLA;
    bf16x8_v wf[NPL][FN], xf[NPL][FM];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
      for (int j = 0; j < FN; ++j) wf[pl][j] = *reinterpret_cast<const bf16x8_v*>(wb + pl * PLB + bro[j]);
#pragma unroll
      for (int i = 0; i < FM; ++i) xf[pl][i] = *reinterpret_cast<const bf16x8_v*>(xa + pl * PLA + aro[i]);
    }
    __builtin_amdgcn_sched_barrier(0);
    static_for<NMF>([&](auto nc) {
      constexpr int n = decltype(nc)::value;
      constexpr int ts = n / PER, j = (n % PER) / FM, i = n % FM;
      acc[j][i] = mma_bf16(wf[TA[ts]][j], xf[TB[ts]][i], acc[j][i]);
      static_for<ND>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        if constexpr (((d + 1) * NMF) / (ND + 1) == n + 1) issue_piece(d, nbuf, kt + 1, live);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  __syncthreads();
  nt_epilogue<BN, FN, FM>(p, acc, smem, m0, n0, z, tid, wm, wn, l15, g);
}

// =============================================================================================
// TN: O[pp][qq] = sum_m P[m][pp] * Xg[m][qq], fp32 operands split (2 terms) on their way into LDS
// =============================================================================================
template <int BP, int BQ, bool IDENT, bool PACKW, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_tn_sp_kernel(const GP p) {
  // NW = 4: waves 2 (p) x 2 (q).  NW = 8: 2 x 4 on the same tile -- twice the wavefronts per CU on the same LDS (the
  // staging phase of one wave overlaps the MFMAs of the others; SQ_WAIT_ANY was 44 % of the wave time with 4 waves)
  typedef float T;
  constexpr int NTHR = 64 * NW, NWQ = NW / 2;
  constexpr int EPC = 4, BK = 32;
  constexpr int NPB = BP / EPC * 8, NQB = BQ / EPC * 8;    // 4 x 4 staging blocks per tile
  constexpr int ITER = (NPB + NQB + NTHR - 1) / NTHR;
  constexpr int WP = BP / 2, WQ = BQ / NWQ;
  static_assert(WQ >= 16, "tile too narrow for this many waves");
  constexpr int FP = WP / 16, FQ = WQ / 16;
  constexpr int BUF = (BP + BQ) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wp = wave / NWQ, wq = wave % NWQ;
  const int l15 = lane & 15, g = lane >> 4;

  const int nwg = p.tiles_m * p.tiles_n;
  int bid, split;
  if (p.splits > 1 && (p.splits & 7) == 0) {
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    bid = slot % nwg;
    split = (slot / nwg) * 8 + xcd;
  } else {
    bid = xcd_remap(blockIdx.x, nwg);
    split = blockIdx.y;
  }
  const int tile_p = bid / p.tiles_n, tile_q = bid - tile_p * p.tiles_n;
  const int p0 = tile_p * BP, q0 = tile_q * BQ;
  const int z = blockIdx.z;
  const char* Pb = p.P + (long long)z * p.p_bs * 4;
  const char* Ab = p.A + (long long)z * p.a_bs * 4;

  const int kbeg = split * p.kper;
  const int kend = min(p.M, kbeg + p.kper);
  const int ktiles = (kend - kbeg + BK - 1) / BK;

  // Staging blocks: 4 positions x 4 channels (one 16-byte load per position).  Block id -> (operand, channel group
  // blk_r, position group blk_k); the operand of a block is uniform per WAVE (NPB, NQB are multiples of 128).
  // All loads are buffer loads with 32-bit offsets: the per-lane part is loop invariant for the gradient operand and
  // for plain rows (the k-tile advance is the scalar offset, rows past the end of this split's slab are zero-filled by
  // the descriptor's range check), so a k-tile costs no address arithmetic and no select on the data -- with 64-bit
  // global loads hipcc recycled the address temporaries of one block into the destination registers of the other and
  // waited for the loads in between (s_waitcnt vmcnt(3..1) at the top of every k-tile: 44 % of the wave time parked).
  int blk_kind[ITER], blk_r[ITER], blk_k[ITER];
  TapC qtap[ITER];
  unsigned bvo[ITER][EPC];                          // loop-invariant byte offsets of the block's 4 positions (P / plain Q)
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    int id = tid + it * NTHR;
    if (id < NPB) { blk_kind[it] = 0; blk_r[it] = id >> 3; blk_k[it] = id & 7; }
    else if (id < NPB + NQB) { id -= NPB; blk_kind[it] = 1; blk_r[it] = id >> 3; blk_k[it] = id & 7; }
    else { blk_kind[it] = 2; blk_r[it] = 0; blk_k[it] = 0; }
    unsigned b0 = kOOB, ldb4 = 0;
    if (blk_kind[it] == 0) {
      const int pc = p0 + blk_r[it] * EPC;
      if (pc < p.Ncols) { b0 = (unsigned)((kbeg + blk_k[it] * EPC) * p.ldp + pc) * 4u; ldb4 = (unsigned)p.ldp * 4u; }
    } else if (blk_kind[it] == 1) {
      const int kc = (q0 + blk_r[it] * EPC) / EPC;
      if (IDENT) {
        qtap[it].ok = kc * EPC < p.K; qtap[it].a = qtap[it].b = qtap[it].c = 0; qtap[it].ci = 0;
        if (qtap[it].ok) { b0 = (unsigned)((kbeg + blk_k[it] * EPC) * p.lda + q0 + blk_r[it] * EPC) * 4u; ldb4 = (unsigned)p.lda * 4u; }
      } else {
        qtap[it] = decode_tap<T, PACKW>(p, kc);
      }
    }
#pragma unroll
    for (int j = 0; j < EPC; ++j) bvo[it][j] = b0 == kOOB ? kOOB : b0 + (unsigned)j * ldb4;
  }
  const __amdgpu_buffer_rsrc_t rsP = make_rsrc(Pb, (unsigned)kend * (unsigned)p.ldp * 4u);
  const __amdgpu_buffer_rsrc_t rsQ = make_rsrc(Ab, IDENT ? (unsigned)kend * (unsigned)p.lda * 4u : p.a_bytes);
  // gather cursor of a Q block (generic path): output position of the block's first row in the next k-tile
  RowC qpos[ITER];
  RowC jump;
  if (!IDENT) {
    jump = decode_row(p, BK - EPC);                 // after a tile the cursor already moved EPC positions
#pragma unroll
    for (int it = 0; it < ITER; ++it)
      if (blk_kind[it] == 1) qpos[it] = decode_row(p, min(kbeg + blk_k[it] * EPC, p.M - 1));
  }
  // branch-free cursor arithmetic (selects only: the k-loop must not grow per-lane control flow around the loads)
  auto pos_next = [&](RowC& r) {
    ++r.w;
    const int c0 = r.w == p.Wr; r.w = c0 ? 0 : r.w; r.h += c0;
    const int c1 = r.h == p.Hr; r.h = c1 ? 0 : r.h; r.t += c1;
    const int c2 = r.t == p.Tr; r.t = c2 ? 0 : r.t; r.n += c2;
  };
  auto pos_jump = [&](RowC& r, const RowC& d) {
    r.w += d.w; const int c0 = r.w >= p.Wr; r.w -= c0 ? p.Wr : 0; r.h += c0;
    r.h += d.h; const int c1 = r.h >= p.Hr; r.h -= c1 ? p.Hr : 0; r.t += c1;
    r.t += d.t; const int c2 = r.t >= p.Tr; r.t -= c2 ? p.Tr : 0; r.n += c2;
    r.n += d.n;
  };
  auto gather_off = [&](const RowC& r, const TapC& tp, bool live) -> unsigned {
    const int ts = r.t * p.st - p.pt + tp.a * p.dt;
    const int hs = r.h * p.sh - p.ph + tp.b * p.dh;
    bool ok = live && tp.ok && (unsigned)ts < (unsigned)p.Ts && (unsigned)hs < (unsigned)p.Hs;
    const int rowpix = ((r.n * p.Ts + ts) * p.Hs + hs) * p.Ws;
    if (PACKW) {
      const int w0 = r.w * p.sw - p.pw + tp.c;       // W-padded stem input: always inside the row
      return ok ? (unsigned)(rowpix + w0) * 16u : kOOB;
    }
    const int ws = r.w * p.sw - p.pw + tp.c * p.dw;
    ok = ok && (unsigned)ws < (unsigned)p.Ws;
    return ok ? (unsigned)((rowpix + ws) * p.lda + tp.ci) * 4u : kOOB;
  };

  uint4 stg[ITER][EPC];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int kind = __builtin_amdgcn_readfirstlane(blk_kind[it]);      // wave uniform
      if (kind == 2) continue;
      if (kind == 0 || IDENT) {
        const unsigned ld = kind == 0 ? (unsigned)p.ldp : (unsigned)p.lda;
        const unsigned soff = (unsigned)(kt * BK) * ld * 4u;
#pragma unroll
        for (int j = 0; j < EPC; ++j) {
          if (kind == 0) stg[it][j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsP, (int)bvo[it][j], (int)soff, 0));
          else stg[it][j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsQ, (int)bvo[it][j], (int)soff, 0));
        }
      } else {
        const int kk = kbeg + kt * BK + blk_k[it] * EPC;
#pragma unroll
        for (int j = 0; j < EPC; ++j) {
          const unsigned vo = gather_off(qpos[it], qtap[it], kk + j < kend);
          stg[it][j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsQ, (int)vo, 0, 0));
          pos_next(qpos[it]);
        }
        pos_jump(qpos[it], jump);
      }
    }
  };
  // block (4 positions x 4 channels, stg[j] = the 4 channels of position j) -> per channel the 4 positions as
  // bf16 h (8 bytes) and m (8 bytes): row = channel, h in the first 64 bytes, m in the second
  auto store_tile = [&](int buf) {
    char* pt = smem + buf * BUF;
    char* qt = pt + BP * 128;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      if (blk_kind[it] == 2) continue;
      char* dst = blk_kind[it] == 0 ? pt : qt;
#pragma unroll
      for (int c = 0; c < EPC; ++c) {
        float a0 = __uint_as_float(word_sel(stg[it][0], c)), a1 = __uint_as_float(word_sel(stg[it][1], c));
        float a2 = __uint_as_float(word_sel(stg[it][2], c)), a3 = __uint_as_float(word_sel(stg[it][3], c));
        const uint32_t h01 = peel(a0, a1), h23 = peel(a2, a3);
        const uint32_t m01 = pack_bf2(a0, a1), m23 = pack_bf2(a2, a3);
        const int row = blk_r[it] * EPC + c;
        const int ch = blk_k[it] >> 1, half = (blk_k[it] & 1) << 3;
        char* rowp = dst + row * 128;
        *reinterpret_cast<uint2*>(rowp + ((ch ^ (row & 7)) << 4) + half) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(rowp + (((4 + ch) ^ (row & 7)) << 4) + half) = make_uint2(m01, m23);
      }
    }
  };

  f32x4_v acc[FQ][FP];
#pragma unroll
  for (int a = 0; a < FQ; ++a)
#pragma unroll
    for (int b = 0; b < FP; ++b) acc[a][b] = f32x4_v{0.f, 0.f, 0.f, 0.f};

  if (ktiles > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();
  for (int kt = 0; kt < ktiles; ++kt) {
    const bool more = kt + 1 < ktiles;
    if (more) load_tile(kt + 1);
    const char* pt = smem + (kt & 1) * BUF;
    const char* qt = pt + BP * 128;
    bf16x8_v pf[2][FP], qf[2][FQ];
#pragma unroll
    for (int i = 0; i < FP; ++i) {
      const int row = wp * WP + i * 16 + l15;
      pf[0][i] = *reinterpret_cast<const bf16x8_v*>(pt + row * 128 + ((g ^ (row & 7)) << 4));
      pf[1][i] = *reinterpret_cast<const bf16x8_v*>(pt + row * 128 + (((4 + g) ^ (row & 7)) << 4));
    }
#pragma unroll
    for (int j = 0; j < FQ; ++j) {
      const int row = wq * WQ + j * 16 + l15;
      qf[0][j] = *reinterpret_cast<const bf16x8_v*>(qt + row * 128 + ((g ^ (row & 7)) << 4));
      qf[1][j] = *reinterpret_cast<const bf16x8_v*>(qt + row * 128 + (((4 + g) ^ (row & 7)) << 4));
    }
#define VLFB_SP_TERM(a, b)                                                        \
    _Pragma("unroll") for (int j = 0; j < FQ; ++j)                                \
      _Pragma("unroll") for (int i = 0; i < FP; ++i)                              \
        acc[j][i] = mma_bf16(qf[a][j], pf[b][i], acc[j][i]);
    VLFB_SP_TERM(1, 0) VLFB_SP_TERM(0, 1) VLFB_SP_TERM(0, 0)
#undef VLFB_SP_TERM
    if (more) store_tile((kt + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue (as gemm_tn_kernel): lane holds qq = qb + 0..3 for output row pp ----
  const bool vec_ok = (p.ldo & 3) == 0;
  const bool to_ws = p.splits > 1;
#pragma unroll
  for (int i = 0; i < FP; ++i) {
    const int pp = p0 + wp * WP + i * 16 + l15;
    if (pp >= p.Ncols) continue;
    const float rs = (to_ws || !p.rowscale) ? 1.f : p.rowscale[pp];
#pragma unroll
    for (int j = 0; j < FQ; ++j) {
      const int qb = q0 + wq * WQ + j * 16 + g * 4;
      if (qb >= p.K) continue;
      const int cnt = (p.K - qb) < 4 ? (p.K - qb) : 4;
      const long long idx = (long long)pp * p.ldo + qb;
      float v[4];
      if (to_ws) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[j][i][r];
        store4<float>(reinterpret_cast<char*>(p.ws), (long long)split * ((long long)p.Ncols * p.ldo) + idx, v, cnt, vec_ok);
      } else {
        char* Ob = p.O + (long long)z * p.o_bs * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = acc[j][i][r] * p.alpha * rs;
          if (p.accumulate && r < cnt) x += ld_elem<float>(Ob, idx + r);
          v[r] = x;
        }
        store4<float>(Ob, idx, v, cnt, vec_ok);
      }
    }
  }
}

template <typename K>
int launch_sp(K kernel, dim3 grid, size_t lds, const GP& gp, hipStream_t s, int threads = 256) {
  static bool configured = false;     // per template instance
  if (!configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    configured = true;
  }
  hipLaunchKernelGGL(kernel, grid, dim3(threads), lds, s, gp);
  return check_launch("split-bf16 conv kernel");
}

template <int NPL, int BN, int NWN>
int launch_nt_sp_waves(const GP& gp, int kind, bool ut, dim3 grid, size_t lds, hipStream_t s) {
  switch (kind) {
    case 0: return launch_sp(gemm_nt_sp_kernel<NPL, BN, true, false, false, false, false, NWN>, grid, lds, gp, s);
    case 1: return ut ? launch_sp(gemm_nt_sp_kernel<NPL, BN, false, false, false, true, false, NWN>, grid, lds, gp, s)
                      : launch_sp(gemm_nt_sp_kernel<NPL, BN, false, false, false, false, false, NWN>, grid, lds, gp, s);
    case 2:
      if constexpr (NPL == 2) {
        if (gp.s2) return launch_sp(gemm_nt_sp_kernel<NPL, BN, false, true, false, true, true, NWN>, grid, lds, gp, s);
      }
      if (gp.s2) return set_error(VLFB_ERR_UNSUPPORTED, "conv: class-major strided DGRAD exists for three-term products only");
      return ut ? launch_sp(gemm_nt_sp_kernel<NPL, BN, false, true, false, true, false, NWN>, grid, lds, gp, s)
                : launch_sp(gemm_nt_sp_kernel<NPL, BN, false, true, false, false, false, NWN>, grid, lds, gp, s);
    default: return launch_sp(gemm_nt_sp_kernel<NPL, BN, false, false, true, false, false, NWN>, grid, lds, gp, s);
  }
}
template <int NPL, int BN>
int launch_nt_sp_shape(const GP& gp, int kind, bool ut, dim3 grid, size_t lds, hipStream_t s) {
  // Three-term products (NPL = 2): 4 x 1 waves of 32 x BN -- measured against 2 x 2 waves of 64 x BN/2 on the benchmarked
  // `mix` step (same box, alternating runs): every launch family faster or equal (64-column res2 convs -16 %, gathered 3x3 /
  // 3x1x1 -6..7 %, plain 1x1x1 rows 0..-4 %), forward NT time 19.26 -> 18.48 ms, 233.0 -> 236.0 clips/s; bit-identical (the
  // accumulation order of an output element does not depend on which wave owns it).  Six-term products keep 2 x 2: a
  // 32 x 128 wave tile would hold three weight planes x eight fragments in registers.
  return launch_nt_sp_waves<NPL, BN, NPL == 2 ? 1 : 2>(gp, kind, ut, grid, lds, s);
}

// ---- fp32 -> bf16 term planes (attention operands; optionally transposed per batch element) -----------------
template <int NPL>
__global__ void split_planes_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long long n, long long plane) {
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < n;
       i += (long long)gridDim.x * blockDim.x * 2) {
    float a = src[i], b = i + 1 < n ? src[i + 1] : 0.f;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      const uint32_t w = pl + 1 < NPL ? peel(a, b) : pack_bf2(a, b);
      if (i + 1 < n) *reinterpret_cast<uint32_t*>(dst + pl * plane + i) = w;
      else dst[pl * plane + i] = (bf16_t)(w & 0xffffu);
    }
  }
}
template <int NPL>
__global__ void split_planes_tr_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long long rows,
                                       long long cols, long long plane) {
  __shared__ float tile[32][33];
  const long long b = blockIdx.z;
  const float* s = src + b * rows * cols;
  bf16_t* d = dst + b * rows * cols;
  const long long c0 = (long long)blockIdx.x * 32, r0 = (long long)blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const long long r = r0 + j, c = c0 + threadIdx.x;
    if (r < rows && c < cols) tile[j][threadIdx.x] = s[r * cols + c];
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const long long c = c0 + j, r = r0 + threadIdx.x;
    if (r < rows && c < cols) {
      float a = tile[threadIdx.x][j], z = 0.f;
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        const uint32_t w = pl + 1 < NPL ? peel(a, z) : pack_bf2(a, z);
        d[pl * plane + c * rows + r] = (bf16_t)(w & 0xffffu);
      }
    }
  }
}

}  // namespace

int launch_nt_split(const GP& gp, int npl, int bn, int kind, bool ut, dim3 grid, size_t lds, hipStream_t s) {
  if (npl == 3) return bn == 128 ? launch_nt_sp_shape<3, 128>(gp, kind, ut, grid, lds, s) : launch_nt_sp_shape<3, 64>(gp, kind, ut, grid, lds, s);
  return bn == 128 ? launch_nt_sp_shape<2, 128>(gp, kind, ut, grid, lds, s) : launch_nt_sp_shape<2, 64>(gp, kind, ut, grid, lds, s);
}

template <int NPL, int BN>
int launch_nt_pl_shape(const GP& gp, int kind, dim3 grid, size_t lds, hipStream_t s) {
  if (kind == 0) return launch_sp(gemm_nt_pl_kernel<NPL, BN, true, false>, grid, lds, gp, s);
  if (kind == 1) return launch_sp(gemm_nt_pl_kernel<NPL, BN, false, false>, grid, lds, gp, s);
  return launch_sp(gemm_nt_pl_kernel<NPL, BN, false, true>, grid, lds, gp, s);
}

int launch_nt_planes(const GP& gp, int npl, int bn, int kind, dim3 grid, size_t lds, hipStream_t s) {
  if (kind == 3) return set_error(VLFB_ERR_UNSUPPORTED, "conv: the packed stem takes its activation operand as fp32");
  if (npl == 3) return bn == 128 ? launch_nt_pl_shape<3, 128>(gp, kind, grid, lds, s) : launch_nt_pl_shape<3, 64>(gp, kind, grid, lds, s);
  return bn == 128 ? launch_nt_pl_shape<2, 128>(gp, kind, grid, lds, s) : launch_nt_pl_shape<2, 64>(gp, kind, grid, lds, s);
}

int launch_tn_split(const GP& gp, int bp, int bq, bool ident, bool packw, dim3 grid, size_t lds, hipStream_t s) {
#define VLFB_TN_SP(BP, BQ)                                                                                     \
  do {                                                                                                         \
    constexpr int NW8 = ((BP) >= 128 && (BQ) >= 128) ? 8 : 4;  /* (64-row tiles measured slower with 8 waves ...) */                                                                  \
    constexpr int NWS = (BQ) >= 128 ? 8 : 4;                   /* (... except the packed stem: 3.55 -> 2.8 ms) */  \
    if (ident) return launch_sp(gemm_tn_sp_kernel<BP, BQ, true, false, NW8>, grid, lds, gp, s, 64 * NW8);      \
    if (packw) return launch_sp(gemm_tn_sp_kernel<BP, BQ, false, true, NWS>, grid, lds, gp, s, 64 * NWS);      \
    return launch_sp(gemm_tn_sp_kernel<BP, BQ, false, false, NW8>, grid, lds, gp, s, 64 * NW8);                \
  } while (0)
  if (bp == 128 && bq == 128) VLFB_TN_SP(128, 128);
  if (bp == 64 && bq == 128) VLFB_TN_SP(64, 128);
  if (bp == 128 && bq == 64) VLFB_TN_SP(128, 64);
  VLFB_TN_SP(64, 64);
#undef VLFB_TN_SP
}

}  // namespace vlfb

using namespace vlfb;

extern "C" int vlfb_split_planes(const float* src, void* dst, int nplanes, int64_t batch, int64_t rows, int64_t cols,
                                 int transpose, vlfb_stream_t stream) {
  VLFB_REQUIRE(src && dst && batch > 0 && rows > 0 && cols > 0, "split_planes: bad args");
  VLFB_REQUIRE(nplanes == 2 || nplanes == 3, "split_planes: nplanes must be 2 or 3");
  const long long n = (long long)batch * rows * cols;
  hipStream_t s = (hipStream_t)stream;
  if (!transpose) {
    VLFB_REQUIRE(n % 2 == 0, "split_planes: an even element count is required");
    const int grid = grid_for(n / 2, 256);
    if (nplanes == 3) hipLaunchKernelGGL(split_planes_kernel<3>, dim3(grid), dim3(256), 0, s, src, (bf16_t*)dst, n, n);
    else hipLaunchKernelGGL(split_planes_kernel<2>, dim3(grid), dim3(256), 0, s, src, (bf16_t*)dst, n, n);
  } else {
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32), (unsigned)batch);
    if (nplanes == 3) hipLaunchKernelGGL(split_planes_tr_kernel<3>, grid, dim3(32, 8), 0, s, src, (bf16_t*)dst, (long long)rows, (long long)cols, n);
    else hipLaunchKernelGGL(split_planes_tr_kernel<2>, grid, dim3(32, 8), 0, s, src, (bf16_t*)dst, (long long)rows, (long long)cols, n);
  }
  return check_launch("split_planes");
}
