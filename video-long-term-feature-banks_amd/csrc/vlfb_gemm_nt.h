// The 128-row implicit-GEMM NT kernel (FPROP / DGRAD / batched NT products), shared by vlfb_gemm.hip (the 16-bit and
// exact-fp32 instances) and vlfb_gemm_pair.hip (the two-plane fp16 instances of the "mix" forward).
#pragma once
#include "vlfb_gemm_common.h"

namespace vlfb {
namespace {

// =============================================================================================
// NT kernel: O[m][n] = sum_k X[m][k] * W[n][k]   (X gathered: FPROP / DGRAD / identity)
// =============================================================================================
template <typename T, typename OutT, int BM, int BN, bool IDENT, bool DGRAD, bool PACKW, int RB, bool PRE, int NW,
          int ST = 2, bool UT = false, bool PAIR = false, bool W2I = false>
__global__ __launch_bounds__(64 * NW) void gemm_nt_kernel(const GP p) {
  // W2I (16-bit unit-stride DGRAD with TWO-TERM weights, vlfb_conv_desc.math = VLFB_MATH_F16W2; the backward of the "mix"
  // path): the weight rows are [tap][Cs / 64][term][64] (vlfb_weight_prep* VLFB_MIX_W2I), i.e. the k-tiles of the weight
  // operand alternate Wh, Wl of the SAME 64 channels of the same tap, and both are contracted with ONE gradient tile: the
  // A tile of a pair is fetched once (into its own double buffer, indexed by the pair), its fragments are read from LDS
  // once and stay in registers for the second term.  Against the doubled-tap form of the same product (kt' = 2 kt,
  // dt = 0: every A tile fetched and read twice) that is 3/4 of the DMA bytes and 2/3 of the LDS fragment reads per MFMA.
  static_assert(!W2I || (sizeof(T) == 2 && RB == 128 && ST == 2 && UT && DGRAD && !PAIR), "W2I: 16-bit unit-stride DGRAD on the tap cursor");
  // PAIR (fp16 only; the forward of the "mix" path, vlfb_conv_desc.math = VLFB_MATH_F16X3): both operands are TWO fp16
  // planes, value = hi + lo (hi = fp16(v), lo = fp16(v - hi): ~22 significant bits; fp16 MFMA operands keep subnormals,
  // probed on MI355X), GP::a_ps / b_ps elements apart.  A 128-byte LDS row holds 32 k of the hi plane (chunks 0-3) and
  // the SAME 32 k of the lo plane (chunks 4-7); the two k-steps of a row become the three products lo.hi + hi.lo + hi.hi
  // (the lo.lo term is below 2^-22).  Nothing is converted in the loop: the operands arrive pre-split from the producing
  // epilogue (O = hi, O2 = lo) and from vlfb_weight_prep.  Plain rows or the scalar tap cursor (taps of whole 32-k tiles).
  static_assert(!PAIR || (sizeof(T) == 2 && RB == 128 && ST == 2 && (IDENT || UT) && !PACKW && !DGRAD), "PAIR: fp16 planes, plain rows or UT FPROP");
  // UT ("uniform tap", gathered convs whose taps span whole k-tiles: Cs * sizeof(T) % RB == 0, and for
  // DGRAD unit stride): the filter tap of a k-tile is the same for the whole workgroup, so it is a
  // scalar cursor advanced once per tile, the source pixel is  row base + scalar tap delta, and only
  // the three padding compares stay per lane.  The generic path decodes the tap per lane and rebuilds
  // the address from (n, t, h, w) every tile: measured 5 VALU instructions per MFMA on the 3x3 layers
  // (SQ_INSTS_VALU 43.1 M vs SQ_INSTS_MFMA 7.2 M), more issue time than the MFMAs themselves.
  static_assert(!UT || (!IDENT && !PACKW), "UT is for gathered, unpacked operands");
  // ST = LDS stages of the k-loop ring (tiles in flight = ST - 1, counted vmcnt on the oldest).
  // Shipped instances use ST = 2.  Measured on MI355X: ST = 5 with 64-byte tile rows (four 16 KiB
  // tiles in flight on the same 80 KiB) is 15-20 % SLOWER on the long-K res5 layers (561 vs 688
  // TFLOP/s) -- the loop is bound by LDS-read + DMA-issue + MFMA phases that do not overlap inside a
  // barrier-synchronised workgroup, not by DMA latency, and halving BK doubles the barriers.
  // NW = 4: waves 2 (m) x 2 (n), 64 x (BN/2) each.  NW = 8: waves 2 x 4, 64 x (BN/4) each -- twice the
  // wavefronts per CU on the same LDS footprint (more latency hiding, 1.5x the LDS fragment reads).
  constexpr int NTHR = 64 * NW;
  constexpr int NWN = NW / 2;              // waves along n
  constexpr int EPC = Elem<T>::EPC;
  constexpr int CPRW = RB / 16;            // 16-byte chunks per tile row
  constexpr int RPPS = NTHR / CPRW;        // tile rows staged per pass
  constexpr int A_IT = BM / RPPS, B_IT = BN / RPPS;
  constexpr int WM = BM / 2, WN = BN / NWN;
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int BUF = (BM + BN) * RB;
  constexpr int KSTEPS = sizeof(T) == 4 ? 1 : RB / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
  const int l15 = lane & 15, g = lane >> 4;

  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = xcd_remap(blockIdx.x, nwg);
  // n-tiles fastest: neighbouring workgroups reuse the same activation rows from L2
  const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
  int m0 = tile_m * BM;
  const int n0 = tile_n * BN;
  const int z = blockIdx.z;
  // strided DGRAD, class-major rows (GP::s2): this tile's parity class and its first row inside the class
  constexpr bool S2C = DGRAD && !UT && !IDENT && !PACKW;       // the only instances a strided DGRAD can reach
  bool s2 = false;
  int s2_ph = 0, s2_pw = 0;
  if constexpr (S2C) {
    s2 = p.s2 != 0;
    if (s2) {
      const int cls = tile_m / p.s2_tpc;
      m0 = (tile_m - cls * p.s2_tpc) * BM;
      s2_ph = cls >> 1; s2_pw = cls & 1;
    }
  }
  const int mrows = s2 ? p.s2_mq : p.M;                         // rows of the enumeration this tile walks
  // row of the enumeration -> (n, t, h, w) and the linear position (= output row)
  auto row_coords = [&](int m) {
    if constexpr (S2C) {
      if (s2) {
        const int w2n = p.Wr >> 1, h2n = p.Hr >> 1;
        RowC r;
        const int w2 = m % w2n; int q = m / w2n;
        const int h2 = q % h2n; q /= h2n;
        r.t = q % p.Tr; r.n = q / p.Tr;
        r.h = 2 * h2 + s2_ph; r.w = 2 * w2 + s2_pw;
        return r;
      }
    }
    return decode_row(p, m);
  };
  auto row_pos = [&](int m) -> long long {
    if constexpr (S2C) {
      if (s2) {
        const RowC r = row_coords(m);
        return (long long)((r.n * p.Tr + r.t) * p.Hr + r.h) * p.Wr + r.w;
      }
    }
    return m;
  };

  const char* Ab = p.A + (long long)z * p.a_bs * (long long)sizeof(T);
  const char* Bb = p.B + (long long)z * p.b_bs * (long long)sizeof(T);

  // Staging: thread t owns LDS slot (row = t / CPRW + RPPS * i, 16-byte slot t % CPRW) of both operand
  // tiles.  The global->LDS copy is asynchronous DMA (buffer_load_dwordx4 ... lds, no VGPR round
  // trip); a wave's 64 slots are 1 KiB contiguous, and because the LDS image is XOR-swizzled the lane
  // fetches global chunk (slot ^ key(row)).
  const int cc = tid % CPRW;
  const int r0 = tid / CPRW;
  const int ccg = cc ^ swz_key<RB>(r0);   // global 16-byte chunk column fetched by this lane
  // PAIR: chunk ccg of a row = 16 bytes at k-offset (ccg & 3) * 8 of plane ccg >> 2
  const unsigned a_cb = PAIR ? (unsigned)(ccg & 3) * 16u + (unsigned)(ccg >> 2) * (unsigned)p.a_ps * 2u : (unsigned)(ccg * EPC) * (unsigned)sizeof(T);
  const unsigned b_cb = PAIR ? (unsigned)(ccg & 3) * 16u + (unsigned)(ccg >> 2) * (unsigned)p.b_ps * 2u : (unsigned)(ccg * EPC) * (unsigned)sizeof(T);
  // bytes a k-tile advances along a row: planes layout = 32 k of each plane (64 bytes); interleaved layout (GP::pair_il: rows of
  // [32 k of hi | the same 32 k of lo] groups, a_ps = b_ps = 32) = one whole 128-byte group
  const int KTB = PAIR ? (p.pair_il ? RB : RB / 2) : RB;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  RowC arow[A_IT];
  bool aok[A_IT];
  int upix[UT ? A_IT : 1];                 // UT: pixel index of the row at tap (0,0,0) (may be "negative")
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    int m = m0 + r0 + RPPS * i;
    aok[i] = m < mrows;
    if (!IDENT) arow[i] = row_coords(aok[i] ? m : 0);
    if (UT) {
      RowC& r = arow[i];                   // (t, h, w) become the tap-(0,0,0) source coordinates
      if (!DGRAD) { r.t = r.t * p.st - p.pt; r.h = r.h * p.sh - p.ph; r.w = r.w * p.sw - p.pw; }
      else { r.t += p.pt; r.h += p.ph; r.w += p.pw; }
      upix[i] = ((r.n * p.Ts + r.t) * p.Hs + r.h) * p.Ws + r.w;
    }
  }
  // byte offset of (row, chunk ccg) at k-tile 0 for the weight rows (and, IDENT / UT, the
  // activation rows) of this lane; kOOB when the row does not exist
  unsigned boff[B_IT], aoff[(IDENT || UT) ? A_IT : 1];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int n = n0 + r0 + RPPS * i;
    boff[i] = n < p.Ncols ? (unsigned)(n * p.ldb) * (unsigned)sizeof(T) + b_cb : kOOB;
  }
  if (IDENT || UT) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int m = m0 + r0 + RPPS * i;
      if (IDENT) aoff[i] = aok[i] ? (unsigned)(m * p.lda) * (unsigned)sizeof(T) + a_cb : kOOB;
      else aoff[i] = (unsigned)(upix[i] * p.lda) * (unsigned)sizeof(T) + a_cb;   // wraps for padding rows
    }
  }
  const auto rsA = make_rsrc(Ab, p.a_bytes);
  const auto rsB = make_rsrc(Bb, p.b_bytes);
  // UT: scalar tap cursor of the NEXT tile to be fetched (tiles are fetched in order 0, 1, 2, ...)
  int u_a = 0, u_b = 0, u_c = 0, u_ci = 0;

  int ktiles = (p.K * (int)sizeof(T) + (PAIR ? RB / 2 : RB) - 1) / (PAIR ? RB / 2 : RB);
  if constexpr (W2I) ktiles *= 2;          // (Wh, Wl) per activation tile
  // class-major strided DGRAD: only the taps with (h + ph - b) and (w + pw - c) even exist for this tile's class
  // (b = b0, b0 + 2, ..; c likewise).  The k-loop walks those taps in ascending order -- the order of the full
  // walk with the structurally-zero taps left out -- through a scalar cursor, one k-tile per load_tile call.
  int s2_kpt = 1, s2_b0 = 0, s2_c0 = 0, s2_nb = 0, s2_nc = 0;
  int s2_a = 0, s2_ib = 0, s2_ic = 0, s2_in = 0;               // cursor: tap (a, b0 + 2 ib, c0 + 2 ic), k-tile inside the tap
  if constexpr (S2C) {
    if (s2) {
      s2_kpt = p.Cs * (int)sizeof(T) / RB;
      s2_b0 = (s2_ph + p.ph) & 1; s2_c0 = (s2_pw + p.pw) & 1;
      s2_nb = p.kh > s2_b0 ? (p.kh - s2_b0 + 1) >> 1 : 0;
      s2_nc = p.kw > s2_c0 ? (p.kw - s2_c0 + 1) >> 1 : 0;
      ktiles = p.kt * s2_nb * s2_nc * s2_kpt;
    }
  }
  auto s2_next_kt = [&]() {                                     // k-tile of the full walk the cursor stands on; advance
    const int tap = (s2_a * p.kh + s2_b0 + 2 * s2_ib) * p.kw + s2_c0 + 2 * s2_ic;
    const int kt = tap * s2_kpt + s2_in;
    if (++s2_in == s2_kpt) {
      s2_in = 0;
      if (++s2_ic == s2_nc) { s2_ic = 0; if (++s2_ib == s2_nb) { s2_ib = 0; ++s2_a; } }
    }
    return kt;
  };

  auto load_tile = [&](int kt_seq, int buf) {
    int kt = kt_seq;
    if constexpr (S2C) {
      if (s2) kt = s2_next_kt();                                // (calls come in sequence 0, 1, 2, ...)
    }
    const int kc = PAIR ? kt * (CPRW / 2) + (ccg & 3) : kt * CPRW + ccg;
    TapC tap;
    if (IDENT || UT) { tap.ok = kc * EPC < p.K; tap.a = tap.b = tap.c = tap.ci = 0; }
    else tap = decode_tap<T, PACKW>(p, kc);
    const bool kok = W2I || kc * EPC < p.K;      // (W2I: taps of whole 64-channel runs, every k-tile is full)
    {
      // W2I: the activation tile of pair kt_seq / 2 has its own double buffer (the A halves of the two ring slots)
      char* xa = smem + (W2I ? (kt_seq >> 1) & 1 : buf) * BUF + wave_u * 1024;
      char* wb = smem + buf * BUF + BM * RB + wave_u * 1024;
      // the last k-tile may end inside the row (K * sizeof(T) % RB != 0): chunks past K are padding
      const unsigned kbyte = (unsigned)kt * KTB;
      if (UT) {
        const int sgn = DGRAD ? -1 : 1;
        const int da = sgn * u_a * p.dt, db = sgn * u_b * p.dh, dc = sgn * u_c * p.dw;   // scalar
        const unsigned dbyte = (unsigned)(((da * p.Hs + db) * p.Ws + dc) * p.lda + ((PAIR && p.pair_il) ? 2 * u_ci : u_ci)) * (unsigned)sizeof(T);
        if (!W2I || !(kt_seq & 1)) {
#pragma unroll
          for (int i = 0; i < A_IT; ++i) {
            const bool ok = aok[i] && (unsigned)(arow[i].t + da) < (unsigned)p.Ts &&
                            (unsigned)(arow[i].h + db) < (unsigned)p.Hs && (unsigned)(arow[i].w + dc) < (unsigned)p.Ws;
            bufglds16(rsA, ok ? aoff[i] + dbyte : kOOB, 0, xa + i * (RPPS * RB));
          }
          u_ci += (PAIR ? RB / 2 : RB) / (int)sizeof(T);
          if (u_ci >= p.Cs) {
            u_ci = 0;
            if (++u_c == p.kw) { u_c = 0; if (++u_b == p.kh) { u_b = 0; ++u_a; } }
          }
        }
      } else if (IDENT) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) bufglds16(rsA, kok ? aoff[i] : kOOB, kbyte, xa + i * (RPPS * RB));
      } else {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
          unsigned off;
          if (PACKW) {
            const int ts = arow[i].t * p.st - p.pt + tap.a * p.dt;
            const int hs = arow[i].h * p.sh - p.ph + tap.b * p.dh;
            const bool ok = aok[i] && tap.ok && (unsigned)ts < (unsigned)p.Ts && (unsigned)hs < (unsigned)p.Hs;
            const int w0 = arow[i].w * p.sw - p.pw + tap.c;            // pw already includes the left padding
            const int pix = ((arow[i].n * p.Ts + ts) * p.Hs + hs) * p.Ws + w0;
            off = ok ? (unsigned)pix * 4u * (unsigned)sizeof(T) : kOOB;
          } else {
            bool ok;
            const long long e = src_offset<DGRAD>(p, arow[i], tap, ok);
            off = (ok && aok[i] && tap.ok) ? (unsigned)e * (unsigned)sizeof(T) : kOOB;
          }
          bufglds16(rsA, off, 0, xa + i * (RPPS * RB));
        }
      }
#pragma unroll
      for (int i = 0; i < B_IT; ++i) bufglds16(rsB, kok ? boff[i] : kOOB, kbyte, wb + i * (RPPS * RB));
    }
  };
  // wait until at most `newer` tiles issued after the wanted one are still in flight (loads retire
  // in order, so the wanted tile and everything older -- incl. the PRE rows -- have landed)
  constexpr int LPT = A_IT + B_IT;          // DMA instructions per lane per tile
  auto ring_wait = [&](int newer) {
    if (newer >= 3 && ST >= 5) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPT) : "memory");
    else if (newer == 2 && ST >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
    else if (newer == 1 && ST >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * LPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  f32x4_v acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b) acc[a][b] = f32x4_v{0.f, 0.f, 0.f, 0.f};

  // ---- epilogue operands (same for every path) ---------------------------------------------------
  char* Ob = p.O + (long long)z * p.o_bs * (long long)sizeof(OutT);
  const char* Rb = p.R ? p.R + (long long)z * p.r_bs * (long long)sizeof(T) : nullptr;
  const char* Mb = p.Mask ? p.Mask + (long long)z * p.r_bs * (long long)sizeof(T) : nullptr;
  const char* R2b = p.R2 ? p.R2 + (long long)z * p.r_bs * (long long)sizeof(T) : nullptr;     // low terms (GP::R2 / O2)
  char* O2b = p.O2 ? p.O2 + (long long)z * p.o_bs * ((PAIR || sizeof(T) == 2) ? 2ll : (long long)sizeof(OutT)) : nullptr;
  constexpr int EPT = 16 / (int)sizeof(OutT);   // output elements per 16-byte store
  constexpr int TPR = BN / EPT;                 // lanes per tile row
  constexpr int RPP = NTHR / TPR;               // rows per pass
  constexpr int NPASS = BM / RPP;
  const int tc = tid % TPR, tr = tid / TPR;
  const int ncol = n0 + tc * EPT;
  // PRE: thin-K launches are epilogue (HBM) bound, so the residual / mask rows of this lane are
  // requested BEFORE the k-loop and arrive while the MFMAs run.
  constexpr bool PREM = PRE && !PAIR;          // (PAIR launches are forward convs: no mask operand)
  constexpr bool PRE2 = PRE && PAIR;           // (PAIR: the residual is two planes -- the low one is requested up front as well)
  const bool nt = p.nt_epi != 0;             // epilogue rows with the non-temporal hint (GP::nt_epi)
  uint4 rpre[PRE ? NPASS : 1], mpre[PREM ? NPASS : 1], r2pre[PRE2 ? NPASS : 1];
  if (PRE) {
#pragma unroll
    for (int gp = 0; gp < NPASS; ++gp) {
      const int m = m0 + gp * RPP + tr;
      const bool ok = m < mrows && ncol < p.Ncols;
      const long long off = (row_pos(ok ? m : 0) * p.ldr + ncol) * (long long)sizeof(T);
      rpre[gp] = ld16_epi(nt, src_or_zero(Rb ? Rb : Ab, off, ok && Rb != nullptr));
      if constexpr (PRE2) r2pre[gp] = ld16_epi(nt, src_or_zero(R2b ? R2b : Ab, off, ok && R2b != nullptr));
      if constexpr (PREM) mpre[gp] = ld16_epi(nt, src_or_zero(Mb ? Mb : Ab, off, ok && Mb != nullptr));
    }
  }
  (void)R2b; (void)O2b;

  static_assert(ST >= 2 && ST <= 5 && 3 * LPT < 64, "ring depth / vmcnt immediate");
#pragma unroll
  for (int s0 = 0; s0 < ST - 1; ++s0)
    if (s0 < ktiles) load_tile(s0, s0);

  int cur = 0, nxt = ST - 1;              // ring slots of tile kt and of tile kt + ST - 1
  if constexpr (W2I) {
    // pairs of k-tiles (Wh, Wl of one activation tile).  Weight tile kt lives in the W half of ring slot kt & 1, the
    // activation tile of the pair in the A half of slot (kt >> 1) & 1.  Behind the first barrier of a pair every wave has
    // finished the previous pair (its Wl tile in slot 1 may be refilled); behind the second one every wave has read its A
    // fragments and the Wh tile, so slot 0's W half and the OTHER A half take the next pair's DMA.
    // (PRE instances hold the prefetched residual / mask rows in registers through the k-loop: there the A fragments are
    // read again for the second term instead of kept -- 148 VGPRs = one workgroup per CU otherwise, measured 0.8x)
    constexpr bool HOLD = !PRE;
    for (int kt = 0; kt < ktiles; kt += 2) {
      typename Mma<T>::Frag xf[KSTEPS][FM];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      load_tile(kt + 1, 1);
      {
        const char* xa = smem + ((kt >> 1) & 1) * BUF;
        const char* wb = smem + BM * RB;
        if constexpr (HOLD) {
#pragma unroll
          for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
            for (int i = 0; i < FM; ++i) xf[ks][i] = Mma<T>::template load<RB>(xa, wm * WM + i * 16 + l15, ks, g);
        }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          typename Mma<T>::Frag wf[FN];
          if constexpr (!HOLD) {
#pragma unroll
            for (int i = 0; i < FM; ++i) xf[ks][i] = Mma<T>::template load<RB>(xa, wm * WM + i * 16 + l15, ks, g);
          }
#pragma unroll
          for (int j = 0; j < FN; ++j) wf[j] = Mma<T>::template load<RB>(wb, wn * WN + j * 16 + l15, ks, g);
#pragma unroll
          for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int i = 0; i < FM; ++i) acc[j][i] = Mma<T>::mma(wf[j], xf[ks][i], acc[j][i]);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      if (kt + 2 < ktiles) load_tile(kt + 2, 0);
      {
        const char* xa = smem + ((kt >> 1) & 1) * BUF;
        const char* wb = smem + BUF + BM * RB;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          typename Mma<T>::Frag wf[FN];
          if constexpr (!HOLD) {
#pragma unroll
            for (int i = 0; i < FM; ++i) xf[ks][i] = Mma<T>::template load<RB>(xa, wm * WM + i * 16 + l15, ks, g);
          }
#pragma unroll
          for (int j = 0; j < FN; ++j) wf[j] = Mma<T>::template load<RB>(wb, wn * WN + j * 16 + l15, ks, g);
#pragma unroll
          for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int i = 0; i < FM; ++i) acc[j][i] = Mma<T>::mma(wf[j], xf[ks][i], acc[j][i]);
        }
      }
    }
  } else
  for (int kt = 0; kt < ktiles; ++kt) {
    const int ahead = ktiles - 1 - kt;    // tiles after kt
    ring_wait(ahead < ST - 2 ? ahead : ST - 2);
    // Bare barrier: __syncthreads() carries a workgroup fence that the compiler lowers to
    // vmcnt(0), which would drain the ring.  Every wave has waited for ITS part of tile kt above and
    // has consumed (lgkmcnt) its LDS reads of tile kt-1 before its last MFMAs, so after the barrier
    // tile kt is complete and the slot of tile kt-1 may be refilled.
    asm volatile("s_barrier" ::: "memory");
    const bool more = kt + ST - 1 < ktiles;
    if (more) load_tile(kt + ST - 1, nxt);
    const char* xa = smem + cur * BUF;
    const char* wb = xa + BM * RB;
    if constexpr (PAIR) {
      // k-step 0 of the row = the hi plane's 32 k, k-step 1 = the lo plane's; small terms first
      typename Mma<T>::Frag xh[FM], xl[FM], wh[FN], wl[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        xh[i] = Mma<T>::template load<RB>(xa, wm * WM + i * 16 + l15, 0, g);
        xl[i] = Mma<T>::template load<RB>(xa, wm * WM + i * 16 + l15, 1, g);
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        wh[j] = Mma<T>::template load<RB>(wb, wn * WN + j * 16 + l15, 0, g);
        wl[j] = Mma<T>::template load<RB>(wb, wn * WN + j * 16 + l15, 1, g);
      }
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int i = 0; i < FM; ++i) acc[j][i] = Mma<T>::mma(wl[j], xh[i], acc[j][i]);
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int i = 0; i < FM; ++i) acc[j][i] = Mma<T>::mma(wh[j], xl[i], acc[j][i]);
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int i = 0; i < FM; ++i) acc[j][i] = Mma<T>::mma(wh[j], xh[i], acc[j][i]);
    } else {
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      typename Mma<T>::Frag xf[FM], wf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) xf[i] = Mma<T>::template load<RB>(xa, wm * WM + i * 16 + l15, ks, g);
#pragma unroll
      for (int j = 0; j < FN; ++j) wf[j] = Mma<T>::template load<RB>(wb, wn * WN + j * 16 + l15, ks, g);
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int i = 0; i < FM; ++i) acc[j][i] = Mma<T>::mma(wf[j], xf[i], acc[j][i]);
    }
    }
    cur = cur + 1 == ST ? 0 : cur + 1;
    nxt = nxt + 1 == ST ? 0 : nxt + 1;
  }
  __syncthreads();                        // all waves are done with the last tile: LDS is reused below

  // ---- epilogue ------------------------------------------------------------------------------
  if (p.vec_epi) {
    // Coalesced path: the fp32 accumulator tile goes through LDS (16-byte chunks XOR-swizzled by
    // row & 7, conflict-free for both the fragment-shaped writes and the row-shaped reads), then
    // every lane handles EPT consecutive columns of one row: residual / mask are read and the
    // result is written with full 16-byte accesses, whole rows of the tile per wavefront.
    // The tile is staged in one pass when it fits the LDS of this launch, else in two halves
    // (rows of wave-row 0, then of wave-row 1): p.epi is 1 or 2.
    constexpr int CPR = BN / 4;          // 16-byte fp32 chunks per tile row
    const int epi = p.epi;
    auto stage = [&](int h) {
      if (epi == 1 || wm == h) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const int row = (epi == 1 ? wm * WM : 0) + i * 16 + l15;
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            const int c = (wn * WN + j * 16 + g * 4) >> 2;
            *reinterpret_cast<float4*>(smem + ((row * CPR + (c ^ (row & 7))) << 4)) =
                make_float4(acc[j][i][0] * p.alpha, acc[j][i][1] * p.alpha, acc[j][i][2] * p.alpha,
                            acc[j][i][3] * p.alpha);
          }
        }
      }
    };
    stage(0);
    __syncthreads();
#pragma unroll
    for (int gp = 0; gp < NPASS; ++gp) {
      if (gp == NPASS / 2 && epi == 2) {
        __syncthreads();
        stage(1);
        __syncthreads();
      }
      const int trow = gp * RPP + tr;                                   // row inside the tile
      const int row = (epi == 2 && gp >= NPASS / 2) ? trow - BM / 2 : trow;   // row inside the staged half
      const int m = m0 + trow;
      if (m < mrows && ncol < p.Ncols) {
        const long long mpos = row_pos(m);
        float v[EPT];
#pragma unroll
        for (int q = 0; q < EPT / 4; ++q) {
          const int c = tc * (EPT / 4) + q;
          const float4 t = *reinterpret_cast<const float4*>(smem + ((row * CPR + (c ^ (row & 7))) << 4));
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
        const long long ridx = mpos * p.ldr + ncol;
        if (Rb) {
          float r[EPT];
          if (PRE) unpack_elems<T, EPT>(rpre[PRE ? gp : 0], r);
          else load_elems_epi<T, EPT>(nt, reinterpret_cast<const T*>(Rb) + ridx, r);
#pragma unroll
          for (int e = 0; e < EPT; ++e) v[e] += r[e];
        }
        if constexpr (sizeof(T) == 2 && sizeof(OutT) == 2) {
          if (R2b) {                                 // low term of a two-term residual
            float r[EPT];
            if constexpr (PRE2) unpack_elems<T, EPT>(r2pre[PRE2 ? gp : 0], r);
            else load_elems_epi<T, EPT>(nt, reinterpret_cast<const T*>(R2b) + ridx, r);
#pragma unroll
            for (int e = 0; e < EPT; ++e) v[e] += r[e];
          }
        }
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < EPT; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (Mb && !PAIR) {
          float r[EPT];
          if (PREM) unpack_elems<T, EPT>(mpre[PREM ? gp : 0], r);
          else load_elems_epi<T, EPT>(nt, reinterpret_cast<const T*>(Mb) + ridx, r);
#pragma unroll
          for (int e = 0; e < EPT; ++e) v[e] = r[e] > 0.f ? v[e] : 0.f;
        }
        OutT* o = reinterpret_cast<OutT*>(Ob) + mpos * p.ldo + ncol;
        if (sizeof(OutT) == 4) {
          st16_epi(nt, o, make_float4(v[0], v[1], v[2], v[3]));
          if constexpr (PAIR) {
            // fp32 output of a two-plane launch (theta / phi / g of a non-local block): O2 = the fp16 copy of the output that
            // the fp16 backward reads (positive values stay positive, as vlfb_half_copy)
            if (O2b) st8_epi(nt, reinterpret_cast<unsigned short*>(O2b) + mpos * p.ldo + ncol,
                             make_uint2(pack_h2_pos(v[0], v[1]), pack_h2_pos(v[2 % EPT], v[3 % EPT])));
          } else if constexpr (sizeof(T) == 2) {
            // fp32 output of a 16-bit launch (the "mix" backward: a DGRAD into an fp32 gradient slot): O2 = the same values
            // rounded to T, for the 16-bit launches that read this gradient next (saves them a cast pass)
            if (O2b) st8_epi(nt, reinterpret_cast<T*>(O2b) + mpos * p.ldo + ncol,
                             make_uint2(Elem<T>::pack2(v[0], v[1]), Elem<T>::pack2(v[2 % EPT], v[3 % EPT])));
          }
        } else {
          const uint4 hv = make_uint4(Elem<OutT>::pack2(v[0], v[1]), Elem<OutT>::pack2(v[2 % EPT], v[3 % EPT]),
                                      Elem<OutT>::pack2(v[4 % EPT], v[5 % EPT]), Elem<OutT>::pack2(v[6 % EPT], v[7 % EPT]));
          st16_epi(nt, o, hv);
          if (O2b) {                                 // low term: what the rounding of the stored value lost
            float h[EPT];
            unpack_elems<OutT, EPT>(hv, h);
            st16_epi(nt, reinterpret_cast<OutT*>(O2b) + mpos * p.ldo + ncol,
                     make_uint4(Elem<OutT>::pack2(v[0] - h[0], v[1] - h[1]), Elem<OutT>::pack2(v[2 % EPT] - h[2 % EPT], v[3 % EPT] - h[3 % EPT]),
                                Elem<OutT>::pack2(v[4 % EPT] - h[4 % EPT], v[5 % EPT] - h[5 % EPT]), Elem<OutT>::pack2(v[6 % EPT] - h[6 % EPT], v[7 % EPT] - h[7 % EPT])));
          }
        }
      }
    }
    return;
  }
  // Generic path (odd column counts / leading dimensions): lane holds n = nb + 0..3 for row m.
  const bool vec_ok = (p.ldo & 3) == 0;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + wm * WM + i * 16 + l15;
    if (m >= mrows) continue;
    const long long mpos = row_pos(m);
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int nb = n0 + wn * WN + j * 16 + g * 4;
      if (nb >= p.Ncols) continue;
      const int cnt = (p.Ncols - nb) < 4 ? (p.Ncols - nb) : 4;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = acc[j][i][r] * p.alpha;
        if (r < cnt) {
          if (p.bias_mode == VLFB_BIAS_COL) x += p.bias[nb + r];
          else if (p.bias_mode == VLFB_BIAS_ROW) x += p.bias[m];
          if (Rb) x += ld_elem<T>(Rb, mpos * p.ldr + nb + r);
          if (p.relu) x = fmaxf(x, 0.f);
          if (Mb) x = ld_elem<T>(Mb, mpos * p.ldr + nb + r) > 0.f ? x : 0.f;
        }
        v[r] = x;
      }
      store4<OutT>(Ob, mpos * p.ldo + nb, v, cnt, vec_ok);
    }
  }
}

}  // namespace
}  // namespace vlfb
