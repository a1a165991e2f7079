// Implicit-GEMM 3-D convolution / batched GEMM on CDNA4 matrix cores (gfx950).
//
// One kernel family covers every contraction on the hot path (SURVEY.md 8a rows M1, M4-M6, M10,
// H3-H5): conv fprop, conv dgrad (NT form: both operands K-contiguous) and conv wgrad / the
// "contract over positions" attention products (TN form: both operands are stored with the
// contraction index as the slow dimension, so the stager transposes 8x8 (bf16) / 4x4 (fp32)
// register blocks on the way into LDS).
//
// Data layout: activations are channels-last [N,T,H,W,C] so a 16-byte global load is a run of
// consecutive K for one output row; weights are [Cout][tap][Cin].  LDS tiles are rows of 128
// bytes (64 bf16 / 32 fp32 of K) with the 16-byte chunk index XOR-ed by (row & 7), which makes
// both the ds_write_b128 of the stager and the ds_read_b128 of the fragment reader
// conflict-free (cdna_hip_programming.md, T2).  Wave tile 64x64 (2x2 waves, 128x128 block) out
// of 16x16 MFMA fragments: v_mfma_f32_16x16x32_bf16 on the throughput path,
// v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain) on the parity path.  The two paths share the
// same lane<->k mapping: lane l owns 8 consecutive k of row (l & 15) at k-offset (l >> 4) * 8.
// MFMA operands are swapped (weights as A, activations as B) so that each lane ends up holding
// 4 consecutive output channels of one output row -> 8/16-byte epilogue stores.
#include "vlfb_gemm_common.h"
#include "vlfb_gemm_nt.h"
#include <string>
#include <unordered_map>

namespace vlfb {
namespace {

// =============================================================================================
// TN kernel: O[pp][qq] = sum_m P[m][pp] * Xg[m][qq]   (qq = (tap, channel) of the gathered input)
// =============================================================================================
// register-block transposes: `in[j]` = 16 bytes of consecutive channels for position j;
// `out[c]` = 16 bytes of consecutive positions for channel c.
__device__ __forceinline__ uint32_t word_of(const uint4& v, int i) {
  return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
}
__device__ __forceinline__ void transpose_block(const uint4 (&in)[8], uint4 (&out)[8], bf16_t) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    uint32_t o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t w0 = word_of(in[2 * q], c >> 1), w1 = word_of(in[2 * q + 1], c >> 1);
      o[q] = (c & 1) ? ((w0 >> 16) | (w1 & 0xffff0000u)) : ((w0 & 0xffffu) | (w1 << 16));
    }
    out[c] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}
__device__ __forceinline__ void transpose_block(const uint4 (&in)[8], uint4 (&out)[8], f16_t) {
  transpose_block(in, out, bf16_t());      // a pure 16-bit shuffle
}
__device__ __forceinline__ void transpose_block(const uint4 (&in)[4], uint4 (&out)[4], float) {
  out[0] = make_uint4(in[0].x, in[1].x, in[2].x, in[3].x);
  out[1] = make_uint4(in[0].y, in[1].y, in[2].y, in[3].y);
  out[2] = make_uint4(in[0].z, in[1].z, in[2].z, in[3].z);
  out[3] = make_uint4(in[0].w, in[1].w, in[2].w, in[3].w);
}

// TN gather state of one staging block: the tap is fixed per thread, positions advance, so the
// (n, t, h) part of the source address and its validity are cached and only refreshed when the
// position wraps to a new output row -- no integer division inside the k loop.
struct QRow {
  int n, t, h, w;
  long long base;   // pixel index of source row (n, ts, hs, 0)
  bool hv;          // ts, hs inside the source
};
__device__ __forceinline__ void qrow_refresh(const GP& p, QRow& r, const TapC& tp) {
  const int ts = r.t * p.st - p.pt + tp.a * p.dt;
  const int hs = r.h * p.sh - p.ph + tp.b * p.dh;
  r.hv = (unsigned)ts < (unsigned)p.Ts && (unsigned)hs < (unsigned)p.Hs;
  r.base = ((long long)(r.n * p.Ts + ts) * p.Hs + hs) * p.Ws;
}
__device__ __forceinline__ void qrow_next(const GP& p, QRow& r, const TapC& tp) {
  if (++r.w == p.Wr) {
    r.w = 0;
    if (++r.h == p.Hr) {
      r.h = 0;
      if (++r.t == p.Tr) { r.t = 0; ++r.n; }
    }
    qrow_refresh(p, r, tp);
  }
}
// r += (dn, dt, dh, dw) in the mixed radix (Tr, Hr, Wr)
__device__ __forceinline__ void qrow_jump(const GP& p, QRow& r, const RowC& d, const TapC& tp) {
  r.w += d.w; if (r.w >= p.Wr) { r.w -= p.Wr; ++r.h; }
  r.h += d.h; if (r.h >= p.Hr) { r.h -= p.Hr; ++r.t; }
  r.t += d.t; if (r.t >= p.Tr) { r.t -= p.Tr; ++r.n; }
  r.n += d.n;
  qrow_refresh(p, r, tp);
}
template <typename T, bool PACKW>
__device__ __forceinline__ uint4 qrow_load(const GP& p, const char* base, const QRow& r, const TapC& tp, bool ok) {
  ok = ok && tp.ok && r.hv;
  if (PACKW) {   // W-padded stem input: always inside the row
    const int w0 = r.w * p.sw - p.pw + tp.c;
    return ld16_if(base, (r.base + w0) * 4 * (long long)sizeof(T), ok);
  } else {
    const int ws = r.w * p.sw - p.pw + tp.c * p.dw;
    return ld16_if(base, ((r.base + ws) * p.lda + tp.ci) * (long long)sizeof(T), ok && (unsigned)ws < (unsigned)p.Ws);
  }
}

template <typename T, typename OutT, int BP, int BQ, bool IDENT, bool PACKW>
__global__ __launch_bounds__(kThreads) void gemm_tn_kernel(const GP p) {
  constexpr int EPC = Elem<T>::EPC;
  constexpr int BK = kRowBytes / (int)sizeof(T);
  constexpr int NPB = BP / EPC * 8, NQB = BQ / EPC * 8;  // staging blocks per tile
  constexpr int ITER = (NPB + NQB + kThreads - 1) / kThreads;
  constexpr int WP = BP / 2, WQ = BQ / 2;
  constexpr int FP = WP / 16, FQ = WQ / 16;
  constexpr int BUF = (BP + BQ) * kRowBytes;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wp = wave >> 1, wq = wave & 1;
  const int l15 = lane & 15, g = lane >> 4;

  const int nwg = p.tiles_m * p.tiles_n;
  int bid, split;
  if (p.splits > 1 && (p.splits & 7) == 0) {
    // 1-D grid, XCD-grouped: workgroup id -> (xcd = id % 8, slot = id / 8).  All output tiles of
    // one position-split run back to back on ONE XCD, so the P / X panels of that split are
    // fetched from HBM once and re-used out of that XCD's L2 by the other tiles.
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

  const char* Pb = p.P + (long long)z * p.p_bs * (long long)sizeof(T);
  const char* Ab = p.A + (long long)z * p.a_bs * (long long)sizeof(T);

  const int kbeg = split * p.kper;
  const int kend = min(p.M, kbeg + p.kper);
  const int ktiles = (kend - kbeg + BK - 1) / BK;

  // static per-thread block assignment
  int blk_kind[ITER], blk_r[ITER], blk_k[ITER];
  TapC qtap[ITER];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    int id = tid + it * kThreads;
    if (id < NPB) { blk_kind[it] = 0; blk_r[it] = id >> 3; blk_k[it] = id & 7; }
    else if (id < NPB + NQB) { id -= NPB; blk_kind[it] = 1; blk_r[it] = id >> 3; blk_k[it] = id & 7; }
    else { blk_kind[it] = 2; blk_r[it] = 0; blk_k[it] = 0; }
    if (blk_kind[it] == 1) {
      int kc = (q0 + blk_r[it] * EPC) / EPC;
      if (IDENT) { qtap[it].ok = kc * EPC < p.K; qtap[it].a = qtap[it].b = qtap[it].c = 0; qtap[it].ci = 0; }
      else qtap[it] = decode_tap<T, PACKW>(p, kc);
    }
  }
  // gather cursors (generic path): first position of the block in k-tile 0, and the BK jump
  QRow qrow[ITER];
  RowC jump;
  if (!IDENT) {
    jump = decode_row(p, BK - EPC);   // after a tile the cursor already moved EPC positions
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      if (blk_kind[it] == 1) {
        const RowC r0 = decode_row(p, min(kbeg + blk_k[it] * EPC, p.M - 1));
        qrow[it].n = r0.n; qrow[it].t = r0.t; qrow[it].h = r0.h; qrow[it].w = r0.w;
        qrow_refresh(p, qrow[it], qtap[it]);
      }
    }
  }

  uint4 stg[ITER][EPC];

  auto load_tile = [&](int kt) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int kk = kbeg + kt * BK + blk_k[it] * EPC;  // first position of the block
      if (blk_kind[it] == 0) {
        const int pc = p0 + blk_r[it] * EPC;
        const bool cok = pc < p.Ncols;
#pragma unroll
        for (int j = 0; j < EPC; ++j) {
          const int k = kk + j;
          stg[it][j] = ld16_if(Pb, ((long long)k * p.ldp + pc) * (long long)sizeof(T), cok && k < kend);
        }
      } else if (blk_kind[it] == 1) {
        const int kc = (q0 + blk_r[it] * EPC) / EPC;
        if (IDENT) {
          RowC unused;
#pragma unroll
          for (int j = 0; j < EPC; ++j) {
            const int k = kk + j;
            stg[it][j] = load_act_chunk<T, true, false, false>(p, Ab, k, k < kend, unused, qtap[it], kc);
          }
        } else {
#pragma unroll
          for (int j = 0; j < EPC; ++j) {
            stg[it][j] = qrow_load<T, PACKW>(p, Ab, qrow[it], qtap[it], kk + j < kend);
            qrow_next(p, qrow[it], qtap[it]);
          }
          qrow_jump(p, qrow[it], jump, qtap[it]);   // to this block's slot in the next k-tile
        }
      }
    }
  };
  auto store_tile = [&](int buf) {
    char* pt = smem + buf * BUF;
    char* qt = pt + BP * kRowBytes;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      if (blk_kind[it] == 2) continue;
      uint4 tr[EPC];
      transpose_block(stg[it], tr, T());
      char* dst = blk_kind[it] == 0 ? pt : qt;
#pragma unroll
      for (int c = 0; c < EPC; ++c)
        *reinterpret_cast<uint4*>(dst + lds_off(blk_r[it] * EPC + c, blk_k[it])) = tr[c];
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
    const char* qt = pt + BP * kRowBytes;
#pragma unroll
    for (int ks = 0; ks < Mma<T>::KSTEPS; ++ks) {
      typename Mma<T>::Frag pf[FP], qf[FQ];
#pragma unroll
      for (int i = 0; i < FP; ++i) pf[i] = Mma<T>::load(pt, wp * WP + i * 16 + l15, ks, g);
#pragma unroll
      for (int j = 0; j < FQ; ++j) qf[j] = Mma<T>::load(qt, wq * WQ + j * 16 + l15, ks, g);
#pragma unroll
      for (int j = 0; j < FQ; ++j)
#pragma unroll
        for (int i = 0; i < FP; ++i) acc[j][i] = Mma<T>::mma(qf[j], pf[i], acc[j][i]);
    }
    if (more) store_tile((kt + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds qq = qb + 0..3 for output row pp ---------------------------------
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
        store4<float>(reinterpret_cast<char*>(p.ws),
                      (long long)split * ((long long)p.Ncols * p.ldo) + idx, v, cnt, vec_ok);
      } else {
        char* Ob = p.O + (long long)z * p.o_bs * (long long)sizeof(OutT);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = acc[j][i][r] * p.alpha * rs;
          if (p.accumulate && r < cnt) x += ld_elem<OutT>(Ob, idx + r);
          v[r] = x;
        }
        store4<OutT>(Ob, idx, v, cnt, vec_ok);
      }
    }
  }
}


// =============================================================================================
// TN kernel, bf16, DMA + LDS transpose-read variant.
// The operand tiles are copied as they lie in HBM ([position][channel], 16-byte pieces by
// global_load_lds) -- no VGPR staging, no register transposes, no ds_write -- and the MFMA fragments
// (8 consecutive positions of one channel per lane) are produced by ds_read_b64_tr_b16, which
// returns to lane i of a 16-lane group column i of a 4(k) x 16(channel) block.
// LDS rows are RS = 2*B bytes (one position); the 32-byte segment index is XOR-ed with a key of
// the row so that the 8 rows touched by a 32-lane half of a tr-read fall on 8 different 32-byte
// bank segments (the DMA destination is lane-linear, so the XOR is applied on the SOURCE chunk).
// The DMAs are issued through bufglds16_hidden: with the builtin, hipcc put s_waitcnt vmcnt(0) in front of the
// transposed reads of every k-tile, i.e. it waited for the NEXT tile's DMA right after issuing it.
// =============================================================================================
template <typename T, typename OutT, int BP, int BQ, bool IDENT, bool PACKW, int NW = 4, bool SP = false>
__global__ __launch_bounds__(64 * NW) void gemm_tn_tr_kernel(const GP p) {
  // NW = 4: waves 2 (p) x 2 (q).  NW = 8: waves 2 x 4 on the same tile -- twice the wavefronts per CU.
  // SP (split-bf16 math on fp32 storage, vlfb_gemm_split.hip): BOTH operands arrive as two bf16 term planes [plane][position]
  // [channel] (h = bf16(x), m = bf16(x - h); a_ps / p_ps elements apart) and a fragment pair issues three MFMAs
  // (m.h, h.m, h.h); k-tiles of 32 positions keep the two-plane tiles at the LDS footprint of the plain kernel.
  typedef typename V16<T>::V vec_t;
  constexpr int NTHR = 64 * NW;
  constexpr int NWQ = NW / 2;
  constexpr int NPL = SP ? 2 : 1;
  constexpr int BK = SP ? 32 : 64;               // positions per k-tile
  constexpr int RSP = BP * 2, RSQ = BQ * 2;      // LDS row bytes (one position)
  constexpr int CP = BP / 8, CQ = BQ / 8;        // 16-byte chunks per row
  constexpr int PI = BK * CP / NTHR, QI = BK * CQ / NTHR;   // DMA pieces per thread
  constexpr int RPP_P = NTHR / CP, RPP_Q = NTHR / CQ;       // rows per pass
  constexpr int WP = BP / 2, WQ = BQ / NWQ;
  static_assert(PI >= 1 && QI >= 1 && WQ >= 16, "tile too small for this many waves");
  constexpr int FP = WP / 16, FQ = WQ / 16;
  constexpr int PLP = BK * RSP, PLQ = BK * RSQ;  // bytes of one plane tile
  constexpr int BUF = NPL * (PLP + PLQ);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
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
  const char* Pb = p.P + (long long)z * p.p_bs * 2;
  const char* Ab = p.A + (long long)z * p.a_bs * 2;
  const int kbeg = split * p.kper;
  const int kend = min(p.M, kbeg + p.kper);
  const int ktiles = (kend - kbeg + BK - 1) / BK;

  // ---- per-thread DMA assignment ---------------------------------------------------------------
  const int pslot = tid % CP, prow = tid / CP;               // LDS slot (row, 16-byte slot) of the P tile
  const int pc = (((pslot >> 1) ^ tr_key<RSP>(prow)) << 1) | (pslot & 1);   // global channel chunk
  const int pch = p0 + pc * 8;
  const bool pok = pch < p.Ncols;
  const int qslot = tid % CQ, qrow0 = tid / CQ;
  const int qc = (((qslot >> 1) ^ tr_key<RSQ>(qrow0)) << 1) | (qslot & 1);
  const int kc = q0 / 8 + qc;                                 // global 16-byte chunk index along K
  TapC qtap;
  if (IDENT) { qtap.ok = kc * 8 < p.K; qtap.a = qtap.b = qtap.c = 0; qtap.ci = 0; }
  else qtap = decode_tap<T, PACKW>(p, kc);
  QRow qcur[QI];
  RowC jump;
  if (!IDENT) {
    jump = decode_row(p, BK);
#pragma unroll
    for (int i = 0; i < QI; ++i) {
      const RowC r = decode_row(p, min(kbeg + qrow0 + RPP_Q * i, p.M - 1));
      qcur[i].n = r.n; qcur[i].t = r.t; qcur[i].h = r.h; qcur[i].w = r.w;
      qrow_refresh(p, qcur[i], qtap);
    }
  }

  // DMA through buffer descriptors whose extent ends at position `kend`: rows past the end of this
  // split's slab (only the last k-tile can have them) are zero-filled by the range check, the k-tile
  // advance is the scalar offset, and the per-lane part is a loop-invariant 32-bit byte offset.
  // (one descriptor per term plane: the range check that zero-fills the ragged last k-tile is per plane)
  __amdgpu_buffer_rsrc_t rsP[NPL], rsQ[NPL];
#pragma unroll
  for (int pl = 0; pl < NPL; ++pl) {
    rsP[pl] = make_rsrc(Pb + (long long)pl * p.p_ps * 2, (unsigned)kend * (unsigned)p.ldp * 2u);
    rsQ[pl] = make_rsrc(Ab + (long long)pl * p.a_ps * 2, IDENT ? (unsigned)kend * (unsigned)p.lda * 2u : p.a_bytes);
  }
  unsigned poff[PI], qoff[IDENT ? QI : 1];
#pragma unroll
  for (int i = 0; i < PI; ++i)
    poff[i] = pok ? (unsigned)((kbeg + prow + RPP_P * i) * p.ldp + pch) * 2u : kOOB;
  if (IDENT) {
#pragma unroll
    for (int i = 0; i < QI; ++i)
      qoff[i] = qtap.ok ? (unsigned)((kbeg + qrow0 + RPP_Q * i) * p.lda + kc * 8) * 2u : kOOB;
  }

  auto load_tile = [&](int kt, int buf) {
    char* pt = smem + buf * BUF + wave_u * 1024;
    char* qt = smem + buf * BUF + NPL * PLP + wave_u * 1024;
    const int kb = kbeg + kt * BK;
    const unsigned pstep = (unsigned)(kt * BK) * (unsigned)p.ldp * 2u;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int i = 0; i < PI; ++i) bufglds16_hidden(rsP[pl], poff[i], pstep, pt + pl * PLP + i * (NTHR * 16));
    if (IDENT) {
      const unsigned qstep = (unsigned)(kt * BK) * (unsigned)p.lda * 2u;
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int i = 0; i < QI; ++i) bufglds16_hidden(rsQ[pl], qoff[i], qstep, qt + pl * PLQ + i * (NTHR * 16));
    } else {
#pragma unroll
      for (int i = 0; i < QI; ++i) {
        const int k = kb + qrow0 + RPP_Q * i;
        unsigned off;
        if (PACKW) {   // W-padded stem input: the two pixels of the chunk are always inside the row
          const int w0 = qcur[i].w * p.sw - p.pw + qtap.c;
          off = (qtap.ok && qcur[i].hv && k < kend) ? (unsigned)(qcur[i].base + w0) * 8u : kOOB;
        } else {
          const int ws = qcur[i].w * p.sw - p.pw + qtap.c * p.dw;
          const bool ok = qtap.ok && qcur[i].hv && k < kend && (unsigned)ws < (unsigned)p.Ws;
          off = ok ? (unsigned)((qcur[i].base + ws) * p.lda + qtap.ci) * 2u : kOOB;
        }
        qrow_jump(p, qcur[i], jump, qtap);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) bufglds16_hidden(rsQ[pl], off, 0, qt + pl * PLQ + i * (NTHR * 16));
      }
    }
  };

  f32x4_v acc[FQ][FP];
#pragma unroll
  for (int a = 0; a < FQ; ++a)
#pragma unroll
    for (int b = 0; b < FP; ++b) acc[a][b] = f32x4_v{0.f, 0.f, 0.f, 0.f};

  // bias gradient (GP::dbias): the workgroups of column tile 0 also sum the P tiles over the positions.  Thread -> one
  // 16-byte channel chunk (bc) of BK / BG consecutive-stride positions; 8 fp32 partial sums per thread, folded at the end.
  constexpr int BG = NTHR / CP;                 // position groups
  constexpr int BPP = BK / BG;                  // positions per thread and k-tile
  static_assert(SP || (BK % BG == 0 && BPP >= 1), "bias-gradient thread map");
  const bool do_bias = !SP && p.dbias != nullptr && tile_q == 0;     // workgroup-uniform
  const int bc = tid % CP, bg = tid / CP;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  if (ktiles > 0) load_tile(0, 0);
  for (int kt = 0; kt < ktiles; ++kt) {
    // own DMA of tile kt landed, then a bare barrier (see gemm_nt_kernel): tile kt is complete and
    // the other buffer, last read for tile kt-1, may be refilled
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (kt + 1 < ktiles) load_tile(kt + 1, (kt + 1) & 1);
    const char* pt = smem + (kt & 1) * BUF;
    const char* qt = pt + NPL * PLP;
    if constexpr (!SP) {
      if (do_bias) {
#pragma unroll
        for (int j = 0; j < BPP; ++j) {
          const int r = bg + BG * j;            // position inside the tile (rows past the slab end were zero-filled)
          const int slot = (((bc >> 1) ^ tr_key<RSP>(r)) << 1) | (bc & 1);
          float v[8];
          Vec16<T>::load(reinterpret_cast<const T*>(pt + r * RSP + slot * 16), v);
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum[e] += v[e];
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      vec_t pf[NPL][FP], qf[NPL][FQ];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
        for (int i = 0; i < FP; ++i) pf[pl][i] = tr_frag<RSP, vec_t>(pt + pl * PLP, wp * WP + i * 16, ks, lane);
#pragma unroll
        for (int j = 0; j < FQ; ++j) qf[pl][j] = tr_frag<RSQ, vec_t>(qt + pl * PLQ, wq * WQ + j * 16, ks, lane);
      }
      if constexpr (SP) {      // cross terms first, then the leading product (16 accumulators between two MFMAs on one)
#pragma unroll
        for (int j = 0; j < FQ; ++j)
#pragma unroll
          for (int i = 0; i < FP; ++i) acc[j][i] = V16<T>::mma(qf[NPL - 1][j], pf[0][i], acc[j][i]);
#pragma unroll
        for (int j = 0; j < FQ; ++j)
#pragma unroll
          for (int i = 0; i < FP; ++i) acc[j][i] = V16<T>::mma(qf[0][j], pf[NPL - 1][i], acc[j][i]);
      }
#pragma unroll
      for (int j = 0; j < FQ; ++j)
#pragma unroll
        for (int i = 0; i < FP; ++i)
          acc[j][i] = V16<T>::mma(qf[0][j], pf[0][i], acc[j][i]);
    }
  }

  // ---- epilogue: lane holds qq = qb + 0..3 for output row pp (same as gemm_tn_kernel) ----------
  const bool vec_ok = (p.ldo & 3) == 0;
  const bool to_ws = p.splits > 1;
  if constexpr (!SP) {
    if (do_bias) {                               // (workgroup-uniform) fold the BG position groups through LDS, in order
      __syncthreads();                           // every wave is done with the last tile
      float* red = reinterpret_cast<float*>(smem);
#pragma unroll
      for (int e = 0; e < 8; ++e) red[bg * BP + bc * 8 + e] = bsum[e];
      __syncthreads();
      if (tid < BP && p0 + tid < p.Ncols) {
        float sum = 0.f;
        for (int j = 0; j < BG; ++j) sum += red[j * BP + tid];
        const int pp = p0 + tid;
        if (to_ws) p.ws[(long long)p.splits * ((long long)p.Ncols * p.ldo) + (long long)split * p.Ncols + pp] = sum;
        else p.dbias[pp] = sum * p.alpha * (p.rowscale ? p.rowscale[pp] : 1.f);
      }
    }
  }
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
        store4<float>(reinterpret_cast<char*>(p.ws),
                      (long long)split * ((long long)p.Ncols * p.ldo) + idx, v, cnt, vec_ok);
      } else {
        char* Ob = p.O + (long long)z * p.o_bs * (long long)sizeof(OutT);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = acc[j][i][r] * p.alpha * rs;
          if (p.accumulate && r < cnt) x += ld_elem<OutT>(Ob, idx + r);
          v[r] = x;
        }
        store4<OutT>(Ob, idx, v, cnt, vec_ok);
      }
    }
  }
}

// =============================================================================================
// Stem WGRAD (conv1, bf16): dW[co][a][b][kw][c] = sum_pos G[pos][co] * X[t+a][2h+b][2w+kw][c].
// With Cin = 3 (packed to 4) the K = (kt*kh) x (8 kw x 4 c) gathered columns of a position come from
// only kt*kh short input rows, so the generic TN kernel spends its time issuing 16-byte gather DMAs
// (9 column tiles x 6 pieces per lane per 64 positions) and re-reads the 411 MB gradient once per
// column tile.  Here one workgroup takes whole OUTPUT ROWS (n, t, h): it stages the kt*kh raw input
// rows that row touches (Ws*8 bytes each, contiguous DMA) plus the Wr x 64 gradient tile, and every
// wave builds its MFMA fragments straight from the raw rows with ds_read_b64_tr_b16 -- the 16 columns
// (4 kw x 4 c) of a position are 32 contiguous bytes, consecutive positions are sw pixels apart.
// 8 waves x 9 column tiles of 16 x all 64 output channels = the whole 64 x K gradient stays in
// registers (144 accumulator VGPRs) across the workgroup's rows; slabs + wgrad_reduce as usual.
// 6x fewer DMA pieces, the gradient is read once.
// =============================================================================================
constexpr int kStemCT = 9;    // column tiles (16 of the K columns each) per wave
template <typename T>
__global__ __launch_bounds__(512) void stem_wgrad_kernel(const GP p) {
  typedef typename V16<T>::V vec_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int l15 = lane & 15, g = lane >> 4;
  const int taps = p.K >> 5;                       // (a, b) pairs
  const int rbytes = p.Ws * 8;                     // one staged input row (Cs = 4 bf16 per pixel)
  const int ppr = rbytes >> 4;                     // 16-byte pieces per row
  const int npieces = taps * ppr;
  const int poff_lds = npieces * 16;               // gradient tile behind the rows
  const int stage = poff_lds + p.Wr * 128;         // Cn = 64 bf16 per position
  const int split = blockIdx.x;
  const int tiles_total = p.tiles_m, tpw = p.kper;
  const int tile_beg = split * tpw;
  const int tile_end = min(tiles_total, tile_beg + tpw);

  const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.A, p.a_bytes);
  const __amdgpu_buffer_rsrc_t rsG = make_rsrc(p.P, p.b_bytes);

  // ---- per-lane DMA assignment (tile-invariant part) -------------------------------------------
  constexpr int RI = 8;                            // row pieces per lane (RI * 512 >= npieces, host-checked)
  int ra[RI], rb[RI], rj[RI];
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int id = tid + 512 * i;
    const int tap = id / ppr;
    ra[i] = tap / p.kh;
    rb[i] = tap - ra[i] * p.kh;
    rj[i] = (id - tap * ppr) * 16;
  }
  // gradient tile: piece id -> (position row, 16-byte slot); the source chunk is XOR-swizzled so
  // that tr_frag<128> reads conflict-free (same layout as gemm_tn_tr_kernel with BP = 64)
  unsigned gvoff[2];
  bool gok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + 512 * i;
    const int row = id >> 3, slot = id & 7;
    const int pc = (((slot >> 1) ^ tr_key<128>(row)) << 1) | (slot & 1);
    gok[i] = row < p.Wr;
    gvoff[i] = (unsigned)(row * p.ldp + pc * 8) * 2u;
  }

  auto load_tile = [&](int tile, int buf) {
    const int h = tile % p.Hr;
    const int nt = tile / p.Hr;
    const int t = nt % p.Tr, n = nt / p.Tr;
    char* base = smem + buf * stage;
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      if (tid + 512 * i < npieces) {
        const int tin = t * p.st - p.pt + ra[i], hin = h * p.sh - p.ph + rb[i];
        const bool ok = (unsigned)tin < (unsigned)p.Ts && (unsigned)hin < (unsigned)p.Hs;
        const unsigned off = ok ? (unsigned)(((n * p.Ts + tin) * p.Hs + hin) * rbytes + rj[i]) : kOOB;
        bufglds16_hidden(rsX, off, 0, base + (i * 512 + wave_u * 64) * 16);
      }
    }
    const unsigned gstep = (unsigned)(tile * p.Wr) * (unsigned)p.ldp * 2u;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (gok[i]) bufglds16_hidden(rsG, gvoff[i], gstep, base + poff_lds + (i * 512 + wave_u * 64) * 16);
  };

  f32x4_v acc[kStemCT][4];
#pragma unroll
  for (int j = 0; j < kStemCT; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_v{0.f, 0.f, 0.f, 0.f};

  // this wave's column tiles: ct = wave * 9 + j  ->  tap = ct >> 1, kw half = ct & 1
  const int ct0 = wave_u * kStemCT;
  const int ncts = 2 * taps;
  const int pl = lane & 15;
  const int ksteps = (p.Wr + 31) >> 5;

  if (tile_beg < tile_end) load_tile(tile_beg, 0);
  for (int tile = tile_beg; tile < tile_end; ++tile) {
    const int buf = (tile - tile_beg) & 1;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (tile + 1 < tile_end) load_tile(tile + 1, buf ^ 1);
    const char* rows = smem + buf * stage;
    const char* gt = rows + poff_lds;
    for (int ks = 0; ks < ksteps; ++ks) {
      const int r0 = ks * 32 + 8 * g + (pl >> 2);            // position read by this lane (first half)
      const bool live = ks * 32 + 8 * g < p.Wr;              // Wr % 8 == 0: the 8 positions of a group live or die together
      vec_t pf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        pf[i] = tr_frag<128, vec_t>(gt, i * 16, ks, lane);
        if (!live) pf[i] = vec_t{0, 0, 0, 0, 0, 0, 0, 0};
      }
      // dead positions read position Wr-1 instead (finite data x the zeroed gradient fragment)
      const int q0 = min(r0, p.Wr - 1), q1 = min(r0 + 4, p.Wr - 1);
      const int a0 = (q0 * p.sw - p.pw) * 8 + (pl & 3) * 8;
      const int a1 = (q1 * p.sw - p.pw) * 8 + (pl & 3) * 8;
      // all 9 fragments first, then the 36 MFMAs (no control flow in between: the column tiles past
      // the end of K, on the last wave only, recompute the last real tile and are dropped at the store)
      vec_t qf[kStemCT];
#pragma unroll
      for (int j = 0; j < kStemCT; ++j) {
        const int ct = min(ct0 + j, ncts - 1);
        const char* rowp = rows + (ct >> 1) * rbytes + (ct & 1) * 32;
        union { struct { s16x4_v a, b; } s; vec_t v; } u;
        u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_v*)(rowp + a0));
        u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_v*)(rowp + a1));
        qf[j] = u.v;
      }
#pragma unroll
      for (int j = 0; j < kStemCT; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[j][i] = V16<T>::mma(qf[j], pf[i], acc[j][i]);
    }
  }

  // ---- slab of this workgroup: ws[split][co][K], lane holds 4 consecutive columns of row co -------
  float* slab = p.ws + (long long)split * ((long long)p.Ncols * p.ldo);
#pragma unroll
  for (int j = 0; j < kStemCT; ++j) {
    const int ct = ct0 + j;
    if (ct >= ncts) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = i * 16 + l15;
      *reinterpret_cast<float4*>(slab + (long long)co * p.ldo + ct * 16 + g * 4) =
          make_float4(acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]);
    }
  }
}

// split-K slab reduction: a workgroup owns CL float4 columns (CL*16 contiguous bytes of every slab
// row); its G lane groups stride over the splits (4 loads in flight each) and fold through LDS; group
// 0 applies the epilogue.  Many splits (skinny weights: few columns, hundreds of slabs) use narrow
// 16-column workgroups so the grid still covers the chip.
// ws_b / dbias / nb (optional): the bias-gradient partial rows of the same split launch (GP::dbias: ws_b[split][nb] behind the
// weight slabs) are folded by the workgroups past the weight part of the grid, splits in order -- one launch per split WGRAD
// instead of two (the stand-alone wgrad_bias_reduce_kernel stays for the families whose column sums are a separate pass).
template <typename OutT, int G, int CL>
__global__ __launch_bounds__(CL * G) void wgrad_reduce_kernel(const float* ws, char* O, const float* rowscale,
                                                              long long n, int ldo, int splits, float alpha,
                                                              int accumulate, const float* __restrict__ ws_b = nullptr,
                                                              float* __restrict__ dbias = nullptr, int nb = 0, int wblocks = 0) {
  if (ws_b != nullptr && (int)blockIdx.x >= wblocks) {
    const int i = ((int)blockIdx.x - wblocks) * (CL * G) + (int)threadIdx.x;
    if (i < nb) {
      float s = 0.f;
      for (int k = 0; k < splits; ++k) s += ws_b[(long long)k * nb + i];
      dbias[i] = s * alpha * (rowscale ? rowscale[i] : 1.f);
    }
    return;
  }
  __shared__ float4 part[G > 1 ? (G - 1) * CL : 1];
  const int lane = threadIdx.x % CL, g = threadIdx.x / CL;
  const long long i = ((long long)blockIdx.x * CL + lane) * 4;
  float4 s = {0.f, 0.f, 0.f, 0.f};
  if (i < n) {
    const float* src = ws + i;
    int k = g;
    for (; k + 3 * G < splits; k += 4 * G) {
      float4 t0 = *reinterpret_cast<const float4*>(src + (long long)k * n);
      float4 t1 = *reinterpret_cast<const float4*>(src + (long long)(k + G) * n);
      float4 t2 = *reinterpret_cast<const float4*>(src + (long long)(k + 2 * G) * n);
      float4 t3 = *reinterpret_cast<const float4*>(src + (long long)(k + 3 * G) * n);
      s.x += (t0.x + t1.x) + (t2.x + t3.x); s.y += (t0.y + t1.y) + (t2.y + t3.y);
      s.z += (t0.z + t1.z) + (t2.z + t3.z); s.w += (t0.w + t1.w) + (t2.w + t3.w);
    }
    for (; k < splits; k += G) {
      float4 t = *reinterpret_cast<const float4*>(src + (long long)k * n);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
  }
  if (G > 1) {
    if (g > 0) part[(g - 1) * CL + lane] = s;
    __syncthreads();
    if (g == 0) {
#pragma unroll
      for (int j = 0; j < G - 1; ++j) {
        float4 t = part[j * CL + lane];
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
    }
  }
  if (g == 0 && i < n) {
    float rs = alpha * (rowscale ? rowscale[i / ldo] : 1.f);
    float v[4] = {s.x * rs, s.y * rs, s.z * rs, s.w * rs};
    if (accumulate) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += ld_elem<OutT>(O, i + r);
    }
    store4<OutT>(O, i, v, 4, true);
  }
}

// bias-gradient partial rows of a split WGRAD (GP::dbias): ws_b[split][n] -> out[n], splits folded in order
__global__ void wgrad_bias_reduce_kernel(const float* __restrict__ ws_b, float* __restrict__ out,
                                         const float* __restrict__ rowscale, int n, int splits, float alpha) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += ws_b[(long long)k * n + i];
  out[i] = s * alpha * (rowscale ? rowscale[i] : 1.f);
}

// ---- host side ------------------------------------------------------------------------------
int ilog2_exact(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return ((1 << l) == v) ? l : -1;
}

struct Plan {
  GP gp;
  bool ident, packw;
  int bm, bn;     // tile (NT: m x n; TN: p x q)
  int splits;
  int rb;         // NT tile-row bytes (64 or 128)
  int tn_tr;      // WGRAD: DMA + LDS transpose-read kernel (bf16)
  int pre;        // NT: prefetch residual / mask rows before the k-loop (thin-K, epilogue-bound launches)
  int threads;    // workgroup size (NT: 256 or 512)
  int ut;         // NT: taps span whole k-tiles (and DGRAD has unit stride): scalar tap cursor
  int stem;       // WGRAD: packed-stem kernel (whole output rows per workgroup, raw input rows in LDS)
  int rows;       // WGRAD: whole-row kernel for thin 64 -> 64 channel convs (vlfb_wgrad_rows.hip)
  int tn8;        // WGRAD: 256 x 256 phase-pipelined kernel (plain rows)
  int nt8;        // NT: 256-row phase-pipelined kernel with this tile width (256 / 128), 0 = 128x128 kernel
  int nt8_bm;     //     rows per tile: 256 or 196
  int nt8_mode;   //     0 plain rows, 1 gathered FPROP, 2 gathered unit-stride DGRAD
  int nts;        // NT: weight-resident streaming kernel (vlfb_gemm_s.hip); nts_mode as nt8_mode
  int rows64;     // FPROP / unit-stride DGRAD of 1x3x3 64 -> 64 convs: direct-convolution kernel (vlfb_conv_rows.hip)
  int stemf;      // FPROP of the packed stem: direct-convolution kernel (vlfb_stem.hip) when the call has no residual / mask
  int nts_mode;
  int sp;         // split-bf16 math (vlfb_gemm_split.hip): bf16 terms per operand (2 | 3), 0 = native MFMA of the dtype
  int skinny;     // NT: at most 64 plain rows (vlfb_gemm_skinny.hip)
  int skinny_sp;  //     ... with split-bf16 math on fp32 rows (two terms, fp32 output without a copy)
  int sp_kind;    //   NT: 0 plain rows, 1 gathered FPROP, 2 gathered DGRAD, 3 packed stem
  int sp_pl;      //   operands arrive as bf16 term planes (WGRAD: both; FPROP / DGRAD: the activation operand)
  int bias_fused; // WGRAD with desc.wgrad_bias: the launch itself produces the column sums of P (gemm_tn_tr_kernel)
  int h2;         // VLFB_MATH_F16X3: both operands as two fp16 planes, three fp16 MFMAs per product (vlfb_gemm_pair.hip)
  int w2i;        // VLFB_MATH_F16W2: unit-stride 16-bit DGRAD, two-term weights interleaved per 64-channel k-tile (gemm_nt_kernel<.., W2I>)
  size_t stem_lds;
  dim3 grid;
  size_t lds;
  long long ws_elems;
};

int make_plan(const vlfb_conv_desc* d, Plan* pl) {
  GP& g = pl->gp;
  ::memset(&g, 0, sizeof(g));
  VLFB_REQUIRE(dtype_ok(d->dtype), "conv: bad dtype %d", d->dtype);
  VLFB_REQUIRE(d->out_dtype == VLFB_F32 || d->out_dtype == d->dtype || (d->out_dtype == VLFB_F16 && d->math != VLFB_MATH_NATIVE), "conv: bad out_dtype");
  VLFB_REQUIRE(d->mode >= 0 && d->mode <= 2, "conv: bad mode %d", d->mode);
  VLFB_REQUIRE(d->math == VLFB_MATH_NATIVE || d->math == VLFB_MATH_BF16X3 || d->math == VLFB_MATH_BF16X6 || d->math == VLFB_MATH_F16X3 ||
                   d->math == VLFB_MATH_F16W2, "conv: bad math %d", d->math);
  pl->h2 = d->math == VLFB_MATH_F16X3;
  pl->w2i = d->math == VLFB_MATH_F16W2;
  VLFB_REQUIRE(!pl->w2i || (is16(d->dtype) && d->mode == VLFB_CONV_DGRAD && d->st == 1 && d->sh == 1 && d->sw == 1 && !d->pack_w &&
                            d->Cs % 64 == 0 && d->batch <= 1 && (d->algo == VLFB_ALGO_AUTO || d->algo == VLFB_ALGO_TILE128)),
               "conv: F16W2 math is the 16-bit DGRAD of a unit-stride conv with Cs %% 64 == 0 (two-term weights, VLFB_MIX_W2I)");
  VLFB_REQUIRE(!pl->h2 || (d->dtype == VLFB_F16 && d->mode == VLFB_CONV_FPROP && d->a_pstride > 0 && d->algo != VLFB_ALGO_STREAM &&
                           d->algo != VLFB_ALGO_CLASSES && d->algo != VLFB_ALGO_CLASS0),
               "conv: F16X3 math is an FPROP / NT product on fp16 planes (dtype VLFB_F16, a_pstride > 0)");
  // (split-bf16 FPROP / DGRAD with out_dtype VLFB_F16: the output -- and the residual, if any -- are two fp16 planes,
  // O / O_lo and R / R_lo of vlfb_conv_args: where an fp32 tensor enters the two-plane forward of the "mix" path)
  VLFB_REQUIRE(d->math == VLFB_MATH_NATIVE || pl->h2 || pl->w2i ||
                   (d->dtype == VLFB_F32 && (d->out_dtype == VLFB_F32 || (d->out_dtype == VLFB_F16 && d->mode != VLFB_CONV_WGRAD && !d->o_planes))),
               "conv: split-bf16 math needs fp32 operands and an fp32 (or two-plane fp16) output");
  VLFB_REQUIRE(d->math != VLFB_MATH_BF16X6 || d->mode != VLFB_CONV_WGRAD, "conv: WGRAD has no BF16X6 form (use BF16X3)");
  VLFB_REQUIRE(!d->accumulate || d->mode == VLFB_CONV_WGRAD, "conv: accumulate (O += ...) is a WGRAD epilogue");
  VLFB_REQUIRE(!d->wgrad_bias || d->mode == VLFB_CONV_WGRAD, "conv: wgrad_bias belongs to WGRAD descriptors");
  pl->sp = d->math == VLFB_MATH_BF16X6 ? 3 : d->math == VLFB_MATH_BF16X3 ? 2 : 0;     // (0 for F16X3 / F16W2)
  pl->sp_kind = 0;
  VLFB_REQUIRE(pl->sp || (!d->a_planes && !d->p_planes && !d->o_planes), "conv: a_planes / p_planes / o_planes belong to split-bf16 math");
  VLFB_REQUIRE(d->mode == VLFB_CONV_WGRAD ? (d->a_planes ? d->a_planes >= 2 && d->p_planes == 2 : d->p_planes == 0) && !d->o_planes
                                         : (d->a_planes == 0 || d->a_planes >= pl->sp) && !d->p_planes && (d->o_planes >= 0 && d->o_planes <= 2),
               "conv: bad a_planes / p_planes / o_planes for this mode");
  pl->sp_pl = d->a_planes > 0;
  // operands handed in as bf16 term planes are 2-byte elements for all address arithmetic below
  const int es = (d->dtype == VLFB_F32 && !pl->sp_pl) ? 4 : 2;
  const int epc = 16 / es;
  const int batch = d->batch > 0 ? d->batch : 1;
  VLFB_REQUIRE(d->N > 0 && d->Tr > 0 && d->Hr > 0 && d->Wr > 0, "conv: empty row space");
  VLFB_REQUIRE(d->Cs > 0 && d->Cn > 0, "conv: Cs/Cn must be positive");
  const long long M = (long long)d->N * d->Tr * d->Hr * d->Wr;
  VLFB_REQUIRE(M < (1ll << 31), "conv: too many rows");
  const int taps = d->kt * d->kh * d->kw;
  VLFB_REQUIRE(taps >= 1, "conv: bad kernel size");
  pl->packw = d->pack_w != 0;
  long long K;
  int cpt;  // 16-byte chunks per tap
  if (pl->packw) {
    VLFB_REQUIRE(d->Cs == 4 && d->dw == 1 && d->pack_w >= d->kw && ilog2_exact(d->pack_w) >= 0,
                 "conv: pack_w needs Cs==4, dw==1 and a power-of-two kw_pad >= kw");
    VLFB_REQUIRE(d->mode != VLFB_CONV_DGRAD, "conv: pack_w has no DGRAD");
    K = (long long)d->kt * d->kh * d->pack_w * 4;
    cpt = d->pack_w * 4 / epc;
  } else {
    VLFB_REQUIRE(d->Cs % epc == 0, "conv: Cs=%d must be a multiple of %d", d->Cs, epc);
    K = (long long)taps * d->Cs;
    cpt = d->Cs / epc;
  }
  pl->ident = !pl->packw && taps == 1 && d->st == 1 && d->sh == 1 && d->sw == 1 && d->pt == 0 &&
              d->ph == 0 && d->pw == 0 && d->Ts == d->Tr && d->Hs == d->Hr && d->Ws == d->Wr;
  if (pl->w2i) pl->ident = false;        // (a 1x1x1 conv walks the tap cursor over its one tap)
  if (!pl->ident) {
    VLFB_REQUIRE(ilog2_exact(cpt) >= 0, "conv: channels per tap must give a power-of-two chunk count");
    VLFB_REQUIRE(ilog2_exact(d->st) >= 0 && ilog2_exact(d->sh) >= 0 && ilog2_exact(d->sw) >= 0,
                 "conv: strides must be powers of two");
    VLFB_REQUIRE(batch == 1, "conv: batched launches must be plain GEMMs");
  }
  g.M = (int)M; g.Ncols = d->Cn; g.K = (int)K;
  g.Tr = d->Tr; g.Hr = d->Hr; g.Wr = d->Wr; g.Ts = d->Ts; g.Hs = d->Hs; g.Ws = d->Ws; g.Cs = d->Cs;
  g.kt = d->kt; g.kh = d->kh; g.kw = d->kw;
  g.inv_khw = 1.0f / (float)(d->kh * d->kw); g.inv_kw = 1.0f / (float)d->kw; g.inv_kh = 1.0f / (float)d->kh;
  g.st = d->st; g.sh = d->sh; g.sw = d->sw; g.pt = d->pt; g.ph = d->ph; g.pw = d->pw;
  g.dt = d->dt; g.dh = d->dh; g.dw = d->dw;
  g.lst = ilog2_exact(d->st); g.lsh = ilog2_exact(d->sh); g.lsw = ilog2_exact(d->sw);
  g.cpt_shift = pl->ident ? 0 : ilog2_exact(cpt);
  g.lda = d->lda ? d->lda : d->Cs;
  g.ldb = d->ldb ? d->ldb : (int)(pl->w2i ? 2 * K : K);      // (F16W2: a weight row holds both terms)
  g.ldp = d->ldp ? d->ldp : d->Cn;
  g.ldo = d->ldo ? d->ldo : (d->mode == VLFB_CONV_WGRAD ? (int)K : d->Cn);
  g.ldr = d->ldr ? d->ldr : g.ldo;
  g.a_bs = d->a_bstride; g.b_bs = d->b_bstride; g.o_bs = d->o_bstride; g.r_bs = d->r_bstride;
  g.p_bs = d->p_bstride;
  g.alpha = d->alpha; g.relu = d->relu; g.bias_mode = d->bias_mode; g.accumulate = d->accumulate;
  g.s2 = 0; g.s2_mq = 0; g.s2_tpc = 0;
  g.a_ps = d->a_pstride; g.p_ps = d->p_pstride; g.o_ps = d->o_pstride; g.op_n = d->o_planes; g.OP = nullptr;
  VLFB_REQUIRE((pl->packw || g.lda % epc == 0) && (d->mode == VLFB_CONV_WGRAD || g.ldb % epc == 0) &&
                   (d->mode != VLFB_CONV_WGRAD || g.ldp % epc == 0),
               "conv: leading dimensions must keep 16-byte alignment");

  pl->splits = 1;
  pl->ws_elems = 0;
  pl->tn8 = 0;
  pl->tn_tr = 0;
  pl->stem = 0;
  pl->rows = 0;
  if (d->mode != VLFB_CONV_WGRAD) {
    pl->bm = 128;
    pl->bn = d->Cn > 64 ? 128 : 64;
    g.tiles_m = (int)((M + pl->bm - 1) / pl->bm);
    g.tiles_n = (d->Cn + pl->bn - 1) / pl->bn;
    pl->grid = dim3((unsigned)(g.tiles_m * g.tiles_n), 1, (unsigned)batch);
    const int ept = d->out_dtype == VLFB_F32 ? 4 : 8;   // elements per 16-byte output store
    g.vec_epi = (d->Cn % ept == 0) && (g.ldo % ept == 0) && (g.ldr % ept == 0) &&
                (d->o_bstride % ept == 0) && (d->r_bstride % ept == 0);
    // DGRAD of a (1, 2, 2)-strided conv: three quarters of the (row, tap) pairs are structural zeros (an input
    // position only meets the taps of its own parity).  Rows enumerated class by class make every tile class-pure,
    // and a tile then walks only its class's taps: 9 -> 1 / 2 / 2 / 4 taps for the 3x3 convs of res3_0 / res4_0,
    // Measured at 8 clips (scratch/nts_probe.cpp): res3_0 2b 161 -> 110 us, res4_0 2b 156 -> 95 us.  NOT for the 1x1x1
    // shortcuts, where one class would do the whole GEMM and the other three only the epilogue: their residual /
    // output rows are then visited as 256-byte pieces of four different passes over the tensor instead of one
    // stream (198 -> 248 us, 153 -> 175 us).  Classes over h only (odd lines epilogue-only, as whole contiguous lines) are
    // no better (246 / 167 us): a tile without a k-loop has nothing to hide its residual loads behind.
    // (dt == 0: the doubled term dimension of two-term fp16 weights, VLFB_MIX_W2 -- the walk is the same per term)
    if (d->mode == VLFB_CONV_DGRAD && d->algo == VLFB_ALGO_AUTO && !pl->sp && !pl->ident && !pl->packw && batch == 1 && d->kh * d->kw > 1 &&
        d->st == 1 && d->sh == 2 && d->sw == 2 && (d->dt == 1 || d->dt == 0) && d->dh == 1 && d->dw == 1 && d->Hr % 2 == 0 &&
        d->Wr % 2 == 0 && ((long long)d->Cs * es) % 128 == 0 && d->bias_mode == VLFB_BIAS_NONE) {
      g.s2 = 1;
      g.s2_mq = (int)(M / 4);
      g.s2_tpc = (g.s2_mq + pl->bm - 1) / pl->bm;
      g.tiles_m = 4 * g.s2_tpc;
      pl->grid = dim3((unsigned)(g.tiles_m * g.tiles_n), 1, 1);
    }
    // The strided 1x1x1 shortcut (VLFB_ALGO_CLASS0): only class (0, 0) meets its one tap, so ONLY that class's tiles are
    // launched -- a quarter of the rows, the whole k-loop, no epilogue-only tiles -- and the other rows of O stay as the
    // caller left them (the engine runs this DGRAD as the SECOND contribution to the block-input gradient, in place on the
    // first).  The "classes" form above lost on these convs because three of four tiles were epilogue-only passes.
    if (d->algo == VLFB_ALGO_CLASS0) {
      const bool ok = d->mode == VLFB_CONV_DGRAD && !pl->sp && !pl->ident && !pl->packw && batch == 1 && d->kh * d->kw == 1 &&
                      is16(d->dtype) && d->out_dtype == d->dtype && d->st == 1 && d->sh == 2 && d->sw == 2 && d->ph == 0 &&
                      d->pw == 0 && (d->dt == 1 || d->dt == 0) && d->Hr % 2 == 0 && d->Wr % 2 == 0 &&
                      ((long long)d->Cs * es) % 128 == 0 && d->bias_mode == VLFB_BIAS_NONE && !d->relu && g.vec_epi;
      VLFB_REQUIRE(ok, "conv: algo = CLASS0 is the 16-bit DGRAD of an unpadded (1, 2, 2)-strided 1x1x1 conv (even H, W; whole 128-byte taps)");
      g.s2 = 1;
      g.s2_mq = (int)(M / 4);
      g.s2_tpc = (g.s2_mq + pl->bm - 1) / pl->bm;
      g.tiles_m = g.s2_tpc;                       // class (0, 0) only
      pl->grid = dim3((unsigned)(g.tiles_m * g.tiles_n), 1, 1);
    }
    // The split-bf16 form of the same walk (gemm_nt_sp_kernel<.., S2>): a scalar tap cursor over the class's taps instead
    // of the per-lane decode.  Measured in the step (8 clips): res3_0 2b 502 -> 321 us, res4_0 2b 486 -> 240 us.  The 1x1x1
    // shortcuts stay on the plain walk here too (VLFB_SPLIT_S2_1X1=1 to try: 466 -> 513 us, 386 -> 380 us).
    static const bool s2_sp_off = getenv("VLFB_SPLIT_S2") && atoi(getenv("VLFB_SPLIT_S2")) == 0;
    static const bool s2_sp_1x1_off = !(getenv("VLFB_SPLIT_S2_1X1") && atoi(getenv("VLFB_SPLIT_S2_1X1")) == 1);
    const bool s2_sp_want = d->algo == VLFB_ALGO_CLASSES || (d->algo == VLFB_ALGO_AUTO && !s2_sp_off && (d->kh * d->kw > 1 || !s2_sp_1x1_off));
    if (d->mode == VLFB_CONV_DGRAD && s2_sp_want && pl->sp == 2 && !pl->sp_pl && !pl->ident && !pl->packw &&
        batch == 1 && d->st == 1 && d->sh == 2 && d->sw == 2 && d->dt == 1 && d->dh == 1 &&
        d->dw == 1 && d->Hr % 2 == 0 && d->Wr % 2 == 0 && d->Cs % 32 == 0 && d->bias_mode == VLFB_BIAS_NONE) {
      g.s2 = 1;
      g.s2_mq = (int)(M / 4);
      g.s2_tpc = (g.s2_mq + pl->bm - 1) / pl->bm;
      g.tiles_m = 4 * g.s2_tpc;
      pl->grid = dim3((unsigned)(g.tiles_m * g.tiles_n), 1, 1);
    }
    VLFB_REQUIRE(d->algo != VLFB_ALGO_CLASSES || g.s2, "conv: algo = CLASSES is the split-math DGRAD of a (1, 2, 2)-strided conv (even H, W; Cs % 32 == 0)");
  } else {
    pl->bm = d->Cn > 64 ? 128 : 64;             // P tile (output rows)
    pl->bn = K > 64 ? 128 : 64;                 // Q tile (output columns)
    // 256 x 256 phase-pipelined kernel (vlfb_gemm8.hip): plain-row bf16 operands with at least 128 of each
    {
      const bool ok = is16(d->dtype) && pl->ident && d->Cn % 8 == 0 && K % 8 == 0 && d->Cn >= 128 && K >= 128 &&
                      g.lda % 8 == 0 && g.ldp % 8 == 0;
      if (d->algo == VLFB_ALGO_PIPE256)
        VLFB_REQUIRE(ok, "conv: algo = PIPE256 WGRAD needs bf16 / f16 plain-row operands with Cn, K >= 128 (multiples of 8)");
      // library choice: every workgroup writes a 256 KiB fp32 slab, so only the large weights win
      // (Cn * K >= 1 M elements: res5 1x1x1 wgrads 1.3-1.66x; 0.5 M and below 0.6-0.97x, scratch/nt8_probe.cpp)
      pl->tn8 = ok && d->algo != VLFB_ALGO_TILE128 &&
                (d->algo == VLFB_ALGO_PIPE256 || (batch == 1 && (long long)d->Cn * K >= (1ll << 20) && d->Cn >= 512 && K >= 512));
      if (pl->tn8) pl->bm = pl->bn = 256;
    }
    // few output rows (res2 / stem, Cout = 64): widen the Q tile so a workgroup still has
    // 32 MFMAs per wave per k-tile of staging and the P panel is re-read half as often
    pl->tn_tr = is16(d->dtype) && d->Cn % 8 == 0;
    if (pl->tn8) pl->tn_tr = 1;
    if (pl->sp_pl) {       // two-plane operands: the DMA + transposed-read kernel in its SP form
      VLFB_REQUIRE(d->Cn % 8 == 0 && (g.lda % 8 == 0 || pl->packw) && g.ldp % 8 == 0 && d->a_pstride % 8 == 0 && d->p_pstride % 8 == 0,
                   "conv: plane operands need channel counts / strides in multiples of 8");
      pl->tn_tr = 1;
    }
    if (!pl->tn_tr && pl->bm == 64 && K >= 256 && is16(d->dtype)) pl->bn = 256;
    if (pl->tn_tr && !pl->sp && pl->packw && d->Cn == 64 && d->pack_w == 8 && d->Wr % 8 == 0 && d->Wr <= 128 &&
        d->dt == 1 && d->dh == 1 && d->splits <= 0 && batch == 1 && g.ldo == (int)K && (d->Ws * 8) % 16 == 0) {
      const int taps_ab = d->kt * d->kh;
      const long long npieces = (long long)taps_ab * (d->Ws * 8 / 16);
      const long long stage = npieces * 16 + (long long)d->Wr * 128;
      const long long tiles_total = (long long)d->N * d->Tr * d->Hr;
      if (npieces <= 4096 && taps_ab * 2 <= 8 * kStemCT && 2 * stage <= 160 * 1024 && tiles_total >= 2) {
        const long long wgs = tiles_total < 256 ? tiles_total : 256;          // one workgroup per CU
        const long long tpw = (tiles_total + wgs - 1) / wgs;
        const int splits = (int)((tiles_total + tpw - 1) / tpw);
        g.tiles_m = (int)tiles_total;
        g.tiles_n = 1;
        g.kper = (int)tpw;
        g.splits = splits;
        pl->splits = splits;
        pl->ws_elems = (long long)splits * d->Cn * K;
        pl->stem = 1;
        pl->stem_lds = (size_t)(2 * stage);
      }
    }
    // whole-row kernel (vlfb_wgrad_rows.hip): 64 -> 64 channels, unit stride, same-size output, rows of <= 64
    // positions (res2 3x3 / 3x1x1): every operand byte is read once, taps are LDS row offsets
    if (!pl->stem && pl->tn_tr && !pl->sp && !pl->packw && !pl->ident && d->algo == VLFB_ALGO_AUTO && d->Cs == 64 && d->Cn == 64 &&
        d->st == 1 && d->sh == 1 && d->sw == 1 && d->dt == 1 && d->dh == 1 && d->dw == 1 && d->Tr == d->Ts &&
        d->Hr == d->Hs && d->Wr == d->Ws && d->Wr % 8 == 0 && d->Ws + d->kw - 1 <= 64 && d->pw < d->kw &&
        wgrad_rows_ct(K) > 0 && d->kt * d->kh == 3 && d->splits <= 0 && batch == 1 && g.ldo == (int)K && g.lda == 64 && g.ldp == 64) {
      const long long tiles_total = (long long)d->N * d->Tr * d->Hr;
      if (tiles_total >= 2) {
        const long long wgs = tiles_total < 256 ? tiles_total : 256;            // one 128 KiB workgroup per CU
        const long long tpw = (tiles_total + wgs - 1) / wgs;
        const int splits = (int)((tiles_total + tpw - 1) / tpw);
        g.tiles_m = (int)tiles_total;
        g.tiles_n = 1;
        g.kper = (int)tpw;
        g.splits = splits;
        pl->splits = splits;
        pl->ws_elems = (long long)splits * d->Cn * K;
        pl->rows = 1;
        pl->stem_lds = (size_t)2 * ((size_t)d->kt * d->kh * (d->Ws + d->kw - 1) * 128 + (size_t)d->Wr * 128) + 1024;
      }
    }
    if (!pl->stem && !pl->rows && pl->tn_tr && !pl->sp && !pl->packw && !pl->ident && d->algo == VLFB_ALGO_AUTO && d->Cs == 256 &&
        d->Cn == 64 && d->kt == 3 && d->kh == 1 && d->kw == 1 && d->st == 1 && d->sh == 1 && d->sw == 1 && d->dt == 1 &&
        d->Tr == d->Ts && d->Hr == d->Hs && d->Wr == d->Ws && d->ph == 0 && d->pw == 0 && d->pt < 3 && d->Wr % 8 == 0 &&
        d->Wr <= 64 && d->splits <= 0 && batch == 1 && g.ldo == (int)K && g.lda == 256 && g.ldp == 64) {
      const long long tiles_total = (long long)d->N * d->Ts * d->Hs;
      if (tiles_total >= 2) {
        const long long wgs = tiles_total < 256 ? tiles_total : 256;
        const long long tpw = (tiles_total + wgs - 1) / wgs;
        const int splits = (int)((tiles_total + tpw - 1) / tpw);
        g.tiles_m = (int)tiles_total;
        g.tiles_n = 1;
        g.kper = (int)tpw;
        g.splits = splits;
        pl->splits = splits;
        pl->ws_elems = (long long)splits * d->Cn * K;
        pl->rows = 2;
        pl->stem_lds = 0;
      }
    }
    if (!pl->stem && !pl->rows) {
      g.tiles_m = (d->Cn + pl->bm - 1) / pl->bm;
      g.tiles_n = (int)((K + pl->bn - 1) / pl->bn);
      const int bk = pl->sp_pl ? 32 : 128 / es;          // positions per k-tile of the kernel that will run
      int splits = d->splits;
      if (splits <= 0) {
        // Pick the split count that fills whole rounds of `slots` workgroups best (every extra split
        // costs one more fp32 slab pass), keeping at least 8 k-tiles of work per split.  Two 64 KiB-LDS
        // workgroups fit a CU (512 slots), but these launches share the chip with the dgrad chain of
        // the main stream: rounds of 256 (one workgroup per CU, half the slab traffic) measured best
        // end to end (336 vs 331 clips/s at 512, 323 at 128).
        const long long tiles = (long long)g.tiles_m * g.tiles_n * batch;
        // (split-bf16 math: one 4-wave workgroup per CU leaves every SIMD with a single wave, whose staging and MFMA
        // phases then run back to back; two per CU -- 2 x 64 KiB of LDS -- overlap them)
        // (re-measured in round 3 on the bf16 path, one box: 256 -> 448.7, 384 -> 436.8, 512 -> 429.9, 768 -> 426.0 clips/s)
        const long long slots = (pl->sp && pl->bm > 64) ? 512 : 256;   // (the 64-row tiles of res2 measured slower at 512)
        long long maxs = (M + 8 * bk - 1) / (8 * bk);
        const long long slab_cap = (96ll << 20) / ((long long)d->Cn * K * 4);   // <= 96 MiB of fp32 slabs
        if (maxs > slab_cap) maxs = slab_cap;
        if (maxs > 1024) maxs = 1024;
        if (batch > 1 || maxs < 1) maxs = 1;
        double best = -1.0;
        splits = 1;
        for (long long sp = 1; sp <= maxs; sp = (sp < 8 ? sp + 1 : sp + 8)) {   // 1..8, then multiples of 8
          const long long total = tiles * sp;
          const double eff = (double)total / (double)(((total + slots - 1) / slots) * slots);
          if (eff > best + 0.03) { best = eff; splits = (int)sp; }
        }
      }
      VLFB_REQUIRE(splits == 1 || batch == 1, "conv: split WGRAD cannot be batched");
      long long kper = (M + splits - 1) / splits;
      kper = (kper + bk - 1) / bk * bk;
      splits = (int)((M + kper - 1) / kper);
      g.kper = (int)kper;
      g.splits = splits;
      pl->splits = splits;
      if (splits > 1) {
        VLFB_REQUIRE(g.ldo == (int)K, "conv: split WGRAD needs a dense output (ldo == K)");
        pl->ws_elems = (long long)splits * d->Cn * K;
      }
      if (splits > 1 && splits % 8 == 0)
        pl->grid = dim3((unsigned)(g.tiles_m * g.tiles_n * splits), 1, 1);
      else
        pl->grid = dim3((unsigned)(g.tiles_m * g.tiles_n), (unsigned)splits, (unsigned)batch);
    } else {
      pl->grid = dim3((unsigned)pl->splits, 1, 1);
    }
  }
  pl->nt8 = 0;
  pl->nt8_mode = 0;
  if (d->mode != VLFB_CONV_WGRAD && d->algo != VLFB_ALGO_TILE128) {
    // 256-row phase-pipelined kernel (vlfb_gemm8.hip): bf16, 16-byte epilogue legal, at least 128 output
    // channels and 128 k; gathered operands need taps that span whole 64-element k-tiles (and unit stride
    // for DGRAD) and at most 32 taps (one validity bit per tap and row)
    // (two fp16 planes, pl->h2: a k-tile is 32 k of both planes -- taps of whole 32-channel runs; FPROP / plain rows only)
    const bool gather_ok = pl->ident || (!d->pack_w && ((long long)d->Cs * es) % (pl->h2 ? 64 : 128) == 0 && taps <= 32 &&
                                         (d->mode == VLFB_CONV_FPROP || (d->st == 1 && d->sh == 1 && d->sw == 1)));
    const bool ok = is16(d->dtype) && g.vec_epi && gather_ok && d->Cn >= 128 && K >= 128 && M >= 1024 && !pl->w2i &&
                    (!pl->h2 || (K % 32 == 0 && batch == 1));
    if (d->algo == VLFB_ALGO_PIPE256)
      VLFB_REQUIRE(ok, "conv: algo = PIPE256 needs bf16 / f16, Cn >= 128, K >= 128, M >= 1024, 16-byte aligned rows and "
                       "taps spanning whole k-tiles");
    // Library choice (measured per layer on MI355X at the 8-clip shapes, scratch/nt8_probe.cpp, tables in
    // profiles/): with ONE 128-160 KiB workgroup per CU the prologue and the epilogue of a tile are exposed,
    // so the pipelined kernel only wins where the k-loop is long and the tile count fills whole rounds of
    // CUs without a heavy epilogue: 512-column outputs with K >= 1024 (res5 3x3 / 3x1x1 / 1x1x1, the
    // non-local theta conv: 1.07-1.20x) and the batched P.g products of the non-local blocks (1.15-1.20x).
    // Elsewhere (Cn = 2048 with residual + mask epilogues, K <= 512, the res3 / res4 shapes whose tiles
    // fill half the chip) the 128x128 kernel with 2-3 co-resident workgroups is 1.1-1.6x faster.
    static const int pair8 = getenv("VLFB_PAIR_PIPE256") ? atoi(getenv("VLFB_PAIR_PIPE256")) : -1;     // (A/B switch: 0 never, 1 every eligible launch)
    // Two fp16 planes (pl->h2): 24 MFMAs per phase on the fragment reads and DMA pieces of the plain form's 16, so the
    // pipelined kernel pays on more shapes (measured per launch at 8 clips, scratch/r6/pair_probe.py: K >= 512 0.74-0.95x the
    // time of the 128-row kernel -- res5 3x3 361 -> 270 us = 1.31 PFLOP/s of MFMA issue, res5 3x1x1 494 -> 364 us -- except
    // the 2048-column layers with K = 512, whose residual epilogue dominates: 231 -> 250 us; K <= 256 0.94-1.36x)
    const bool want_h2 = pl->h2 && ok && (pair8 == 1 || (pair8 != 0 && K >= 512 && (d->Cn <= 1024 || K >= 1024)));
    const bool want = d->algo == VLFB_ALGO_PIPE256 || want_h2 ||
                      (ok && !pl->h2 && ((d->Cn == 512 && K >= 1024) ||
                              (batch > 1 && K >= 768 && d->Cn >= 256 && d->Cn <= 512 && d->out_dtype == d->dtype)));
    if (ok && want) {
      // tile shape: 256 / 196 rows (196 = two wave rows of 98: 7 of 8 fragment rows useful) x 256 / 128 columns,
      // whichever needs the fewest MFMA slots over whole rounds of 256 workgroups (one per CU)
      long long best = -1;
      for (int bn = 256; bn >= 128; bn -= 128)
        for (int bm = 256; bm >= 196; bm -= 60) {
          const long long tn = (d->Cn + bn - 1) / bn, tm = (M + bm - 1) / bm;
          const long long cost = (tn * tm * batch + 255) / 256 * (bm == 196 ? 7 : 8) * (bn / 128);
          if (best < 0 || cost < best) { best = cost; pl->nt8 = bn; pl->nt8_bm = bm; }
        }
      pl->nt8_mode = pl->ident ? 0 : (d->mode == VLFB_CONV_FPROP ? 1 : 2);
      g.tiles_n = (d->Cn + pl->nt8 - 1) / pl->nt8;
      g.tiles_m = (int)((M + pl->nt8_bm - 1) / pl->nt8_bm);
    }
  }
  pl->nts = 0;
  pl->nts_mode = 0;
  if (d->mode != VLFB_CONV_WGRAD && d->algo != VLFB_ALGO_TILE128 && d->algo != VLFB_ALGO_PIPE256) {
    // weight-resident streaming kernel (vlfb_gemm_s.hip): the whole weight operand (<= ~150 KiB) lives in LDS,
    // every wave streams its own 16 / 32-position blocks without workgroup barriers -- for the HBM-bound layers
    const int mode = pl->ident ? 0 : (d->mode == VLFB_CONV_FPROP ? 1 : 2);
    const bool gather_ok = pl->ident || (!d->pack_w && taps <= 32 &&
                                         (d->mode == VLFB_CONV_FPROP || (d->st == 1 && d->sh == 1 && d->sw == 1)));
    const int uk = nts_chunk(mode, K);
    const bool shape_ok = (d->Cn == 64 || d->Cn == 128 || d->Cn == 256) && d->Cs % 64 == 0 && uk > 0 && M < (1ll << 24);
    const bool ok = is16(d->dtype) && !pl->h2 && !pl->w2i && d->out_dtype == d->dtype && batch == 1 && shape_ok && gather_ok &&
                    g.lda % 8 == 0 && g.ldb % 8 == 0 && g.ldo % 8 == 0 && g.ldr % 8 == 0 &&
                    (d->bias_mode == VLFB_BIAS_NONE || d->bias_mode == VLFB_BIAS_COL) &&
                    (long long)d->Cn * K * 2 + d->Cn * 4 <= 156 * 1024 &&
                    M * g.ldo * 2 < (1ll << 31) && M * g.ldr * 2 < (1ll << 31);
    if (d->algo == VLFB_ALGO_STREAM)
      VLFB_REQUIRE(ok, "conv: algo = STREAM needs bf16 / f16 in and out, batch 1, Cn in {64, 128, 256}, Cs %% 64 == 0, a weight "
                       "operand of at most 156 KiB, fewer than 2^24 rows, 16-byte aligned rows and operands below 2 GiB");
    // Library choice (measured at the 8-clip shapes, scratch/nts_probe.cpp and bench.py --detail): the 256-column
    // layers of res2 (0.8 M positions: 2c / shortcut fprop 1.04-1.08x alone, the 3x1x1 dgrad with residual + mask
    // 1.19x alone and 1.4x under the concurrent wgrad stream).  The 64-column variants are VALU-bound by the
    // per-block row decode and lose to the tiled kernel, res3 (0.1 M positions) has 3 blocks per wave.
    const bool want = d->algo == VLFB_ALGO_STREAM || (ok && M >= 400000 && d->Cn == 256);
    if (ok && want) { pl->nts = 1; pl->nts_mode = mode; pl->nt8 = 0; }
  }
  // packed stem FPROP: direct convolution, whole output rows per wave (vlfb_stem.hip)
  pl->stemf = d->mode == VLFB_CONV_FPROP && pl->packw && d->algo == VLFB_ALGO_AUTO && !pl->h2 &&
              (d->bias_mode == VLFB_BIAS_NONE || d->bias_mode == VLFB_BIAS_COL) &&
              stem_fprop_ok(g, d->pack_w, d->dtype, d->out_dtype, batch);
  // ... and its two-plane form (fp16 planes in and out; one 134-KB stage, vlfb_stem.hip)
  static const bool pair_stem_off = getenv("VLFB_PAIR_STEM_DIRECT") && atoi(getenv("VLFB_PAIR_STEM_DIRECT")) == 0;     // (A/B switch)
  if (pl->h2 && !pair_stem_off && d->mode == VLFB_CONV_FPROP && pl->packw && d->algo == VLFB_ALGO_AUTO && d->out_dtype == VLFB_F16 &&
      (d->bias_mode == VLFB_BIAS_NONE || d->bias_mode == VLFB_BIAS_COL) && stem_fprop_pair_ok(g, d->pack_w, batch))
    pl->stemf = 1;
  pl->rows64 = d->mode != VLFB_CONV_WGRAD && !pl->packw && !pl->ident && d->algo == VLFB_ALGO_AUTO && !pl->h2 && !pl->w2i &&
               (d->bias_mode == VLFB_BIAS_NONE || d->bias_mode == VLFB_BIAS_COL) &&
               conv_rows64_ok(g, d->mode, d->dtype, d->out_dtype, batch);
  // a handful of plain rows (the FBO head on one row per RoI): 16-column workgroups whose waves split K
  static const bool skinny_off = getenv("VLFB_SKINNY") && atoi(getenv("VLFB_SKINNY")) == 0;      // (A/B switch)
  pl->skinny = d->mode != VLFB_CONV_WGRAD && d->algo == VLFB_ALGO_AUTO && !skinny_off && !pl->sp && !pl->h2 && !pl->w2i && !pl->rows64 &&
               (d->out_dtype == d->dtype || d->out_dtype == VLFB_F32) && skinny_nt_ok(g, d->dtype, batch, pl->ident);
  // ... and the same rows in the fp32 head of the "mix" / "split" paths: two-term split-bf16 products (FPROP and DGRAD of the
  // FBO convs on one row per RoI: 26-29 us each on four 128 x 128 workgroups, 77 us for K = 2048)
  pl->skinny_sp = d->mode != VLFB_CONV_WGRAD && d->algo == VLFB_ALGO_AUTO && !skinny_off && pl->sp == 2 && !pl->sp_pl &&
                  d->out_dtype == VLFB_F32 && d->o_planes <= 1 && skinny_nt_split_ok(g, batch, pl->ident);
  pl->rb = 128;    // (64-byte tile rows were measured slower: 314 vs 348 TFLOP/s at the time, twice the barriers)
  pl->pre = 0;
  pl->threads = kThreads;
  if (d->mode == VLFB_CONV_WGRAD && pl->tn_tr && pl->bm == 128 && pl->bn == 128) {
    pl->threads = 512;       // 8 waves per workgroup
  }
  if (d->mode != VLFB_CONV_WGRAD && is16(d->dtype)) pl->threads = 512;   // 8 waves (2 x 4), both tile widths
  {
    // extents behind the buffer descriptors of the DMA kernels (one batch element)
    const long long a_rows = pl->ident ? M : (long long)d->N * d->Ts * d->Hs * d->Ws;
    long long a_bytes = a_rows * (pl->packw ? 4 : g.lda) * es;
    long long b_bytes = d->mode == VLFB_CONV_WGRAD ? M * g.ldp * es : (long long)d->Cn * g.ldb * es;
    if (pl->sp && d->mode != VLFB_CONV_WGRAD) {
      // the weight operand is pl->sp bf16 planes b_ps elements apart; one descriptor spans all of them
      g.b_ps = d->b_pstride > 0 ? d->b_pstride : (long long)batch * (batch > 1 ? d->b_bstride : (long long)d->Cn * g.ldb);
      VLFB_REQUIRE(K % 8 == 0 && g.ldb % 8 == 0 && g.b_ps % 8 == 0 && d->b_bstride % 8 == 0,
                   "conv: split-bf16 math needs K, ldb and the plane / batch strides of B in multiples of 8");
      VLFB_REQUIRE(g.vec_epi, "conv: split-bf16 math needs 16-byte aligned output rows (Cn, ldo, ldr multiples of 4)");
      b_bytes = ((long long)(pl->sp - 1) * g.b_ps + (long long)d->Cn * g.ldb) * 2;
    }
    long long a_bytes_all = a_bytes;
    if (pl->h2) {
      // (batched plain products -- the attention scores of a non-local block, theta x phi^T: the planes of ALL batch elements lie
      // a_pstride / b_pstride apart, a batch element a_bstride / b_bstride inside its plane)
      g.b_ps = d->b_pstride > 0 ? d->b_pstride : (long long)batch * (batch > 1 ? d->b_bstride : (long long)d->Cn * g.ldb);
      VLFB_REQUIRE(K % 8 == 0 && g.ldb % 8 == 0 && (pl->packw || g.lda % 8 == 0) && g.b_ps % 8 == 0 && d->a_pstride % 8 == 0 && g.vec_epi &&
                       d->a_bstride % 8 == 0 && d->b_bstride % 8 == 0,
                   "conv: F16X3 math needs K, lda, ldb, the plane / batch strides and the output rows in multiples of 8 elements");
      a_bytes_all = g.a_ps * 2 + a_bytes;          // one descriptor spans both planes
      b_bytes = (g.b_ps + (long long)d->Cn * g.ldb) * 2;
    }
    a_bytes = a_bytes_all;
    VLFB_REQUIRE(a_bytes < (1ll << 31) && b_bytes < (1ll << 31),
                 "conv: an operand of %lld / %lld bytes exceeds the 2 GiB a buffer descriptor addresses; split the batch",
                 a_bytes, b_bytes);
    g.a_bytes = (unsigned)a_bytes;
    g.b_bytes = (unsigned)b_bytes;
  }
  pl->ut = 0;
  if (d->mode != VLFB_CONV_WGRAD) {
    pl->ut = !pl->ident && !d->pack_w && ((pl->sp || pl->h2) ? d->Cs % 32 == 0 : ((long long)d->Cs * es) % pl->rb == 0) &&
             (d->mode == VLFB_CONV_FPROP || (d->st == 1 && d->sh == 1 && d->sw == 1));
    const long long ktiles = pl->h2 ? (K + 31) / 32 : (pl->w2i ? 2 : 1) * ((K * es + pl->rb - 1) / pl->rb);   // (two planes: a 128-byte row is 32 k)
    if (pl->h2) {
      if (pl->packw && d->pack_w == 8) {
        // the packed stem as a kw = 1 conv of 32 "channels" per (a, b) tap row (see the split-bf16 form below)
        g.kw = 1; g.Cs = d->pack_w * 4; g.lda = 4;
        pl->ut = 1;
      }
      VLFB_REQUIRE(pl->ident || pl->ut, "conv: F16X3 math needs plain rows or taps that span whole 32-element k-tiles");
    }
    VLFB_REQUIRE(!pl->w2i || (pl->ut && g.vec_epi), "conv: F16W2 math needs taps of whole 64-channel runs and 16-byte aligned output rows");
    const size_t buf = (size_t)(pl->bm + pl->bn) * pl->rb;
    pl->lds = (ktiles <= 1 ? 1 : 2) * buf;           // a single k-tile needs no second buffer
    const size_t tile = (size_t)pl->bm * pl->bn * 4;
    g.epi = (int)((tile + pl->lds - 1) / pl->lds);
    if (g.epi > 2) { pl->lds = tile / 2; g.epi = 2; }
    // (re-measured in round 3: prefetching for <= 16 / 36 / all k-tiles instead of 8 moves the bf16 step by -0.1 .. -0.7 %)
    static const int pre_kt = getenv("VLFB_PAIR_PRE_KT") ? atoi(getenv("VLFB_PAIR_PRE_KT")) : 16;       // (A/B switch)
    pl->pre = g.vec_epi && ktiles <= (pl->h2 ? pre_kt : 8);   // host decides; only launches with R / Mask use it (two planes: 32-k tiles)
    if (pl->sp) {
      pl->sp_kind = pl->ident ? 0 : pl->packw ? 3 : d->mode == VLFB_CONV_FPROP ? 1 : 2;
      pl->threads = kThreads;
      pl->pre = 0;
      if (pl->packw && d->mode == VLFB_CONV_FPROP && d->pack_w == 8) {
        // The packed stem at k-tiles of 32 elements: one k-tile IS one (a, b) tap row (8 kw pixels x 4 channels), so the
        // gather is the scalar-cursor one of a conv with kw = 1, 32 "channels" per tap and 4 elements per pixel -- no
        // per-lane tap decode (it cost ~100 VALU per k-tile next to 48 MFMAs).  The W-padded input keeps every w in range.
        g.kw = 1; g.Cs = d->pack_w * 4; g.lda = 4;
        pl->ut = 1;
        pl->sp_kind = 1;
      }
      if (g.s2) { pl->ut = 1; pl->sp_kind = 2; }
      VLFB_REQUIRE(!pl->sp_pl || ((pl->ident || pl->ut) && d->a_pstride % 8 == 0 && (g.lda % 8 == 0 || pl->packw)),
                   "conv: a pre-split activation operand needs plain rows or taps that span whole 32-element k-tiles");
      VLFB_REQUIRE(d->o_planes != 2 || (d->o_pstride % 4 == 0 && batch == 1), "conv: o_planes = 2 needs batch 1 and an aligned o_pstride");
      const size_t sbuf = (pl->sp_pl ? (size_t)pl->sp * 128 * 64 : (size_t)128 * 128) + (size_t)pl->sp * pl->bn * 64;
      pl->lds = 2 * sbuf;
      if (pl->lds < (size_t)128 * pl->bn * 4) pl->lds = (size_t)128 * pl->bn * 4;
    }
  } else {
    pl->lds = (size_t)2 * (pl->bm + pl->bn) * 128;
    if (pl->stem || pl->rows) { pl->lds = pl->stem_lds; pl->threads = 512; }
  }
  // bias gradient next to the weight gradient (vlfb_conv_run_wgrad_bias): inside the transposed-read kernel where that
  // is what runs; every other family gets a column-sum pass behind it
  pl->bias_fused = d->mode == VLFB_CONV_WGRAD && d->wgrad_bias && is16(d->dtype) && pl->tn_tr && !pl->sp && !pl->stem &&
                   !pl->rows && !pl->tn8 && batch == 1 && !d->accumulate;
  if (pl->bias_fused && pl->splits > 1) pl->ws_elems += (long long)pl->splits * d->Cn;
  // (other families: per-slab column sums behind the weight slabs, folded in order -- no atomics)
  if (d->mode == VLFB_CONV_WGRAD && d->wgrad_bias && !pl->bias_fused)
    pl->ws_elems += (long long)colsum_slabs(pl->sp_pl ? VLFB_BF16 : d->dtype, M, d->Cn) * d->Cn;
  return VLFB_OK;
}

// every instance declares the LDS it needs once (64 KiB for the 128x128 tile)
template <typename K>
void launch_k(K kernel, const Plan& pl, hipStream_t s) {
  static bool configured = false;  // per template instance
  if (!configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    configured = true;
  }
  hipLaunchKernelGGL(kernel, pl.grid, dim3(pl.threads), pl.lds, s, pl.gp);
}
// one tile shape, with or without the uniform-tap gather (UT only exists for gathered, unpacked operands)
template <typename T, typename OutT, int BM, int BN, bool IDENT, bool DGRAD, bool PACKW, int RB, bool PRE, int NW>
void launch_nt_shape(const Plan& pl, hipStream_t s) {
  if constexpr (!IDENT && !PACKW) {
    if constexpr (DGRAD && sizeof(T) == 2) {
      if (pl.w2i) {
        launch_k(gemm_nt_kernel<T, OutT, BM, BN, IDENT, DGRAD, PACKW, RB, PRE, NW, 2, true, false, true>, pl, s);
        return;
      }
    }
    if (pl.ut) {
      launch_k(gemm_nt_kernel<T, OutT, BM, BN, IDENT, DGRAD, PACKW, RB, PRE, NW, 2, true>, pl, s);
      return;
    }
  }
  launch_k(gemm_nt_kernel<T, OutT, BM, BN, IDENT, DGRAD, PACKW, RB, PRE, NW, 2, false>, pl, s);
}

template <typename T, typename OutT, bool IDENT, bool DGRAD, bool PACKW>
void launch_nt(const Plan& pl, hipStream_t s) {
  constexpr bool BF = sizeof(T) == 2;
  constexpr bool CAN_PRE = BF && sizeof(OutT) == 2 && !PACKW;
  constexpr int NW = BF ? 8 : 4;        // bf16: 8 waves (2 x 4) per workgroup; fp32 parity path: 4 (2 x 2)
  if (pl.bn == 64) {
    if (CAN_PRE && pl.pre) launch_nt_shape<T, OutT, 128, 64, IDENT, DGRAD, PACKW, 128, CAN_PRE, NW>(pl, s);
    else launch_nt_shape<T, OutT, 128, 64, IDENT, DGRAD, PACKW, 128, false, NW>(pl, s);
    return;
  }
  if (CAN_PRE && pl.pre) launch_nt_shape<T, OutT, 128, 128, IDENT, DGRAD, PACKW, 128, CAN_PRE, NW>(pl, s);
  else launch_nt_shape<T, OutT, 128, 128, IDENT, DGRAD, PACKW, 128, false, NW>(pl, s);
}
template <typename T, typename OutT, bool IDENT, bool PACKW>
void launch_tn(const Plan& pl, hipStream_t s) {
  if (pl.bm == 128 && pl.bn == 128) launch_k(gemm_tn_kernel<T, OutT, 128, 128, IDENT, PACKW>, pl, s);
  else if (pl.bm == 64 && pl.bn == 256) launch_k(gemm_tn_kernel<T, OutT, 64, 256, IDENT, PACKW>, pl, s);
  else if (pl.bm == 64 && pl.bn == 128) launch_k(gemm_tn_kernel<T, OutT, 64, 128, IDENT, PACKW>, pl, s);
  else if (pl.bm == 128 && pl.bn == 64) launch_k(gemm_tn_kernel<T, OutT, 128, 64, IDENT, PACKW>, pl, s);
  else launch_k(gemm_tn_kernel<T, OutT, 64, 64, IDENT, PACKW>, pl, s);
}

template <typename T, typename OutT, bool IDENT, bool PACKW, bool SP = false>
void launch_tn_tr(const Plan& pl, hipStream_t s) {
  if (pl.bm == 128 && pl.bn == 128) launch_k(gemm_tn_tr_kernel<T, OutT, 128, 128, IDENT, PACKW, 8, SP>, pl, s);   // 8 waves
  else if (pl.bm == 64 && pl.bn == 128) launch_k(gemm_tn_tr_kernel<T, OutT, 64, 128, IDENT, PACKW, 4, SP>, pl, s);
  else if (pl.bm == 128 && pl.bn == 64) launch_k(gemm_tn_tr_kernel<T, OutT, 128, 64, IDENT, PACKW, 4, SP>, pl, s);
  else launch_k(gemm_tn_tr_kernel<T, OutT, 64, 64, IDENT, PACKW, 4, SP>, pl, s);
}

template <typename T, typename OutT>
int dispatch(const vlfb_conv_desc* d, const Plan& pl, hipStream_t s) {
  if constexpr (sizeof(T) == 2) {
    if (d->mode == VLFB_CONV_WGRAD && pl.stem) {
      static bool configured = false;     // per element type (template instance)
      if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_wgrad_kernel<T>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        configured = true;
      }
      hipLaunchKernelGGL(stem_wgrad_kernel<T>, pl.grid, dim3(512), pl.lds, s, pl.gp);
      return check_launch("conv wgrad (stem) kernel");
    }
    if (d->mode == VLFB_CONV_WGRAD && pl.rows == 2) return launch_wgrad_rows_fat(pl.gp, pl.splits, d->dtype, s);
    if (d->mode == VLFB_CONV_WGRAD && pl.rows) return launch_wgrad_rows(pl.gp, pl.splits, pl.lds, d->dtype, s);
    if (d->mode == VLFB_CONV_WGRAD && pl.tn8) return launch_tn8(pl.gp, pl.grid, d->dtype, sizeof(OutT) == 4, s);
    if (d->mode == VLFB_CONV_WGRAD && pl.tn_tr) {
      if (pl.ident) launch_tn_tr<T, OutT, true, false>(pl, s);
      else if (pl.packw) launch_tn_tr<T, OutT, false, true>(pl, s);
      else launch_tn_tr<T, OutT, false, false>(pl, s);
      return check_launch("conv wgrad (tr) kernel");
    }
    if (d->mode != VLFB_CONV_WGRAD && pl.skinny) return launch_skinny_nt(pl.gp, d->dtype, sizeof(OutT) == 4, s);
    if (d->mode == VLFB_CONV_FPROP && pl.stemf && !pl.gp.R && !pl.gp.Mask) return launch_stem_fprop(pl.gp, d->dtype, s);
    if (d->mode != VLFB_CONV_WGRAD && pl.rows64) return launch_conv_rows64(pl.gp, d->mode, d->dtype, s);
    if (d->mode != VLFB_CONV_WGRAD && pl.nts) return launch_nts(pl.gp, pl.nts_mode, d->dtype, s);
    if (d->mode != VLFB_CONV_WGRAD && pl.nt8)
      return launch_nt8(pl.gp, pl.nt8_bm, pl.nt8, pl.nt8_mode, d->dtype, sizeof(OutT) == 4,
                        (unsigned)(d->batch > 0 ? d->batch : 1), s);
  }
  if (d->mode == VLFB_CONV_WGRAD) {
    if (pl.ident) launch_tn<T, OutT, true, false>(pl, s);
    else if (pl.packw) launch_tn<T, OutT, false, true>(pl, s);
    else launch_tn<T, OutT, false, false>(pl, s);
  } else if (d->mode == VLFB_CONV_DGRAD) {
    if (pl.ident) launch_nt<T, OutT, true, false, false>(pl, s);
    else launch_nt<T, OutT, false, true, false>(pl, s);
  } else {
    if (pl.ident) launch_nt<T, OutT, true, false, false>(pl, s);
    else if (pl.packw) launch_nt<T, OutT, false, false, true>(pl, s);
    else launch_nt<T, OutT, false, false, false>(pl, s);
  }
  return check_launch("conv kernel");
}

}  // namespace
}  // namespace vlfb

using namespace vlfb;

extern "C" void vlfb_conv_desc_init(vlfb_conv_desc* d) {
  ::memset(d, 0, sizeof(*d));
  d->dtype = VLFB_BF16; d->out_dtype = VLFB_BF16;
  d->N = d->Tr = d->Hr = d->Wr = 1;
  d->Ts = d->Hs = d->Ws = 1;
  d->kt = d->kh = d->kw = 1;
  d->st = d->sh = d->sw = 1;
  d->dt = d->dh = d->dw = 1;
  d->batch = 1;
  d->alpha = 1.0f;
}

// Plans are pure functions of the descriptor: cached per thread, keyed by the descriptor bytes (a training
// step replays the same ~280 descriptors; the planner's split search is not free).
static int cached_plan(const vlfb_conv_desc* d, Plan* out) {
  static thread_local std::unordered_map<std::string, Plan> cache;
  const std::string key(reinterpret_cast<const char*>(d), sizeof(*d));
  auto it = cache.find(key);
  if (it != cache.end()) { *out = it->second; return VLFB_OK; }
  const int rc = make_plan(d, out);
  if (rc == VLFB_OK) {
    if (cache.size() > 4096) cache.clear();
    cache.emplace(key, *out);
  }
  return rc;
}

extern "C" int64_t vlfb_conv_workspace_bytes(const vlfb_conv_desc* d) {
  Plan pl;
  if (cached_plan(d, &pl) != VLFB_OK) return -1;
  return pl.ws_elems * 4;
}

// Which kernel family, tile shape and split count the library runs for a descriptor (the planner is a pure function
// of the descriptor, so this IS what vlfb_conv_run launches; R / Mask only decide between the direct stem FPROP and the
// tiled kernel).  Test / bench support: "the plan under test is the plan under the stopwatch".
extern "C" int vlfb_conv_plan_describe(const vlfb_conv_desc* d, char* buf, int64_t buf_bytes) {
  VLFB_REQUIRE(d && buf && buf_bytes >= 96, "conv_plan_describe: buf of at least 96 bytes");
  Plan pl;
  const int rc = cached_plan(d, &pl);
  if (rc != VLFB_OK) return rc;
  const bool h16 = is16(d->dtype);
  const char* dt = d->dtype == VLFB_F32 ? (pl.sp ? (pl.sp == 3 ? "f32x6" : "f32x3") : "f32") : pl.h2 ? "f16x3" : d->dtype == VLFB_F16 ? "f16" : "bf16";
  if (d->mode == VLFB_CONV_WGRAD) {
    const char* fam = pl.sp ? (pl.sp_pl ? "tn_tr_planes" : "tn_split")
                      : (h16 && pl.stem) ? "stem_wgrad" : (h16 && pl.rows == 2) ? "wgrad_rows_fat" : (h16 && pl.rows) ? "wgrad_rows"
                      : (h16 && pl.tn8) ? "tn8" : (h16 && pl.tn_tr) ? "tn_tr" : "tn";
    snprintf(buf, (size_t)buf_bytes, "%s %s %dx%d splits=%d", fam, dt, pl.bm, pl.bn, pl.splits);
  } else {
    const char* fam;
    int bm = pl.bm, bn = pl.bn;
    if (pl.skinny_sp) { fam = "nt_skinny_split"; bm = 64; bn = 16; }
    else if (pl.sp) fam = pl.sp_pl ? "nt_planes" : "nt_split";
    else if (pl.h2 && pl.stemf) fam = "stem_fprop_pair";
    else if (pl.h2 && pl.nt8) { fam = "nt8_pair"; bm = pl.nt8_bm; bn = pl.nt8; }
    else if (pl.h2) fam = "nt_pair";
    else if (h16 && pl.skinny) fam = "nt_skinny";
    else if (h16 && d->mode == VLFB_CONV_FPROP && pl.stemf) fam = "stem_fprop";
    else if (h16 && pl.rows64) fam = "conv_rows64";
    else if (h16 && pl.nts) fam = "nt_stream";
    else if (h16 && pl.nt8) { fam = "nt8"; bm = pl.nt8_bm; bn = pl.nt8; }
    else fam = "nt";
    snprintf(buf, (size_t)buf_bytes, "%s %s %dx%d%s%s%s%s", fam, dt, bm, bn, pl.ut ? " ut" : "", pl.gp.s2 ? (d->algo == VLFB_ALGO_CLASS0 ? " class0" : " classes") : "",
             pl.w2i ? " w2" : "", pl.pre ? " pre" : "");
  }
  return VLFB_OK;
}

extern "C" int64_t vlfb_query_workspace(int op, const void* arg) {
  if (!arg) { set_error(VLFB_ERR_ARG, "query_workspace: arg is required"); return -1; }
  switch (op) {
    case VLFB_WS_CONV: return vlfb_conv_workspace_bytes(static_cast<const vlfb_conv_desc*>(arg));
    case VLFB_WS_MAXPOOL_ARGMAX: {
      const vlfb_pool_desc* d = static_cast<const vlfb_pool_desc*>(arg);
      const int es = vlfb_pool_argmax_bytes(d);
      if (es <= 0) return -1;
      return (int64_t)d->N * d->To * d->Ho * d->Wo * d->C * es;
    }
    case VLFB_WS_FBO_ATTN_BWD: {
      const int64_t* v = static_cast<const int64_t*>(arg);
      if (v[0] <= 0 || v[1] <= 0) { set_error(VLFB_ERR_ARG, "query_workspace: r, k must be positive"); return -1; }
      return v[0] * v[1] * 4;
    }
    case VLFB_WS_ATTN_SCORES: {
      const int64_t* v = static_cast<const int64_t*>(arg);
      if (v[0] <= 0 || v[1] <= 0 || v[2] <= 0) { set_error(VLFB_ERR_ARG, "query_workspace: b, l1, l2 must be positive"); return -1; }
      return v[0] * v[1] * v[2] * 4;
    }
    case VLFB_WS_BN: {
      const int64_t* v = static_cast<const int64_t*>(arg);
      const int64_t n = vlfb_bn_workspace_bytes((int)v[0], v[1], v[2]);
      if (n < 0) set_error(VLFB_ERR_ARG, "query_workspace: bad dtype / rows / C for SpatialBN");
      return n;
    }
    default: set_error(VLFB_ERR_ARG, "query_workspace: unknown op %d", op); return -1;
  }
}

extern "C" int vlfb_conv_run(const vlfb_conv_desc* d, const void* A, const void* B, const void* P,
                             void* O, const float* bias, const float* rowscale, const void* R,
                             const void* Mask, void* workspace, int64_t workspace_bytes,
                             vlfb_stream_t stream) {
  return vlfb_conv_run_planes(d, A, B, P, O, bias, rowscale, R, Mask, workspace, workspace_bytes, nullptr, stream);
}

static int conv_run_impl(const vlfb_conv_desc* d, const void* A, const void* B, const void* P,
                         void* O, const float* bias, const float* rowscale, const void* R,
                         const void* Mask, void* workspace, int64_t workspace_bytes, void* O_planes, float* dbias,
                         vlfb_stream_t stream, const void* R_lo = nullptr, void* O_lo = nullptr);

extern "C" int vlfb_conv_run_args(const vlfb_conv_desc* d, const vlfb_conv_args* a, vlfb_stream_t stream) {
  VLFB_REQUIRE(d && a, "conv_run_args: descriptor and arguments are required");
  VLFB_REQUIRE(!d->wgrad_bias == !a->dbias, "conv_run_args: dbias goes with desc.wgrad_bias");
  return conv_run_impl(d, a->A, a->B, a->P, a->O, a->bias, a->rowscale, a->R, a->Mask, a->workspace, a->workspace_bytes,
                       a->O_planes, a->dbias, stream, a->R_lo, a->O_lo);
}

extern "C" int vlfb_conv_run_planes(const vlfb_conv_desc* d, const void* A, const void* B, const void* P,
                                    void* O, const float* bias, const float* rowscale, const void* R,
                                    const void* Mask, void* workspace, int64_t workspace_bytes, void* O_planes,
                                    vlfb_stream_t stream) {
  VLFB_REQUIRE(!d->wgrad_bias, "conv: a descriptor with wgrad_bias runs through vlfb_conv_run_wgrad_bias");
  return conv_run_impl(d, A, B, P, O, bias, rowscale, R, Mask, workspace, workspace_bytes, O_planes, nullptr, stream);
}

extern "C" int vlfb_conv_run_wgrad_bias(const vlfb_conv_desc* d, const void* A, const void* P, void* O, float* dbias,
                                        const float* rowscale, void* workspace, int64_t workspace_bytes,
                                        vlfb_stream_t stream) {
  VLFB_REQUIRE(d->mode == VLFB_CONV_WGRAD && d->wgrad_bias && dbias, "conv_run_wgrad_bias: a WGRAD descriptor with wgrad_bias = 1 and dbias");
  return conv_run_impl(d, A, nullptr, P, O, nullptr, rowscale, nullptr, nullptr, workspace, workspace_bytes, nullptr, dbias, stream);
}

static int conv_run_impl(const vlfb_conv_desc* d, const void* A, const void* B, const void* P,
                         void* O, const float* bias, const float* rowscale, const void* R,
                         const void* Mask, void* workspace, int64_t workspace_bytes, void* O_planes, float* dbias,
                         vlfb_stream_t stream, const void* R_lo, void* O_lo) {
  Plan pl;
  int rc = cached_plan(d, &pl);
  if (rc != VLFB_OK) return rc;
  const bool sp_pair = pl.sp && d->out_dtype == VLFB_F16;     // split launch writing (and adding) two fp16 planes
  VLFB_REQUIRE(!sp_pair || (O_lo && (!R || R_lo)), "conv: a two-plane fp16 output needs O_lo (and R_lo with R)");
  VLFB_REQUIRE(!pl.h2 || d->out_dtype == VLFB_F32 || O_lo, "conv: F16X3 math with a 16-bit output writes two planes (O, O_lo)");
  VLFB_REQUIRE(!pl.h2 || !Mask, "conv: F16X3 launches take no mask");
  // (a 16-bit FPROP / DGRAD with an fp32 output: O_lo = the same values rounded to the operand type, for the 16-bit launches
  // that read this tensor next)
  const bool copy16 = O_lo && !R_lo && !pl.sp && !pl.h2 && is16(d->dtype) && d->out_dtype == VLFB_F32 && d->mode != VLFB_CONV_WGRAD;
  if ((R_lo || O_lo) && !sp_pair && !(pl.h2 && d->out_dtype == VLFB_F32 && !R_lo)) {
    // two-term residual / output: the epilogues of the tiled NT families (128 x 128, 256-row pipelined) carry it
    VLFB_REQUIRE(d->mode != VLFB_CONV_WGRAD && is16(d->dtype) && (d->out_dtype == d->dtype || copy16) && !pl.sp,
                 "conv: R_lo / O_lo belong to 16-bit FPROP / DGRAD launches with 16-bit outputs (O_lo alone: also with an fp32 output)");
    VLFB_REQUIRE(!R_lo || R, "conv: R_lo without R");
    if (pl.skinny || pl.rows64 || pl.nts || (d->mode == VLFB_CONV_FPROP && pl.stemf && !pl.h2)) {      // (the two-plane stem writes O / O_lo itself)
      vlfb_conv_desc d2 = *d;
      d2.algo = VLFB_ALGO_TILE128;
      rc = cached_plan(&d2, &pl);
      if (rc != VLFB_OK) return rc;
    }
    VLFB_REQUIRE(pl.gp.vec_epi, "conv: R_lo / O_lo need 16-byte aligned output rows");
  }
  VLFB_REQUIRE(A && O, "conv: A and O are required");
  if (d->mode == VLFB_CONV_WGRAD) VLFB_REQUIRE(P != nullptr, "conv: WGRAD needs P");
  else VLFB_REQUIRE(B != nullptr, "conv: FPROP/DGRAD need B");
  VLFB_REQUIRE(d->bias_mode == VLFB_BIAS_NONE || bias != nullptr, "conv: bias_mode set but bias is NULL");
  if (pl.ws_elems > 0 && (workspace == nullptr || workspace_bytes < pl.ws_elems * 4))
    return set_error(VLFB_ERR_WORKSPACE, "conv: split WGRAD needs %lld workspace bytes, got %lld",
                     (long long)pl.ws_elems * 4, (long long)workspace_bytes);
  GP& g = pl.gp;
  g.A = (const char*)A; g.B = (const char*)B; g.P = (const char*)P; g.O = (char*)O;
  g.bias = bias; g.rowscale = rowscale; g.R = (const char*)R; g.Mask = (const char*)Mask;
  g.ws = (float*)workspace;
  VLFB_REQUIRE((d->o_planes > 0) == (O_planes != nullptr), "conv: O_planes goes with desc.o_planes");
  g.OP = (char*)O_planes;
  g.dbias = pl.bias_fused ? dbias : nullptr;
  g.R2 = (const char*)R_lo; g.O2 = (char*)O_lo;
  g.pair_io = sp_pair ? 1 : 0;
  g.pair_il = (pl.h2 && d->a_pstride == 32) ? 1 : 0;
  // Non-temporal epilogue rows (GP::nt_epi) where a launch moves at least 4 bytes per output element through its epilogue
  // -- fp32 or two-plane outputs / residuals, two-term gradients: the "mix" and "split" launches, +1.0-1.3 % on their steps --
  // and not on plain 16-bit launches, whose steps measured -0.25 % with it (their rows are half as large, the next launch
  // finds more of them still cached).  VLFB_NT_EPI: 0 never, 1 this rule (default), 2 every launch.
  static const int nt_epi = getenv("VLFB_NT_EPI") ? atoi(getenv("VLFB_NT_EPI")) : 1;
  const bool wide_epi = pl.h2 || pl.sp || pl.w2i || d->dt == 0 || d->out_dtype == VLFB_F32 || R_lo || O_lo;
  g.nt_epi = nt_epi >= 2 || (nt_epi == 1 && wide_epi);
  hipStream_t s = (hipStream_t)stream;
  if (!R && !Mask) pl.pre = 0;
  if (pl.skinny_sp) rc = launch_skinny_nt_split(g, s);
  else if (pl.sp) {
    if (d->mode == VLFB_CONV_WGRAD && pl.sp_pl) {
      if (pl.ident) launch_tn_tr<bf16_t, float, true, false, true>(pl, s);
      else if (pl.packw) launch_tn_tr<bf16_t, float, false, true, true>(pl, s);
      else launch_tn_tr<bf16_t, float, false, false, true>(pl, s);
      rc = check_launch("conv wgrad (planes) kernel");
    } else if (d->mode == VLFB_CONV_WGRAD) rc = launch_tn_split(g, pl.bm, pl.bn, pl.ident, pl.packw, pl.grid, pl.lds, s);
    else if (pl.sp_pl) rc = launch_nt_planes(g, pl.sp, pl.bn, pl.sp_kind, pl.grid, pl.lds, s);
    else rc = launch_nt_split(g, pl.sp, pl.bn, pl.sp_kind, pl.ut != 0, pl.grid, pl.lds, s);
  } else if (pl.h2 && pl.stemf && !R && !Mask) rc = launch_stem_fprop_pair(g, s);
  else if (pl.h2 && pl.nt8) rc = launch_nt8_pair(g, pl.nt8_bm, pl.nt8, pl.nt8_mode, d->out_dtype == VLFB_F32, s);
  else if (pl.h2) rc = launch_nt_pair(g, pl.bn, pl.ident, pl.pre != 0, d->out_dtype == VLFB_F32, pl.grid, pl.lds, s);
  else if (d->dtype == VLFB_F32) rc = dispatch<float, float>(d, pl, s);
  else if (d->dtype == VLFB_F16) rc = d->out_dtype == VLFB_F32 ? dispatch<f16_t, float>(d, pl, s) : dispatch<f16_t, f16_t>(d, pl, s);
  else rc = d->out_dtype == VLFB_F32 ? dispatch<bf16_t, float>(d, pl, s) : dispatch<bf16_t, bf16_t>(d, pl, s);
  if (rc != VLFB_OK) return rc;
  if (dbias && !pl.bias_fused) {
    // families without the fused column sums: one pass over P behind the launch -- per-slab partial rows behind the
    // weight slabs, folded in slab order (then alpha * rowscale, as the weights): deterministic
    VLFB_REQUIRE(!pl.sp_pl, "conv: wgrad_bias with pre-split operands is not supported (pass the fp32 operands)");
    const int slabs = colsum_slabs(d->dtype, g.M, d->Cn);
    float* part = g.ws + (pl.splits > 1 ? (long long)pl.splits * ((long long)d->Cn * g.ldo) : 0);
    rc = colsum_partials(P, d->dtype, g.M, d->Cn, g.ldp, part, s);
    if (rc != VLFB_OK) return rc;
    hipLaunchKernelGGL(wgrad_bias_reduce_kernel, dim3((unsigned)((d->Cn + 63) / 64)), dim3(64), 0, s, part, dbias, rowscale,
                       d->Cn, slabs, d->alpha);
  }
  // (a launch with fused column sums and splits: its bias partial rows are folded by the weight reduce below)
  const float* bias_rows = (dbias && pl.bias_fused && pl.splits > 1) ? g.ws + (long long)pl.splits * ((long long)d->Cn * g.ldo) : nullptr;
  if (pl.splits > 1) {
    const long long n = (long long)d->Cn * g.K;
    VLFB_REQUIRE(n % 4 == 0, "conv: split WGRAD output size must be a multiple of 4");
    const int G = pl.splits >= 32 ? 16 : pl.splits >= 8 ? 4 : 1;
#define VLFB_REDUCE(OT, GG, CC)                                                                                         \
  do {                                                                                                                  \
    const unsigned wblocks = (unsigned)((n / 4 + CC - 1) / CC);                                                         \
    const unsigned bblocks = bias_rows ? (unsigned)((d->Cn + CC * GG - 1) / (CC * GG)) : 0u;                            \
    hipLaunchKernelGGL((wgrad_reduce_kernel<OT, GG, CC>), dim3(wblocks + bblocks), dim3(CC * GG), 0, s, g.ws, g.O,     \
                       rowscale, n, g.ldo, pl.splits, d->alpha, d->accumulate, bias_rows, bias_rows ? dbias : nullptr,  \
                       (int)d->Cn, (int)wblocks);                                                                       \
  } while (0)
    if (d->out_dtype == VLFB_F32) {
      if (G == 16) VLFB_REDUCE(float, 16, 16); else if (G == 4) VLFB_REDUCE(float, 4, 64); else VLFB_REDUCE(float, 1, 64);
    } else if (d->out_dtype == VLFB_F16) {
      if (G == 16) VLFB_REDUCE(f16_t, 16, 16); else if (G == 4) VLFB_REDUCE(f16_t, 4, 64); else VLFB_REDUCE(f16_t, 1, 64);
    } else {
      if (G == 16) VLFB_REDUCE(bf16_t, 16, 16); else if (G == 4) VLFB_REDUCE(bf16_t, 4, 64); else VLFB_REDUCE(bf16_t, 1, 64);
    }
#undef VLFB_REDUCE
    return check_launch("wgrad reduce");
  }
  return VLFB_OK;
}
