// Non-local attention scores fused with their row softmax (gfx950).
//
// The spacetime non-local block (lib/models/nonlocal_helper.py:94-121) computes  P = softmax(scale * theta^T phi)
// over L2 = 784 (32 x 224^2 clips) / 1024 (test crop 256) keys and, backwards,
// dS = scale * P o (dP - rowsum(dP o P))  with  dP = dY g^T.  As three launches (batched NT GEMM with an fp32
// output, row softmax, and their backward twins) the fp32 score matrix makes two round trips through HBM:
// 315 MB written + 315 MB read per res3 block and direction at 8 clips.  Here a workgroup owns 64 query rows
// and ALL keys, so the scores never leave the registers:
//   * 8 waves, wave w owns the keys [w * 16 FN, (w + 1) * 16 FN) of all 64 queries: 4 x FN accumulator
//     fragments (FN = 7: up to 896 keys, FN = 8: up to 1024);
//   * the query tile (64 x Ci, the only operand the waves share) is DMA'd once into XOR-swizzled LDS; the key
//     operand is private to a wave, so its MFMA fragments come straight from global memory / L2 (16 bytes
//     per lane, next k-step prefetched into a second register set) -- an LDS round trip would buy no reuse;
//   * MFMA with the keys as the A operand: a lane ends up with 4 consecutive keys of one query, so the row
//     statistics (max / sum of exp, or rowsum(dP o P)) are a register tree + two cross-lane steps + one
//     8-entry exchange through LDS;
//   * the probability tile is staged in LDS as 16-bit rows and leaves (forward) or enters and leaves
//     (backward) with 16-byte accesses on whole rows.
// Arithmetic: fp32 scores / statistics, the same formulas as vlfb_softmax_fwd / vlfb_softmax_bwd.
#include "vlfb_gemm_common.h"

namespace vlfb {
namespace {

struct AttnP {
  const char* A;        // queries  [batch][L1][Ci]   (theta, or dY)
  const char* B;        // keys     [batch][L2][Ci]   (phi, or g)
  const char* Pin;      // backward: probabilities [batch][L1][L2]
  char* Out;            // forward: P, backward: dS   [batch][L1][L2]
  int L1, L2, Ci;
  float scale;
};

template <typename T, int FN, bool BWD>
__global__ __launch_bounds__(512) void attn_rows_kernel(const AttnP p) {
  typedef typename V16<T>::V vec_t;
  constexpr int L2P = 8 * 16 * FN;                 // keys covered by the 8 waves
  constexpr int ROWB = L2P * 2 + 16;               // LDS row of the 16-bit probability tile (+16: bank spread)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int q0 = blockIdx.x * 64;
  const long long z = blockIdx.y;
  const char* Ab = p.A + z * (long long)p.L1 * p.Ci * 2;
  const char* Bb = p.B + z * (long long)p.L2 * p.Ci * 2;
  float* red = reinterpret_cast<float*>(smem + 64 * ROWB);       // [8 waves][64 queries] exchange

  // ---- query tile -> LDS (64 rows x Ci, rows of 128 B k-chunks: tile[kc][row][128 B], swizzled) -------
  const int kchunks = p.Ci >> 6;                   // 64-element (128-byte) k chunks
  {
    const auto rsA = make_rsrc(Ab, (unsigned)p.L1 * (unsigned)p.Ci * 2u);
    const int r = tid >> 3, slot = tid & 7;        // 64 rows x 8 slots of 16 B = one k chunk per pass
    const int cg = slot ^ (r & 7);
    const unsigned off = (q0 + r) < p.L1 ? (unsigned)((q0 + r) * p.Ci + cg * 8) * 2u : kOOB;
    for (int kc = 0; kc < kchunks; ++kc) bufglds16(rsA, off, (unsigned)kc * 128u, smem + kc * 8192 + wave * 1024);
  }
  // ---- key fragments straight from global memory ---------------------------------------------------------
  const char* brow[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    int key = wave * (16 * FN) + j * 16 + l15;
    if (key > p.L2 - 1) key = p.L2 - 1;            // padding keys read a real row; masked below
    brow[j] = Bb + ((long long)key * p.Ci + g * 8) * 2;
  }
  f32x4_v acc[FN][4];
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_v{0.f, 0.f, 0.f, 0.f};
  vec_t bf0[FN], bf1[FN];                          // two statically indexed sets (a runtime-indexed one goes to scratch)
#pragma unroll
  for (int j = 0; j < FN; ++j) bf0[j] = *reinterpret_cast<const vec_t*>(brow[j]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the query tile (and the first key fragments) landed
  __syncthreads();
  const int ksteps = p.Ci >> 5;                    // 32 k per MFMA; even (Ci % 64 == 0)
  const int key7 = l15 & 7;
  const int arow = l15 * 128;
  const int ko0 = ((0 + g) ^ key7) << 4, ko1 = ((4 + g) ^ key7) << 4;
  auto step = [&](const vec_t (&bf)[FN], const char* at, int kofs) {
    vec_t af[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const vec_t*>(at + i * 2048 + arow + kofs);
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] = V16<T>::mma(bf[j], af[i], acc[j][i]);
  };
  for (int ks = 0; ks < ksteps; ks += 2) {
    const char* at = smem + (ks >> 1) * 8192;      // one 128-byte k chunk = two k-steps
#pragma unroll
    for (int j = 0; j < FN; ++j) bf1[j] = *reinterpret_cast<const vec_t*>(brow[j] + (ks + 1) * 64);
    step(bf0, at, ko0);
    if (ks + 2 < ksteps) {
#pragma unroll
      for (int j = 0; j < FN; ++j) bf0[j] = *reinterpret_cast<const vec_t*>(brow[j] + (ks + 2) * 64);
    }
    step(bf1, at, ko1);
  }
  __syncthreads();                                 // every wave is done with the query tile: LDS is reused below

  // lane holds, for query q = i*16 + l15, the keys kb(j) + 0..3 with kb(j) = wave*16*FN + j*16 + g*4
  const int kbase = wave * (16 * FN) + g * 4;
  auto row_reduce = [&](float (&v)[4], bool is_max) {   // over the g lanes of a query, then over the 8 waves
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float o = __shfl_xor(v[i], 16);
      v[i] = is_max ? fmaxf(v[i], o) : v[i] + o;
      o = __shfl_xor(v[i], 32);
      v[i] = is_max ? fmaxf(v[i], o) : v[i] + o;
    }
    if (g == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) red[wave * 64 + i * 16 + l15] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a = red[i * 16 + l15];
#pragma unroll
      for (int w = 1; w < 8; ++w) a = is_max ? fmaxf(a, red[w * 64 + i * 16 + l15]) : a + red[w * 64 + i * 16 + l15];
      v[i] = a;
    }
    __syncthreads();
  };
  char* Ob = p.Out + z * (long long)p.L1 * p.L2 * 2;
  const int row_chunks = (p.L2 * 2) >> 4;          // 16-byte chunks of an output row (L2 % 8 == 0)

  if (!BWD) {
    float m[4], sum[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) m[i] = -INFINITY;
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float s = (kbase + j * 16 + r) < p.L2 ? acc[j][i][r] * p.scale : -INFINITY;
          acc[j][i][r] = s;
          m[i] = fmaxf(m[i], s);
        }
    row_reduce(m, true);
#pragma unroll
    for (int i = 0; i < 4; ++i) sum[i] = 0.f;
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __expf(acc[j][i][r] - m[i]);
          acc[j][i][r] = e;
          sum[i] += e;
        }
    row_reduce(sum, false);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float inv = 1.0f / sum[i];
      char* row = smem + (i * 16 + l15) * ROWB;
#pragma unroll
      for (int j = 0; j < FN; ++j)
        *reinterpret_cast<uint2*>(row + (kbase + j * 16) * 2) =
            make_uint2(Elem<T>::pack2(acc[j][i][0] * inv, acc[j][i][1] * inv), Elem<T>::pack2(acc[j][i][2] * inv, acc[j][i][3] * inv));
    }
    __syncthreads();
  } else {
    // probabilities of this tile: HBM -> LDS rows (16-byte pieces, whole rows)
    const char* Pb = p.Pin + z * (long long)p.L1 * p.L2 * 2;
    for (int idx = tid; idx < 64 * row_chunks; idx += 512) {
      const int r = idx / row_chunks, c = idx - r * row_chunks;
      if (q0 + r < p.L1)
        *reinterpret_cast<uint4*>(smem + r * ROWB + c * 16) =
            *reinterpret_cast<const uint4*>(Pb + ((long long)(q0 + r) * p.L2) * 2 + c * 16);
    }
    __syncthreads();
    float dot[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dot[i] = 0.f;
    // two passes over the staged probabilities (LDS reads are cheap; keeping them in registers next to the 4 x FN
    // accumulator fragments would not fit)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if ((kbase + j * 16) < p.L2) {                           // L2 % 4 == 0: the four keys live or die together
          const uint2 w = *reinterpret_cast<const uint2*>(smem + (i * 16 + l15) * ROWB + (kbase + j * 16) * 2);
          dot[i] += (acc[j][i][0] * Elem<T>::lo(w.x) + acc[j][i][1] * Elem<T>::hi(w.x)) +
                    (acc[j][i][2] * Elem<T>::lo(w.y) + acc[j][i][3] * Elem<T>::hi(w.y));
        }
      }
    row_reduce(dot, false);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      char* row = smem + (i * 16 + l15) * ROWB;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        if ((kbase + j * 16) < p.L2) {
          uint2* cell = reinterpret_cast<uint2*>(row + (kbase + j * 16) * 2);
          const uint2 w = *cell;                                 // this lane is the only reader / writer of the cell
          *cell = make_uint2(Elem<T>::pack2(p.scale * Elem<T>::lo(w.x) * (acc[j][i][0] - dot[i]),
                                            p.scale * Elem<T>::hi(w.x) * (acc[j][i][1] - dot[i])),
                             Elem<T>::pack2(p.scale * Elem<T>::lo(w.y) * (acc[j][i][2] - dot[i]),
                                            p.scale * Elem<T>::hi(w.y) * (acc[j][i][3] - dot[i])));
        }
      }
    }
    __syncthreads();
  }
  // ---- staged rows -> HBM, 16 bytes per lane on whole rows ---------------------------------------------
  for (int idx = tid; idx < 64 * row_chunks; idx += 512) {
    const int r = idx / row_chunks, c = idx - r * row_chunks;
    if (q0 + r < p.L1)
      *reinterpret_cast<uint4*>(Ob + ((long long)(q0 + r) * p.L2) * 2 + c * 16) = *reinterpret_cast<const uint4*>(smem + r * ROWB + c * 16);
  }
}

// -------------------------------------------------------------------------------------------------------------
// Second tiling, for the training shapes (up to 784 keys, Ci <= 256): the WAVES split the QUERIES (16 each, 128 per
// workgroup) and every wave covers ALL keys, so that the key operand -- the big one -- is shared by the 8 waves and
// goes through LDS in whole lines (a k-chunk of 32: [keys][64 B], three buffers, DMA two chunks ahead with a
// counted vmcnt), while the 16 x Ci query fragments of a wave sit in registers for the whole k-loop.  The kernel
// above reads its key fragments fragment-shaped from L2 (16 rows x 64 B per instruction), which costs it more than
// it saves at 64 queries per workgroup: 188 us vs 221 us composed on the res3 shape, and a loss at Ci = 512.
// A lane ends up with one query (l15) and 4 consecutive keys per fragment: row statistics never leave the wave.
// -------------------------------------------------------------------------------------------------------------
template <typename T, bool BWD>
__global__ __launch_bounds__(512) void attn_qrows_kernel(const AttnP p) {
  typedef typename V16<T>::V vec_t;
  constexpr int NJ = 49;                           // key fragments of a wave (49 x 16 = 784 keys: L2 == 784)
  constexpr int QF = 8;                            // k-steps of 32 (Ci == 256)
  constexpr int CHUNK = NJ * 16 * 64;              // bytes of one key k-chunk buffer
  constexpr int HALF = 400;                        // keys per staging pass
  constexpr int SROW = HALF * 2 + 16;              // staging row (one query, half the keys)
  constexpr int WREG = 3 * CHUNK / 8;              // staging bytes per wave (>= 16 * SROW)
  static_assert(16 * SROW <= WREG, "staging region");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int q0 = blockIdx.x * 128 + wave * 16;
  const long long z = blockIdx.y;
  const char* Ab = p.A + z * (long long)p.L1 * p.Ci * 2;
  const char* Bb = p.B + z * (long long)p.L2 * p.Ci * 2;

  // ---- the wave's query fragments (B operand of the MFMA): 16 queries x 32 k per step, one step ahead, straight
  // from global memory into registers.  The load of step ks + 1 is issued BEFORE the DMAs of chunk ks + 2, so the
  // counted wait "all but the last 7" at the top of step ks + 1 covers it.
  int qrow = q0 + l15;
  if (qrow > p.L1 - 1) qrow = p.L1 - 1;            // rows past the end compute garbage that is never stored
  const char* qsrc = Ab + ((long long)qrow * p.Ci + g * 8) * 2;
  vec_t qn = *reinterpret_cast<const vec_t*>(qsrc);
  __builtin_amdgcn_sched_barrier(0);

  // ---- key chunk DMA: piece id = pass * 512 + tid -> (key row id >> 2, 16-byte slot id & 3); 3136 pieces -----
  // pass i reads key row i * 128 + (tid >> 2); i * 128 / 4 is a multiple of 4, so the swizzle key (row >> 2) & 3 is
  // the lane's own.  Passes 0..5 are whole; pass 6 has 64 pieces (wave 0): the other waves issue a zero-filling
  // dummy into a 1 KiB dump, and chunks past the last are zero-fills too, so that EVERY wave issues exactly 7 DMAs
  // per step and one immediate (vmcnt(7)) serves the whole loop.
  const auto rsB = make_rsrc(Bb, (unsigned)p.L2 * (unsigned)p.Ci * 2u);
  const unsigned koff0 = (unsigned)((tid >> 2) * p.Ci + (((tid & 3) ^ ((tid >> 4) & 3)) * 8)) * 2u;
  const unsigned kstride = (unsigned)p.Ci * 256u;
  auto dma_chunk = [&](int ks, int buf) {
    const bool real = ks < QF;
    char* dst = smem + buf * CHUNK + wave * 1024;
#pragma unroll
    for (int i = 0; i < 6; ++i)
      bufglds16(rsB, real ? koff0 + (unsigned)i * kstride : kOOB, (unsigned)ks * 64u, dst + i * 8192);
    bufglds16(rsB, real && wave == 0 ? koff0 + 6u * kstride : kOOB, (unsigned)ks * 64u,
              wave == 0 ? smem + buf * CHUNK + 6 * 8192 : smem + 3 * CHUNK + (wave - 1) * 1024);
  };

  f32x4_v acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) acc[j] = f32x4_v{0.f, 0.f, 0.f, 0.f};
  dma_chunk(0, 0);
  dma_chunk(1, 1);
  const int fofs = l15 * 64 + ((g ^ (l15 >> 2)) << 4);   // lane's 16 bytes inside a key fragment (64-byte rows)
  int cur = 0;                                     // ring slot of chunk ks
  // (a rolled loop on purpose: the accumulators are loop-carried, so each keeps ONE register tuple; unrolled over
  //  ks the allocator rotates them through fresh tuples and spills)
#pragma unroll 1
  for (int ks = 0; ks < QF; ++ks) {
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");   // chunk ks and the query fragment of step ks have landed
    asm volatile("s_barrier" ::: "memory");         // ... for every wave; slot of chunk ks + 2 was last read at step ks - 1
    const vec_t qc = qn;
    qn = *reinterpret_cast<const vec_t*>(qsrc + (ks + 1 < QF ? ks + 1 : ks) * 64);
    __builtin_amdgcn_sched_barrier(0);
    const int nxt = cur == 0 ? 2 : cur - 1;         // (cur + 2) % 3
    dma_chunk(ks + 2, nxt);
    unsigned kofs = (unsigned)(cur * CHUNK) + (unsigned)fofs;
    asm volatile("" : "+v"(kofs));                   // one base register per chunk: fragment j is an immediate offset
    const char* kb = smem + kofs;
    // two fragment registers in rotation (the accumulators leave room for no more)
    vec_t f0 = *reinterpret_cast<const vec_t*>(kb), f1 = *reinterpret_cast<const vec_t*>(kb + 1024);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (j & 1) {
        acc[j] = V16<T>::mma(f1, qc, acc[j]);
        if (j + 2 < NJ) f1 = *reinterpret_cast<const vec_t*>(kb + (j + 2) * 1024);
      } else {
        acc[j] = V16<T>::mma(f0, qc, acc[j]);
        if (j + 2 < NJ) f0 = *reinterpret_cast<const vec_t*>(kb + (j + 2) * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    cur = cur == 2 ? 0 : cur + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the trailing zero-fills)
  __syncthreads();                                 // the chunk buffers become per-wave staging

  // lane: query l15 of the wave, keys 16 j + 4 g + (0..3) in acc[j]
  char* st = smem + wave * WREG;
  const long long orow0 = ((long long)z * p.L1 + q0) * p.L2 * 2;   // byte offset of the wave's first output row
  auto wave_red = [&](float v, bool is_max) {
    float o = __shfl_xor(v, 16); v = is_max ? fmaxf(v, o) : v + o;
    o = __shfl_xor(v, 32); return is_max ? fmaxf(v, o) : v + o;
  };
  auto rows_out = [&](int h) {                     // staged half h -> Out, 16 bytes per lane on whole row pieces
    const int k0 = h * HALF, kn = min(p.L2 - k0, HALF);
    const int nch = (kn * 2) >> 4;
    for (int idx = lane; idx < 16 * nch; idx += 64) {
      const int r = idx / nch, c = idx - r * nch;
      if (q0 + r < p.L1)
        *reinterpret_cast<uint4*>(p.Out + orow0 + ((long long)r * p.L2 + k0) * 2 + c * 16) = *reinterpret_cast<const uint4*>(st + r * SROW + c * 16);
    }
  };
  auto rows_in = [&](int h) {                      // P half h -> staging
    const int k0 = h * HALF, kn = min(p.L2 - k0, HALF);
    const int nch = (kn * 2) >> 4;
    for (int idx = lane; idx < 16 * nch; idx += 64) {
      const int r = idx / nch, c = idx - r * nch;
      if (q0 + r < p.L1)
        *reinterpret_cast<uint4*>(st + r * SROW + c * 16) = *reinterpret_cast<const uint4*>(p.Pin + orow0 + ((long long)r * p.L2 + k0) * 2 + c * 16);
    }
  };
  constexpr int nhalves = 2;                       // (compile-time halves: nothing of half 1 is computed ahead of half 0's stores)
  if (!BWD) {
    // The accumulators stay read-only (an in-place update of single elements breaks up the MFMA register tuples
    // and costs spills): max over the raw logits (scale > 0), then exp2((s - max) * scale * log2 e) twice -- once
    // for the row sum, once for the stored probabilities.
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      m = fmaxf(fmaxf(m, fmaxf(acc[j][0], acc[j][1])), fmaxf(acc[j][2], acc[j][3]));
      if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
    m = wave_red(m, true);
    const float c = p.scale * 1.44269504088896341f, mc = -m * c;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      f32x4_v e;
      e[0] = __builtin_amdgcn_exp2f(fmaf(acc[j][0], c, mc)); e[1] = __builtin_amdgcn_exp2f(fmaf(acc[j][1], c, mc));
      e[2] = __builtin_amdgcn_exp2f(fmaf(acc[j][2], c, mc)); e[3] = __builtin_amdgcn_exp2f(fmaf(acc[j][3], c, mc));
      acc[j] = e;                                   // (whole-tuple replacement)
      sum += (e[0] + e[1]) + (e[2] + e[3]);
      if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    sum = wave_red(sum, false);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int h = 0; h < nhalves; ++h) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int kl = 16 * j + 4 * g - h * HALF;   // HALF % 16 == 0: a fragment lies in one half
        if (j / (HALF / 16) == h) {
          *reinterpret_cast<uint2*>(st + l15 * SROW + kl * 2) = make_uint2(
              Elem<T>::pack2(acc[j][0] * inv, acc[j][1] * inv), Elem<T>::pack2(acc[j][2] * inv, acc[j][3] * inv));
          __builtin_amdgcn_sched_barrier(0);        // (keeps the conversions from piling up registers)
        }
      }
      rows_out(h);
    }
  } else {
    float dot = 0.f;
#pragma unroll
    for (int h = 0; h < nhalves; ++h) {
      rows_in(h);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int kl = 16 * j + 4 * g - h * HALF;
        if (j / (HALF / 16) == h) {
          const uint2 w = *reinterpret_cast<const uint2*>(st + l15 * SROW + kl * 2);
          dot += (acc[j][0] * Elem<T>::lo(w.x) + acc[j][1] * Elem<T>::hi(w.x)) + (acc[j][2] * Elem<T>::lo(w.y) + acc[j][3] * Elem<T>::hi(w.y));
        }
        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
    dot = wave_red(dot, false);                     // (rows past L1 hold stale staging data; they are never stored)
#pragma unroll
    for (int h = 0; h < nhalves; ++h) {
      rows_in(h);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int kl = 16 * j + 4 * g - h * HALF;
        if (j / (HALF / 16) == h) {
          uint2* cell = reinterpret_cast<uint2*>(st + l15 * SROW + kl * 2);
          const uint2 w = *cell;
          *cell = make_uint2(Elem<T>::pack2(p.scale * Elem<T>::lo(w.x) * (acc[j][0] - dot), p.scale * Elem<T>::hi(w.x) * (acc[j][1] - dot)),
                             Elem<T>::pack2(p.scale * Elem<T>::lo(w.y) * (acc[j][2] - dot), p.scale * Elem<T>::hi(w.y) * (acc[j][3] - dot)));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      rows_out(h);
    }
  }
}

template <typename T, bool BWD>
void launch_attn_q(const AttnP& p, long long batch, hipStream_t s) {
  static bool configured = false;       // per template instance
  if (!configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_qrows_kernel<T, BWD>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    configured = true;
  }
  hipLaunchKernelGGL((attn_qrows_kernel<T, BWD>), dim3((unsigned)((p.L1 + 127) / 128), (unsigned)batch), dim3(512),
                     (size_t)3 * 49 * 16 * 64 + 7 * 1024, s, p);
}

template <typename T, int FN, bool BWD>
void launch_attn(const AttnP& p, long long batch, hipStream_t s) {
  constexpr size_t tile = 64 * (size_t)(8 * 16 * FN * 2 + 16);
  const size_t q = (size_t)64 * p.Ci * 2;
  const size_t lds = (tile > q ? tile : q) + 8 * 64 * 4;
  static bool configured = false;       // per template instance
  if (!configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_rows_kernel<T, FN, BWD>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    configured = true;
  }
  hipLaunchKernelGGL((attn_rows_kernel<T, FN, BWD>), dim3((unsigned)((p.L1 + 63) / 64), (unsigned)batch), dim3(512), lds, s, p);
}

template <bool BWD>
int run_attn(const AttnP& p, int dtype, long long batch, hipStream_t s) {
  if (p.L2 == 784 && p.Ci == 256) {     // queries-per-wave tiling (res3 non-local blocks of a 32 x 224^2 clip)
    if (dtype == VLFB_F16) launch_attn_q<f16_t, BWD>(p, batch, s); else launch_attn_q<bf16_t, BWD>(p, batch, s);
    return check_launch(BWD ? "attn_scores_bwd" : "attn_scores_fwd");
  }
  const bool wide = p.L2 > 896;
  if (dtype == VLFB_F16) {
    if (wide) launch_attn<f16_t, 8, BWD>(p, batch, s); else launch_attn<f16_t, 7, BWD>(p, batch, s);
  } else {
    if (wide) launch_attn<bf16_t, 8, BWD>(p, batch, s); else launch_attn<bf16_t, 7, BWD>(p, batch, s);
  }
  return check_launch(BWD ? "attn_scores_bwd" : "attn_scores_fwd");
}

bool supported(int dtype, long long l1, long long l2, long long ci) {
  // the red[] exchange sits behind the probability tile; the query tile (64 x Ci) must fit in front of it too
  return is16(dtype) && l2 >= 512 && l2 <= 1024 && l2 % 8 == 0 && ci % 64 == 0 && ci >= 64 && ci <= 1024 && l1 >= 64 &&
         l1 * ci < (1ll << 30) && l1 * l2 < (1ll << 30);
}

}  // namespace
}  // namespace vlfb

using namespace vlfb;

extern "C" int vlfb_attn_scores_supported(int dtype, int64_t l1, int64_t l2, int64_t ci) {
  if (!supported(dtype, l1, l2, ci)) return 0;
  // Measured on MI355X (scratch/attn_bench.py, profiles/r02_attn_fused.txt): only the queries-per-wave forward
  // kernel beats GEMM + softmax (150 us vs 213 us on 32 x 3136 x 784 x 256); the backward kernels tie or lose
  // (they read the probabilities twice), so the planner keeps the composed backward.
  int r = VLFB_ATTN_CAN_RUN;
  if (l2 == 784 && ci == 256) r |= VLFB_ATTN_FWD_FASTER;                            // 38 us vs 49 us (8 x 3136 x 784)
  if (l2 == 1024 && ci == 256) r |= VLFB_ATTN_FWD_FASTER | VLFB_ATTN_BWD_FASTER;   // 66 vs 76 us, 79 vs 87 us (8 x 4096 x 1024)
  return r;
}

extern "C" int vlfb_attn_scores_fwd(const void* theta, const void* phi, void* prob, int dtype, int64_t batch,
                                    int64_t l1, int64_t l2, int64_t ci, float scale, vlfb_stream_t stream) {
  VLFB_REQUIRE(theta && phi && prob && batch > 0, "attn_scores_fwd: bad args");
  VLFB_REQUIRE(scale > 0.f, "attn_scores_fwd: scale must be positive (the row maximum is taken before scaling)");
  if (!supported(dtype, l1, l2, ci))
    return set_error(VLFB_ERR_UNSUPPORTED, "attn_scores_fwd: needs a 16-bit dtype, 512 <= L2 <= 1024 (L2 %% 8 == 0), Ci %% 64 == 0");
  AttnP p{(const char*)theta, (const char*)phi, nullptr, (char*)prob, (int)l1, (int)l2, (int)ci, scale};
  return run_attn<false>(p, dtype, batch, (hipStream_t)stream);
}

extern "C" int vlfb_attn_scores_bwd(const void* dy, const void* g, const void* prob, void* ds, int dtype,
                                    int64_t batch, int64_t l1, int64_t l2, int64_t ci, float scale,
                                    vlfb_stream_t stream) {
  VLFB_REQUIRE(dy && g && prob && ds && batch > 0, "attn_scores_bwd: bad args");
  VLFB_REQUIRE(scale > 0.f, "attn_scores_bwd: scale must be positive");
  if (!supported(dtype, l1, l2, ci))
    return set_error(VLFB_ERR_UNSUPPORTED, "attn_scores_bwd: needs a 16-bit dtype, 512 <= L2 <= 1024 (L2 %% 8 == 0), Ci %% 64 == 0");
  AttnP p{(const char*)dy, (const char*)g, (const char*)prob, (char*)ds, (int)l1, (int)l2, (int)ci, scale};
  return run_attn<true>(p, dtype, batch, (hipStream_t)stream);
}
