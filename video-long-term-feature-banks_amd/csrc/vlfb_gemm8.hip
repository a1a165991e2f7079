// 256-row, phase-pipelined implicit-GEMM kernels for the MFMA-bound layers (gfx950 only).
//
// The 128x128 kernels of vlfb_gemm.hip synchronise the whole workgroup once per k-tile and drain the
// DMA queue at that barrier: measured 40 % of wave time parked at s_waitcnt/s_barrier and the MFMA
// pipe 27 % busy (profiles/r01_pmc_nt_kernel_res5_3x3.txt).  Here the k-loop is software-pipelined the
// CDNA4 way (cdna_hip_programming.md, "256^2 8-phase template", T2-T5):
//   * tile 256 (positions) x BN (256 or 128 channels) x 64 (k), 8 waves as 2 (m) x 4 (n): a wave owns
//     128 x BN/4 outputs = 32 / 16 accumulator fragments, so an LDS fragment byte feeds twice the MFMAs
//     of the 64 x 32 wave tile of the old kernel;
//   * a k-tile is consumed in PHASES of 16 MFMAs (a 64 x 32 quadrant of the wave tile x 64 k); each
//     phase = { ds_read the fragments it needs; issue ONE half-tile (128 rows x 128 B = 16 KiB) of DMA
//     for a k-tile 1-2 tiles ahead; counted s_waitcnt vmcnt(N); s_barrier; 16 MFMAs; s_barrier }.
//     vmcnt is never 0 inside the loop: four half-tiles (64 KiB per CU) stay in flight across the
//     barriers;
//   * the LDS ring is managed per HALF-TILE SLOT, the slots being the row sets that are read in the same
//     phase (A rows of quadrant row 0 / 1 of both wave rows, B rows of quadrant column 0 / 1 of all four
//     wave columns), so a slot is refilled two phases after its last ds_read (the WAR distance that is
//     safe with staggered wave groups) and has 4-5 phases of flight time before its next read;
//   * the two wave rows run staggered by one barrier: while waves 0-3 issue MFMAs, waves 4-7 (same
//     SIMDs) issue their LDS reads / DMA, and vice versa; s_setprio(1) around the MFMA cluster lets the
//     matrix wave win issue arbitration;
//   * operands are addressed through buffer descriptors (32-bit offsets, hardware zero-fill for padding
//     taps, rows past M, channels past Cn and k past K); a gathered conv keeps ONE validity bit per
//     (row, filter tap) in a register and a scalar tap cursor per staging stream, so the per-k-tile
//     gather arithmetic is 2 VALU instructions per 16-byte piece;
//   * LDS rows are 128 B with the 16-byte chunk XOR-swizzled by row & 7 on the SOURCE side (the DMA
//     destination is lane-linear), fragment reads are conflict-free ds_read_b128;
//   * the fp32 accumulator tile leaves through LDS (swizzled) so that bias / residual / ReLU / mask and
//     the store are 16 bytes per lane on whole output rows.
// The accumulation order (k ascending, 32 k per MFMA, fp32) is the one of the 128x128 kernel, so both
// kernels produce bit-identical outputs (tests/test_kernels_gpu.py::test_nt8_matches_nt128_bitwise).
#include "vlfb_gemm_common.h"
#include "vlfb_gemm_nt8.h"

namespace vlfb {
namespace {

// (gemm_nt8_kernel: vlfb_gemm_nt8.h)


// =============================================================================================
// TN: O[pp][qq] = sum_m P[m][pp] * X[m][qq]  (conv wgrad of 1x1x1 layers, attention "contract over
// positions" products), 256 x 256 output tile, 64 positions per k-tile, same phase pipeline as the NT
// kernel above.  Both operands lie [position][channel] in HBM; a k-tile is staged as four 16 KiB
// sub-tiles [64 positions][128 channels] (the channels a quadrant row / column of ALL waves needs, so
// a sub-tile is again "what one phase reads" and is refilled two phases after its last read), copied
// as they lie (DMA, 32-byte segments XOR-swizzled on the source side) and read TRANSPOSED by
// ds_read_b64_tr_b16.  Split along positions: fp32 slabs + wgrad_reduce_kernel (vlfb_gemm.hip).
// =============================================================================================
template <typename T, typename OutT>
__global__ __launch_bounds__(512) void gemm_tn8_kernel(const GP p) {
  typedef typename V16<T>::V vec_t;
  constexpr int HT = 64 * 256;                   // bytes of one sub-tile (64 positions x 128 channels)
  constexpr int BUFSZ = 4 * HT;                  // P0 P1 Q0 Q1
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave >> 2, wq = wave & 3;
  const int l15 = lane & 15, g = lane >> 4;

  const int nwg = p.tiles_m * p.tiles_n;
  int bid, split;
  if (p.splits > 1 && (p.splits & 7) == 0) {     // XCD-grouped: all tiles of one split share an L2
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    bid = slot % nwg;
    split = (slot / nwg) * 8 + xcd;
  } else {
    bid = xcd_remap(blockIdx.x, nwg);
    split = blockIdx.y;
  }
  const int tile_p = bid / p.tiles_n, tile_q = bid - tile_p * p.tiles_n;
  const int p0 = tile_p * 256, q0 = tile_q * 256;
  const int z = blockIdx.z;
  const char* Pb = p.P + (long long)z * p.p_bs * 2;
  const char* Ab = p.A + (long long)z * p.a_bs * 2;
  const int kbeg = split * p.kper;
  const int kend = min(p.M, kbeg + p.kper);
  const int T_ = (kend - kbeg + 63) >> 6;

  // ---- staging: piece (row r0 + 32 i, 16-byte slot tid & 15) of every sub-tile ---------------------------
  const int r0 = tid >> 4;                                      // 0..31
  const int slot16 = tid & 15;
  const int cch = ((((slot16 >> 1) ^ tr_key<256>(r0)) << 1) | (slot16 & 1));   // global chunk (8 channels) of the sub-tile
  // buffer extents end at position kend: rows past the end of this split are zero-filled by the range check
  const __amdgpu_buffer_rsrc_t rsP = make_rsrc(Pb, (unsigned)kend * (unsigned)p.ldp * 2u);
  const __amdgpu_buffer_rsrc_t rsQ = make_rsrc(Ab, (unsigned)kend * (unsigned)p.lda * 2u);
  unsigned poff[2][2], qoff[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int pch = p0 + (cch >> 3) * 128 + h * 64 + (cch & 7) * 8;
    const int qch = q0 + (cch >> 2) * 64 + h * 32 + (cch & 3) * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = kbeg + r0 + 32 * i;
      poff[h][i] = pch < p.Ncols ? (unsigned)(row * p.ldp + pch) * 2u : kOOB;
      qoff[h][i] = qch < p.K ? (unsigned)(row * p.lda + qch) * 2u : kOOB;
    }
  }
  int pkt[2] = {0, 0}, qkt[2] = {0, 0};
  auto stage_p = [&](int h, int buf) {
    const int kt = pkt[h];
    const bool live = kt < T_;
    char* dst = smem + buf * BUFSZ + h * HT + wave * 1024;
    const unsigned step = (unsigned)(kt * 64) * (unsigned)p.ldp * 2u;
#pragma unroll
    for (int i = 0; i < 2; ++i) bufglds16_hidden(rsP, live ? poff[h][i] : kOOB, step, dst + i * 8192);
    pkt[h] = kt + 1;
  };
  auto stage_q = [&](int h, int buf) {
    const int kt = qkt[h];
    const bool live = kt < T_;
    char* dst = smem + buf * BUFSZ + (2 + h) * HT + wave * 1024;
    const unsigned step = (unsigned)(kt * 64) * (unsigned)p.lda * 2u;
#pragma unroll
    for (int i = 0; i < 2; ++i) bufglds16_hidden(rsQ, live ? qoff[h][i] : kOOB, step, dst + i * 8192);
    qkt[h] = kt + 1;
  };

  f32x4_v acc[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = f32x4_v{0.f, 0.f, 0.f, 0.f};
  vec_t pf[4][2], qf[2][2][2];

  auto read_p = [&](const char* buf, int ph) {
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) pf[ii][ks] = tr_frag<256, vec_t>(buf + ph * HT, wp * 64 + ii * 16, ks, lane);
  };
  auto read_q = [&](const char* buf, int qh) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) qf[qh][jj][ks] = tr_frag<256, vec_t>(buf + (2 + qh) * HT, wq * 32 + jj * 16, ks, lane);
  };
  auto mma_quadrant = [&](int ph, int qh) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
          acc[qh * 2 + jj][ph * 4 + ii] = V16<T>::mma(qf[qh][jj][ks], pf[ii][ks], acc[qh * 2 + jj][ph * 4 + ii]);
    __builtin_amdgcn_s_setprio(0);
  };
#define VLFB_PHASE_TAIL(N_INFLIGHT, PH, QH)           \
  VLFB_VMCNT(N_INFLIGHT);                             \
  VLFB_BAR();                                         \
  __builtin_amdgcn_sched_barrier(0);                  \
  mma_quadrant(PH, QH);                               \
  __builtin_amdgcn_sched_barrier(0);                  \
  VLFB_BAR();                                         \
  __builtin_amdgcn_sched_barrier(0)

  // prologue, in steady-state issue order: P0(0) Q0(0) Q1(0) P1(0) P0(1) Q0(1)
  stage_p(0, 0); stage_q(0, 0); stage_q(1, 0); stage_p(1, 0); stage_p(0, 1); stage_q(0, 1);
  VLFB_VMCNT(8);
  VLFB_BAR();
  if (wp == 1) VLFB_BAR();
  __builtin_amdgcn_sched_barrier(0);
  int cb = 0;
  for (int kt = 0; kt < T_; ++kt) {
    const char* bc = smem + cb * BUFSZ;
    const int nb = cb ^ 1;
    read_p(bc, 0); read_q(bc, 0);
    stage_q(1, nb);
    VLFB_PHASE_TAIL(8, 0, 0);
    read_q(bc, 1);
    stage_p(1, nb);
    VLFB_PHASE_TAIL(8, 0, 1);
    read_p(bc, 1);
    stage_p(0, cb);
    VLFB_PHASE_TAIL(8, 1, 1);
    stage_q(0, cb);
    VLFB_PHASE_TAIL(8, 1, 0);
    cb = nb;
  }
  if (wp == 0) VLFB_BAR();
  VLFB_VMCNT(0);
#undef VLFB_PHASE_TAIL

  // ---- epilogue: lane holds qq = qb + 0..3 of output row pp ---------------------------------------------
  const bool vec_ok = (p.ldo & 3) == 0;
  const bool to_ws = p.splits > 1;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int pp = p0 + wp * 128 + i * 16 + l15;
    if (pp >= p.Ncols) continue;
    const float rs = (to_ws || !p.rowscale) ? 1.f : p.rowscale[pp];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int qb = q0 + wq * 64 + (j >> 1) * 32 + (j & 1) * 16 + g * 4;
      if (qb >= p.K) continue;
      const int cnt = (p.K - qb) < 4 ? (p.K - qb) : 4;
      const long long idx = (long long)pp * p.ldo + qb;
      float v[4];
      if (to_ws) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[j][i][r];
        store4<float>(reinterpret_cast<char*>(p.ws), (long long)split * ((long long)p.Ncols * p.ldo) + idx, v, cnt, vec_ok);
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

template <typename K>
void launch8(K kernel, const GP& gp, dim3 grid, size_t lds, hipStream_t s) {
  static bool configured = false;   // per template instance
  if (!configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    configured = true;
  }
  hipLaunchKernelGGL(kernel, grid, dim3(512), lds, s, gp);
}

template <typename T, typename OutT, int BN, int RV>
void launch_nt8_bn(const GP& gp, int mode, dim3 grid, hipStream_t s) {
  constexpr size_t lds = (BN == 256 ? 2 : 3) * (size_t)(2 + BN / 128) * 16384;
  const bool ktail = (gp.K & 63) != 0;
  if (mode == 0) {
    if (ktail) launch8(gemm_nt8_kernel<T, OutT, BN, 0, true, RV>, gp, grid, lds, s);
    else launch8(gemm_nt8_kernel<T, OutT, BN, 0, false, RV>, gp, grid, lds, s);
  } else if (mode == 1) {
    launch8(gemm_nt8_kernel<T, OutT, BN, 1, false, RV>, gp, grid, lds, s);
  } else {
    launch8(gemm_nt8_kernel<T, OutT, BN, 2, false, RV>, gp, grid, lds, s);
  }
}
template <typename T, typename OutT>
void launch_nt8_t(const GP& gp, int mode, int bm, int bn, dim3 grid, hipStream_t s) {
  if (bn == 256) {
    if (bm == 196) launch_nt8_bn<T, OutT, 256, 98>(gp, mode, grid, s); else launch_nt8_bn<T, OutT, 256, 128>(gp, mode, grid, s);
  } else {
    if (bm == 196) launch_nt8_bn<T, OutT, 128, 98>(gp, mode, grid, s); else launch_nt8_bn<T, OutT, 128, 128>(gp, mode, grid, s);
  }
}

}  // namespace

int launch_nt8(const GP& gp, int bm, int bn, int mode, int dtype, bool out_f32, unsigned batch, hipStream_t s) {
  const dim3 grid((unsigned)(gp.tiles_m * gp.tiles_n), 1, batch);
  if (dtype == VLFB_F16) {
    if (out_f32) launch_nt8_t<f16_t, float>(gp, mode, bm, bn, grid, s); else launch_nt8_t<f16_t, f16_t>(gp, mode, bm, bn, grid, s);
  } else {
    if (out_f32) launch_nt8_t<bf16_t, float>(gp, mode, bm, bn, grid, s); else launch_nt8_t<bf16_t, bf16_t>(gp, mode, bm, bn, grid, s);
  }
  return check_launch("conv kernel (256-row pipelined)");
}

int launch_tn8(const GP& gp, dim3 grid, int dtype, bool out_f32, hipStream_t s) {
  constexpr size_t lds = 2 * 4 * 16384;
  if (dtype == VLFB_F16) {
    if (out_f32) launch8(gemm_tn8_kernel<f16_t, float>, gp, grid, lds, s); else launch8(gemm_tn8_kernel<f16_t, f16_t>, gp, grid, lds, s);
  } else {
    if (out_f32) launch8(gemm_tn8_kernel<bf16_t, float>, gp, grid, lds, s); else launch8(gemm_tn8_kernel<bf16_t, bf16_t>, gp, grid, lds, s);
  }
  return check_launch("conv wgrad kernel (256x256 pipelined)");
}

}  // namespace vlfb
