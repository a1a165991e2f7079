// Whole-row WGRAD for the 64 -> 64 channel 3x3 / 3x1x1 convolutions of res2 (gfx950 only).
//
//   dW[co][a][b][c][ci] = sum_pos G[pos][co] * X[pos + (a - pt, b - ph, c - pw)][ci]        (unit stride, same size)
//
// The generic TN kernel cuts the K = taps x 64 gathered columns into 128-column tiles; every column tile re-reads
// the gradient G and gathers its own shifted copy of X (per launch 2-3x the algorithmic bytes through L2, 5 VALU
// per MFMA of gather arithmetic; 129 us alone / 335 us under the concurrent dgrad stream for the res2 3x3 at 8
// clips: 0.8 M positions, 206 MB).  Both operands are THIN here (64 channels = 128 bytes per position), and the whole
// 64 x K gradient (K <= 640) fits the registers of one workgroup.  So, as in stem_wgrad_kernel:
//   * a workgroup walks whole OUTPUT ROWS (n, t, h) of Wr <= 64 positions; per row it stages, by contiguous DMA, the
//     gradient row [Wr][64] and the kt x kh input rows that row's taps touch, each as [Ws + kw - 1][64] with zero
//     halo positions (out-of-range pieces are requested at the out-of-range offset: the DMA writes zeros), both
//     in the 32-byte-segment XOR-swizzled layout ds_read_b64_tr_b16 reads conflict-free;
//   * a tap's shift along w is an LDS ROW OFFSET (position + c), its shift along t / h selects one of the staged
//     rows: no gather arithmetic, no masks, every operand byte comes from HBM once (neighbouring rows re-read the
//     input rows they share out of L1 / L2); two output rows are consumed per barrier, double-buffered (128 KiB),
//     their four k-steps software-pipelined (fragments of k-step u + 1 requested before the MFMAs of k-step u);
//     the DMAs are hidden from the compiler (bufglds16_hidden) and fragment addresses are computed once per lane;
//   * 8 waves x CT column tiles of 16 x all 64 output channels: the gradient lives in CT x 4 accumulator fragments
//     per wave across the workgroup's rows; fp32 slabs + wgrad_reduce_kernel as for every split WGRAD.
#include "vlfb_gemm_common.h"

namespace vlfb {
namespace {

// NR = kt * kh staged input rows per output row; NS = output rows in the LDS ring (NS - 1 in flight while one is
// consumed: a row is 4 x 8 KiB, the DMA latency is several rows of MFMA work)
// NW = waves per workgroup (8: one 128 KiB workgroup per CU; 4: two 64 KiB workgroups per CU that drift apart, so one's
// LDS reads overlap the other's MFMAs instead of all waves of a CU meeting at the same barrier)
// TS = output rows consumed per barrier (their 2 * TS k-steps form ONE software pipeline: the fragments of k-step
// u + 1 are requested before the MFMAs of k-step u, so the LDS round trip is exposed once per barrier, not per row)
template <typename T, int CT, int NR, int NS, int NW, int TS>
__global__ __launch_bounds__(NW * 64) void wgrad_rows_kernel(const GP p) {
  typedef typename V16<T>::V vec_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4, pl = lane & 15;
  constexpr int xrb = 64 * 128;                    // one staged row: 64 positions (Ws + kw - 1 used) x 128 bytes
  constexpr int goff_lds = NR * xrb;               // gradient tile behind the input rows
  constexpr int stage = goff_lds + xrb;
  const int split = blockIdx.x;
  const int tiles_total = p.tiles_m, tpw = p.kper;
  const int tile_beg = split * tpw;
  const int tile_end = min(tiles_total, tile_beg + tpw);

  const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.A, p.a_bytes);
  const __amdgpu_buffer_rsrc_t rsG = make_rsrc(p.P, p.b_bytes);

  // ---- per-lane DMA assignment: piece id -> (position row, 16-byte slot); the source chunk is XOR-swizzled by the
  //      LDS row so that the transposed reads are conflict-free (layout of gemm_tn_tr_kernel, 128-byte rows) ----
  constexpr int PP = 8 / NW;                       // pieces per thread and staged row (512 pieces of 16 bytes)
  bool x_ok[PP];
  unsigned xvoff[PP], gvoff[PP];
#pragma unroll
  for (int i = 0; i < PP; ++i) {
    const int id = tid + i * NW * 64;
    const int prow = id >> 3, slot = id & 7;
    const int pc = (((slot >> 1) ^ tr_key<128>(prow)) << 1) | (slot & 1);
    const int xw = prow - p.pw;                                   // source w of staged position `prow`
    x_ok[i] = xw >= 0 && xw < p.Ws;
    xvoff[i] = (unsigned)(xw * p.lda + pc * 8) * 2u;              // + row base (wraps for the halo: masked by x_ok)
    gvoff[i] = prow < p.Wr ? (unsigned)(prow * p.ldp + pc * 8) * 2u : kOOB;
  }

  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
  // every wave issues exactly PP * (NR + 1) DMA instructions per row (pieces that do not exist are requested out of range
  // and arrive as zeros), so the counted waits below are the same for all waves
  auto load_tile = [&](int tile, int buf) {
    const bool tile_ok = tile < tile_end;
    const int h = tile % p.Hr;
    const int nt = tile / p.Hr;
    const int t = nt % p.Tr, n = nt / p.Tr;
    const unsigned base = lds0 + buf * stage + wave * 1024;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int a = r / p.kh, b = r - a * p.kh;
      const int tin = t + a - p.pt, hin = h + b - p.ph;
      const bool ok = tile_ok && (unsigned)tin < (unsigned)p.Ts && (unsigned)hin < (unsigned)p.Hs;
      const unsigned rowbase = (unsigned)(((n * p.Ts + tin) * p.Hs + hin) * p.Ws) * (unsigned)p.lda * 2u;
#pragma unroll
      for (int i = 0; i < PP; ++i)
        bufglds16_hidden(rsX, (ok && x_ok[i]) ? rowbase + xvoff[i] : kOOB, 0u, base + r * xrb + i * NW * 1024);
    }
    const unsigned gstep = (unsigned)(tile * p.Wr) * (unsigned)p.ldp * 2u;
#pragma unroll
    for (int i = 0; i < PP; ++i)
      bufglds16_hidden(rsG, tile_ok ? gvoff[i] : kOOB, tile_ok ? gstep : 0u, base + goff_lds + i * NW * 1024);
  };

  f32x4_v acc[CT][4];
#pragma unroll
  for (int j = 0; j < CT; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_v{0.f, 0.f, 0.f, 0.f};

  // this wave's column tiles: ct = wave * CT + j  ->  tap = ct >> 2 (four 16-channel tiles per tap), c16 = ct & 3;
  // tap = (a * kh + b) * kw + c  ->  staged row a * kh + b, row shift c
  const int ct0 = wave * CT;
  const int ncts = p.K >> 4;
  int qrow[CT], qshift[CT], qseg[CT];
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int ct = min(ct0 + j, ncts - 1);         // tiles past the end of K recompute the last one, dropped at the store
    const int tap = ct >> 2;
    const int ab = tap / p.kw;
    qrow[j] = ab * xrb;
    qshift[j] = tap - ab * p.kw;
    qseg[j] = ct & 3;
  }
  // ---- fragment addresses, once: they depend on the lane, the k-step and the column tile, not on the row ----------
  // (computed inside the row loop they were 5 VALU instructions per MFMA -- more issue time than the MFMAs)
  constexpr int KS = 2;                                      // k-steps of 32 positions (Wr <= 64)
  int poff[KS][4][2], qoff[KS][CT][2];
  bool live[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int r0 = ks * 32 + 8 * g + (pl >> 2);              // position read by this lane (first half; second: + 4)
    live[ks] = ks * 32 + 8 * g < p.Wr;                       // Wr % 8 == 0: the 8 positions of a group live or die together
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      poff[ks][i][0] = goff_lds + r0 * 128 + ((i ^ tr_key<128>(r0)) << 5) + ((pl & 3) << 3);
      poff[ks][i][1] = goff_lds + (r0 + 4) * 128 + ((i ^ tr_key<128>(r0 + 4)) << 5) + ((pl & 3) << 3);
    }
    // dead positions read position Wr-1 instead (finite data x the zeroed gradient fragment)
    const int q0 = min(r0, p.Wr - 1), q1 = min(r0 + 4, p.Wr - 1);
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      const int j0 = q0 + qshift[j], j1 = q1 + qshift[j];
      qoff[ks][j][0] = qrow[j] + j0 * 128 + ((qseg[j] ^ tr_key<128>(j0)) << 5) + ((pl & 3) << 3);
      qoff[ks][j][1] = qrow[j] + j1 * 128 + ((qseg[j] ^ tr_key<128>(j1)) << 5) + ((pl & 3) << 3);
    }
  }
  auto tr8 = [](const char* a0, const char* a1) {
    union { struct { s16x4_v a, b; } s; vec_t v; } u;
    u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_v*)a0);
    u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_v*)a1);
    return u.v;
  };

  // slot s of the ring = TS consecutive output rows, each with its own NR + 1 staged rows
  auto load_group = [&](int tile, int slot) {
#pragma unroll
    for (int q = 0; q < TS; ++q) load_tile(tile + q, slot * TS + q);
  };
  auto read_unit = [&](const char* rows, int u, vec_t (&pf)[4], vec_t (&qf)[CT]) {
    const char* r = rows + (u >> 1) * stage;
    const int ks = u & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) pf[i] = tr8(r + poff[ks][i][0], r + poff[ks][i][1]);
#pragma unroll
    for (int j = 0; j < CT; ++j) qf[j] = tr8(r + qoff[ks][j][0], r + qoff[ks][j][1]);
  };
#pragma unroll
  for (int i = 0; i < NS - 1; ++i) load_group(tile_beg + i * TS, i);
  int buf = 0;
  for (int tile = tile_beg; tile < tile_end; tile += TS) {
    // group `tile` has landed when at most the NS - 2 younger groups are outstanding; the barrier also says every
    // wave is done reading the slot of the previous group, which the group NS - 1 ahead now overwrites
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * (NR + 1) * PP * TS) : "memory");
    load_group(tile + (NS - 1) * TS, buf == 0 ? NS - 1 : buf - 1);
    const char* rows = smem + buf * (TS * stage);
    // rows past the end of this workgroup's range were requested out of range: zero gradient rows, zero contribution
    vec_t pf[2][4], qf[2][CT];
    read_unit(rows, 0, pf[0], qf[0]);
#pragma unroll
    for (int u = 0; u < 2 * TS; ++u) {
      if (u + 1 < 2 * TS) read_unit(rows, u + 1, pf[(u + 1) & 1], qf[(u + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (!live[u & 1]) pf[u & 1][i] = vec_t{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = V16<T>::mma(qf[u & 1][j], pf[u & 1][i], acc[j][i]);
      __builtin_amdgcn_sched_barrier(0);
    }
    buf = buf == NS - 1 ? 0 : buf + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the zero-fill DMAs of the tail

  // ---- slab of this workgroup: ws[split][co][K], lane holds 4 consecutive columns of row co -------
  float* slab = p.ws + (long long)split * ((long long)p.Ncols * p.ldo);
#pragma unroll
  for (int j = 0; j < CT; ++j) {
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

// =============================================================================================
// The same idea for the 3x1x1 convs with a FAT input (res2 branch2a: 256 -> 64 channels):
//   dW[co][a][ci] = sum_pos G[pos][co] * X[pos + (a - pt) * H * W][ci]
//                 = sum_pos' X[pos'][ci] * G[pos' - (a - pt) * H * W][co]
// -- shift the THIN operand.  A workgroup walks input rows (n, t, h): the 512-byte-per-position X row is staged once
// (NG = Cs / 64 channel groups of [64 positions][128 B]) next to the kt gradient rows of the frames t - (a - pt)
// (rows of frames that do not exist are requested out of range: zeros).  Wave w owns input channels
// [32 w, 32 w + 32): two 16-channel tiles x kt taps x all 64 output channels = 6 x 4 accumulator fragments.  X is read
// from HBM exactly once, the gradient rows kt times out of L1 / L2 (the generic kernel reads X once per 128-column
// tile of the gradient -- 6 times -- and was at 2.3 TB/s of algorithmic bytes).
// =============================================================================================
template <typename T, int NG, int KT>
__global__ __launch_bounds__(512) void wgrad_rows_fat_kernel(const GP p) {
  typedef typename V16<T>::V vec_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(NG == 4, "8 waves x 32 channels");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4, pl = lane & 15;
  constexpr int slotb = 64 * 128;                  // one staged [64 positions][128 B] piece
  constexpr int goff_lds = NG * slotb;             // gradient rows behind the X groups
  constexpr int stage = (NG + KT) * slotb;
  constexpr int NS = 2;
  const int split = blockIdx.x;
  const int tiles_total = p.tiles_m, tpw = p.kper;
  const int tile_beg = split * tpw;
  const int tile_end = min(tiles_total, tile_beg + tpw);
  const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.A, p.a_bytes);
  const __amdgpu_buffer_rsrc_t rsG = make_rsrc(p.P, p.b_bytes);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));

  const int prow = tid >> 3, slot = tid & 7;
  const int pc = (((slot >> 1) ^ tr_key<128>(prow)) << 1) | (slot & 1);
  const bool row_ok = prow < p.Wr;
  const unsigned xvoff = (unsigned)(prow * p.lda + pc * 8) * 2u;      // + group * 128 + row base
  const unsigned gvoff = (unsigned)(prow * p.ldp + pc * 8) * 2u;
  const int frame = p.Hs * p.Ws;                                       // positions per frame

  // every wave issues exactly NG + KT DMA instructions per row
  auto load_tile = [&](int tile, int buf) {
    const bool tile_ok = tile < tile_end;
    const int h = tile % p.Hs;
    const int nt = tile / p.Hs;
    const int t = nt % p.Ts, n = nt / p.Ts;
    const unsigned base = lds0 + buf * stage + wave * 1024;
    const unsigned xbase = (unsigned)(tile * p.Ws) * (unsigned)p.lda * 2u;
#pragma unroll
    for (int c = 0; c < NG; ++c)
      bufglds16_hidden(rsX, (tile_ok && row_ok) ? xbase + xvoff + c * 128u : kOOB, 0u, base + c * slotb);
#pragma unroll
    for (int a = 0; a < KT; ++a) {
      const int tg = t - (a - p.pt);                                   // frame of the gradient row tap a pairs with
      const bool ok = tile_ok && row_ok && (unsigned)tg < (unsigned)p.Tr;
      const unsigned gbase = (unsigned)(((n * p.Tr + tg) * p.Hr + h) * p.Wr) * (unsigned)p.ldp * 2u;
      bufglds16_hidden(rsG, ok ? gbase + gvoff : kOOB, 0u, base + goff_lds + a * slotb);
    }
  };
  (void)frame;

  f32x4_v acc[2 * KT][4];
#pragma unroll
  for (int j = 0; j < 2 * KT; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_v{0.f, 0.f, 0.f, 0.f};

  // fragment addresses, once per lane: the gradient fragments of tap a are the ones of tap 0 + a * slotb
  constexpr int KS = 2;
  int poff[KS][4][2], qoff[KS][2][2];
  const int cg = wave >> 1;                                            // channel group of this wave's 32 channels
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int r0 = ks * 32 + 8 * g + (pl >> 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      poff[ks][i][0] = goff_lds + r0 * 128 + ((i ^ tr_key<128>(r0)) << 5) + ((pl & 3) << 3);
      poff[ks][i][1] = goff_lds + (r0 + 4) * 128 + ((i ^ tr_key<128>(r0 + 4)) << 5) + ((pl & 3) << 3);
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int seg = ((wave & 1) << 1) | e;
      qoff[ks][e][0] = cg * slotb + r0 * 128 + ((seg ^ tr_key<128>(r0)) << 5) + ((pl & 3) << 3);
      qoff[ks][e][1] = cg * slotb + (r0 + 4) * 128 + ((seg ^ tr_key<128>(r0 + 4)) << 5) + ((pl & 3) << 3);
    }
  }
  auto tr8 = [](const char* a0, const char* a1) {
    union { struct { s16x4_v a, b; } s; vec_t v; } u;
    u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_v*)a0);
    u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_v*)a1);
    return u.v;
  };

  load_tile(tile_beg, 0);
  int buf = 0;
  for (int tile = tile_beg; tile < tile_end; ++tile) {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    load_tile(tile + 1, buf ^ 1);
    const char* rows = smem + buf * stage;
    // positions past the row end were staged as zeros on both sides: no masks
    vec_t qf[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int e = 0; e < 2; ++e) qf[ks][e] = tr8(rows + qoff[ks][e][0], rows + qoff[ks][e][1]);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      vec_t pf[KT][4];
#pragma unroll
      for (int a = 0; a < KT; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) pf[a][i] = tr8(rows + a * slotb + poff[ks][i][0], rows + a * slotb + poff[ks][i][1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < KT; ++a)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[a * 2 + e][i] = V16<T>::mma(qf[ks][e], pf[a][i], acc[a * 2 + e][i]);
      __builtin_amdgcn_sched_barrier(0);
    }
    buf ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the zero-fill DMAs of the tail

  // ---- slab: ws[split][co][a * Cs + ci], lane holds 4 consecutive ci of row co --------------------------------
  float* slab = p.ws + (long long)split * ((long long)p.Ncols * p.ldo);
#pragma unroll
  for (int a = 0; a < KT; ++a)
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int co = i * 16 + l15;
        const int col = a * p.Cs + (wave * 2 + e) * 16 + g * 4;
        *reinterpret_cast<float4*>(slab + (long long)co * p.ldo + col) =
            make_float4(acc[a * 2 + e][i][0], acc[a * 2 + e][i][1], acc[a * 2 + e][i][2], acc[a * 2 + e][i][3]);
      }
}

template <typename K>
void launch_rows(K kernel, const GP& gp, unsigned splits, unsigned threads, size_t lds, hipStream_t s) {
  static bool configured = false;   // per template instance
  if (!configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    configured = true;
  }
  hipLaunchKernelGGL(kernel, dim3(splits), dim3(threads), lds, s, gp);
}

template <typename T>
void launch_rows_t(const GP& gp, int ct, unsigned splits, hipStream_t s) {
  // 8 waves, one workgroup per CU: ring of 2 groups x 2 output rows x (3 + 1) staged rows x 8 KiB = 128 KiB
  constexpr size_t kLds = 2 * 2 * 4 * 64 * 128;
  if (ct == 2) launch_rows(wgrad_rows_kernel<T, 2, 3, 2, 8, 2>, gp, splits, 512, kLds, s);
  else launch_rows(wgrad_rows_kernel<T, 5, 3, 2, 8, 2>, gp, splits, 512, kLds, s);
}

}  // namespace

// column tiles per wave (of an 8-wave workgroup) the kernel would use for K gathered columns (0: not supported)
int wgrad_rows_ct(long long K) {
  const long long ncts = K / 16;
  if (K % 64 || ncts < 1) return 0;
  return ncts <= 16 ? 2 : ncts <= 40 ? 5 : 0;
}

int launch_wgrad_rows_fat(const GP& gp, int splits, int dtype, hipStream_t s) {
  constexpr size_t kLds = 2 * (4 + 3) * 64 * 128;       // 2 rows x (4 X groups + 3 gradient rows) x 8 KiB
  if (dtype == VLFB_F16) launch_rows(wgrad_rows_fat_kernel<f16_t, 4, 3>, gp, (unsigned)splits, 512, kLds, s);
  else launch_rows(wgrad_rows_fat_kernel<bf16_t, 4, 3>, gp, (unsigned)splits, 512, kLds, s);
  return check_launch("conv wgrad (whole rows, fat input) kernel");
}

int launch_wgrad_rows(const GP& gp, int splits, size_t lds, int dtype, hipStream_t s) {
  (void)lds;
  const int ct = wgrad_rows_ct(gp.K);
  if (dtype == VLFB_F16) launch_rows_t<f16_t>(gp, ct, (unsigned)splits, s);
  else launch_rows_t<bf16_t>(gp, ct, (unsigned)splits, s);
  return check_launch("conv wgrad (whole rows) kernel");
}

}  // namespace vlfb
