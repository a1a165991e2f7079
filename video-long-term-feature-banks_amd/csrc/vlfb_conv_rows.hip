// 1x3x3 convolutions with 64 -> 64 channels (res2 branch2b: FPROP and the unit-stride DGRAD) as a direct
// convolution on the matrix cores, weights resident and input rows rolling through LDS (gfx950 only).
//
// At 8 clips these layers have 0.8 M positions, K = 9 x 64 = 576 and N = 64: 24 us of MFMA work and ~55 us of HBM
// traffic, but the 128x64 tiled kernel takes 140 us -- with 64 output channels a wave owns a 32x32 register tile
// (1 KB of fragment reads per MFMA) and the im2col'd activation tile, every pixel nine times, is written into LDS
// first: it is bound by LDS bandwidth, like the stem (vlfb_stem.hip).  Here
//   * the whole weight matrix (72 KB) is copied into LDS once per workgroup;
//   * a workgroup walks down the rows of its frames, four output rows per step: the six input rows a step touches
//     live in a ring of ten 8 KB row slots (row pitch 64 pixels x 128 B, pixel 0 and Wr + 1 are the zero halo, the
//     rows above / below the frame arrive as zeros from the buffer range check), and the four rows of the NEXT step
//     are copied in (LDS DMA, contiguous) while this step computes -- every input pixel is fetched exactly once;
//   * wave (r, q) owns output row r of the step x the 32 channels q*32 .. +31: 4 position fragments x 2 channel
//     fragments, the activation fragment of tap (b, c) is just the staged row shifted by b rows and c pixels
//     (ds_read_b128, 128-byte pixel rows XOR-swizzled by the pixel index like the tile rows of the tiled kernels);
//   * weight rows use the permuted channel order and the swizzle of vlfb_stem.hip: two neighbouring fragments give
//     a lane 8 consecutive channels of a position, so alpha / bias / residual / ReLU / mask and the 16-byte store
//     happen in registers.
// DGRAD (stride 1) is the same walk with the taps mirrored (source pixel = position + pad - tap).  k runs in
// ascending tap order, 32 k per v_mfma_f32_16x16x32, and the epilogue order is that of the tiled kernel: the
// outputs are bit-identical (tests/test_conv_rows_gpu.py).
#include "vlfb_gemm_common.h"

namespace vlfb {
namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_r;

constexpr int kRowsStep = 4;          // output rows per step
constexpr int kRing = 10;             // input-row slots: 6 live + 4 arriving
constexpr int kRowPitch = 8192;       // 64 pixels x 128 B

// TM = false: 1x3x3 (frames = (n, t), rows = h, taps reach rows h - 1 .. h + 1 and pixels w - 1 .. w + 1).
// TM = true: 3x1x1 (res2_0 branch2a): the same walk with the roles turned -- "frames" = (n, h), "rows" = t, the three
// taps reach rows t - 1 .. t + 1 of the same pixel; consecutive rows are a whole frame apart in memory.
template <typename T, bool DG, bool TM>
__global__ __launch_bounds__(512) void conv_rows64_kernel(const GP p, const int nframes, const int fpw) {
  typedef typename V16<T>::V vec_t;
  constexpr int MT = 4;
  constexpr int NKS = TM ? 6 : 18;      // k-steps: taps x 2 halves of the 64 input channels
  constexpr int kWBytes = NKS * 4096;   // NKS k-steps x 64 channels x 64 B
  const int nrow = TM ? p.Tr : p.Hr;    // rows a frame has (source and output: same-size convs)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int wr = wave >> 1, wq = wave & 1;              // output row of the step, channel half
  char* ring = smem + kWBytes;
  float* bias_l = reinterpret_cast<float*>(smem + kWBytes + kRing * kRowPitch);

  const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.A, p.a_bytes);
  const __amdgpu_buffer_rsrc_t rsW = make_rsrc(p.B, p.b_bytes);
  const unsigned obytes = (unsigned)p.M * (unsigned)p.ldo * 2u, rbytes = (unsigned)p.M * (unsigned)p.ldr * 2u;
  const __amdgpu_buffer_rsrc_t rsO = make_rsrc(p.O, obytes);
  const __amdgpu_buffer_rsrc_t rsR = make_rsrc(p.R ? p.R : p.O, p.R ? rbytes : obytes);
  const __amdgpu_buffer_rsrc_t rsM = make_rsrc(p.Mask ? p.Mask : p.O, p.Mask ? rbytes : obytes);
  const bool hasR = p.R != nullptr, hasM = p.Mask != nullptr;

  // ---- weights -> LDS, once: [k-step][permuted channel row][64 B, chunk ^ key(row)] --------------------------
  {
    const unsigned wbase = lds_addr_of(smem);
#pragma unroll
    for (int i = 0; i < NKS / 2; ++i) {                 // NKS * 256 pieces / 512 threads
      const int id = tid + 512 * i;
      const int chunk = id & 3, r = (id >> 2) & 63, ks = id >> 8;
      const int c = (r & ~31) | (((r >> 2) & 3) << 3) | (((r >> 4) & 1) << 2) | (r & 3);
      const unsigned off = (unsigned)(c * p.ldb + ks * 32 + (chunk ^ ((0 - (r >> 2)) & 3)) * 8) * 2u;
      bufglds16_hidden(rsW, off, 0u, (unsigned)__builtin_amdgcn_readfirstlane((int)(wbase + (i * 512 + wave * 64) * 16)));
    }
    if (tid < 64) bias_l[tid] = p.bias_mode == VLFB_BIAS_COL ? p.bias[tid] : 0.f;
  }
  __syncthreads();                                      // the bias (the weight DMAs are waited for with the first rows)

  // ---- per-lane row DMA: piece tid of every row = pixel slot tid >> 3 (pixel w = slot - 1), chunk slot tid & 7 ----
  const int dp = tid >> 3, dslot = tid & 7;
  const bool dpx = dp >= 1 && dp <= p.Wr;
  const unsigned dsrc = (unsigned)((dp - 1) * p.lda * 2 + ((dslot ^ (dp & 7)) << 4));
  const unsigned ring_base = lds_addr_of(ring);
  const int rowbytes = p.Ws * p.lda * 2;                // one source row
  auto load_row = [&](int frame, int i) {               // input row i of the frame (any integer) -> slot (i + 1) % 10
    const bool ok = dpx && (unsigned)i < (unsigned)nrow && frame < nframes;
    // spatial: row i of frame (n, t) = line (n * T + t) * H + i; temporal: row t = i of "frame" (n, h) = line (n * T + i) * H + h
    const int line = TM ? ((frame / p.Hs) * p.Ts + i) * p.Hs + frame % p.Hs : frame * p.Hs + i;
    const unsigned off = ok ? (unsigned)(line * rowbytes) + dsrc : kOOB;
    const int slot = (i + 1 + kRing) % kRing;
    bufglds16_hidden(rsX, off, 0u, (unsigned)__builtin_amdgcn_readfirstlane((int)(ring_base + slot * kRowPitch + wave * 1024)));
  };

  // ---- fragment addressing --------------------------------------------------------------------------------
  // weights: fragment j of this wave's channel half = LDS rows wq*32 + j*16 + l15
  const int w_lane = (wq * 32 + l15) * 64 + ((g ^ ((0 - (l15 >> 2)) & 3)) << 4);

  const int steps = (nrow + kRowsStep - 1) / kRowsStep;
  const int wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int f_beg = wg * fpw, f_end = min(nframes, f_beg + fpw);

  for (int frame = f_beg; frame < f_end; ++frame) {
    // prime the ring: rows -1 .. 4 of the frame (all waves are past the previous frame's last step: barrier below)
    asm volatile("s_barrier" ::: "memory");
#pragma unroll
    for (int i = -1; i <= 4; ++i) load_row(frame, i);
    for (int st = 0; st < steps; ++st) {
      const int h0 = st * kRowsStep;
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      const bool more = st + 1 < steps;

      // residual / mask rows of this wave's outputs: requested now, used after the MFMAs (an absent operand is read
      // at the out-of-range offset: zeros, no memory access, no branch around the loads)
      const int h = h0 + wr;
      const int row0 = (TM ? ((frame / p.Hr) * p.Tr + h) * p.Hr + frame % p.Hr : frame * p.Hr + h) * p.Wr;
      u32x4_r rq[MT], mq[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int w = m * 16 + l15;
        const unsigned ro = (w < p.Wr && h < nrow) ? (unsigned)((row0 + w) * p.ldr + wq * 32 + g * 8) * 2u : kOOB;
        rq[m] = __builtin_amdgcn_raw_buffer_load_b128(rsR, (int)(hasR ? ro : kOOB), 0, 0);
        mq[m] = __builtin_amdgcn_raw_buffer_load_b128(rsM, (int)(hasM ? ro : kOOB), 0, 0);
      }

      f32x4_v acc[MT][2];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[m][j] = f32x4_v{0.f, 0.f, 0.f, 0.f};

      // k-step ks = tap * 2 + half, tap = b * 3 + c: input row h0 + wr + (DG ? 1 - b : b - 1), pixel slot w + (DG ? 2 - c : c)
      vec_t wf[2][2], af[2][MT];
      auto read_step = [&](int ks, vec_t (&w)[2], vec_t (&x)[MT]) {
        const int tap = ks >> 1, hf = ks & 1, b = TM ? tap : tap / 3, c = TM ? 1 : tap - b * 3;
        const int ib = DG ? 2 - b : b, ic = TM ? 1 : (DG ? 2 - c : c);
        const int slot = (h0 + wr + ib) % kRing;          // row h0 + wr + ib - 1  ->  slot (row + 1) % 10
        const char* row = ring + slot * kRowPitch;
#pragma unroll
        for (int j = 0; j < 2; ++j) w[j] = *reinterpret_cast<const vec_t*>(smem + ks * 4096 + j * 1024 + w_lane);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int px = m * 16 + l15 + ic;
          x[m] = *reinterpret_cast<const vec_t*>(row + px * 128 + (((hf * 4 + g) ^ (px & 7)) << 4));
        }
      };
      read_step(0, wf[0], af[0]);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        if (ks + 1 < NKS) read_step(ks + 1, wf[(ks + 1) & 1], af[(ks + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[m][j] = V16<T>::mma(wf[ks & 1][j], af[ks & 1][m], acc[m][j]);
        __builtin_amdgcn_sched_barrier(0);
        // the four rows of the next step, spread over the k-steps
        if (TM) { if (more && ks < 4) load_row(frame, h0 + 5 + ks); }
        else if (more && (ks & 3) == 1 && ks < 16) load_row(frame, h0 + 5 + (ks >> 2));
        __builtin_amdgcn_sched_barrier(0);
      }

      // ---- epilogue in registers: lane = position m*16 + l15 of row h, channels wq*32 + g*8 .. +7 -----------------
      if (h < nrow) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int w = m * 16 + l15;
          const bool live = w < p.Wr;
          const unsigned oo = live ? (unsigned)((row0 + w) * p.ldo + wq * 32 + g * 8) * 2u : kOOB;
          float v[8];
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[e * 4 + r] = __fmul_rn(acc[m][e][r], p.alpha);
          if (p.bias_mode == VLFB_BIAS_COL) {
            const float4 b0 = *reinterpret_cast<const float4*>(bias_l + wq * 32 + g * 8);
            const float4 b1 = *reinterpret_cast<const float4*>(bias_l + wq * 32 + g * 8 + 4);
            v[0] = __fadd_rn(v[0], b0.x); v[1] = __fadd_rn(v[1], b0.y); v[2] = __fadd_rn(v[2], b0.z); v[3] = __fadd_rn(v[3], b0.w);
            v[4] = __fadd_rn(v[4], b1.x); v[5] = __fadd_rn(v[5], b1.y); v[6] = __fadd_rn(v[6], b1.z); v[7] = __fadd_rn(v[7], b1.w);
          }
          if (hasR) {
            float r[8];
            unpack_elems<T, 8>(make_uint4(rq[m].x, rq[m].y, rq[m].z, rq[m].w), r);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = __fadd_rn(v[e], r[e]);
          }
          if (p.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          if (hasM) {
            float r[8];
            unpack_elems<T, 8>(make_uint4(mq[m].x, mq[m].y, mq[m].z, mq[m].w), r);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = r[e] > 0.f ? v[e] : 0.f;
          }
          u32x4_r o;
          o.x = Elem<T>::pack2(v[0], v[1]); o.y = Elem<T>::pack2(v[2], v[3]);
          o.z = Elem<T>::pack2(v[4], v[5]); o.w = Elem<T>::pack2(v[6], v[7]);
          __builtin_amdgcn_raw_buffer_store_b128(o, rsO, (int)oo, 0, 0);
        }
      }
    }
  }
}

template <typename K>
int launch_rows64(K kernel, const GP& gp, int nks, int nframes, int fpw, unsigned nwg, hipStream_t s) {
  static bool configured = false;   // per template instance
  const size_t lds = (size_t)nks * 4096 + kRing * kRowPitch + 256;
  if (!configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    configured = true;
  }
  hipLaunchKernelGGL(kernel, dim3(nwg), dim3(512), lds, s, gp, nframes, fpw);
  return check_launch("conv kernel (64-channel rows, direct)");
}

}  // namespace

// 1x3x3, stride 1, pad (0, 1, 1), 64 -> 64 channels, rows of at most 62 positions, same-size source
bool conv_rows64_ok(const GP& gp, int mode, int dtype, int out_dtype, long long batch) {
  if (!(dtype == VLFB_BF16 || dtype == VLFB_F16) || out_dtype != dtype || batch != 1) return false;
  if (mode != VLFB_CONV_FPROP && mode != VLFB_CONV_DGRAD) return false;
  if (gp.Cs != 64 || gp.Ncols != 64) return false;
  const bool spatial = gp.kt == 1 && gp.kh == 3 && gp.kw == 3 && gp.K == 576 && gp.pt == 0 && gp.ph == 1 && gp.pw == 1;
  const bool temporal = gp.kt == 3 && gp.kh == 1 && gp.kw == 1 && gp.K == 192 && gp.pt == 1 && gp.ph == 0 && gp.pw == 0;
  if (!spatial && !temporal) return false;
  if (gp.st != 1 || gp.sh != 1 || gp.sw != 1 || gp.dt != 1 || gp.dh != 1 || gp.dw != 1) return false;
  if (gp.Ts != gp.Tr || gp.Hs != gp.Hr || gp.Ws != gp.Wr || gp.Wr > 62 || gp.Hr < 1) return false;
  if (gp.lda % 8 || gp.ldb % 8 || gp.ldo % 8 || gp.ldr % 8) return false;
  if ((long long)gp.M * gp.ldo * 2 >= (1ll << 31) || (long long)gp.M * gp.ldr * 2 >= (1ll << 31) ||
      (long long)gp.M * gp.lda * 2 >= (1ll << 31)) return false;
  return true;
}

int launch_conv_rows64(const GP& gp, int mode, int dtype, hipStream_t s) {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, n = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
    ncu = n / 8 * 8;
  }
  const bool tm = gp.kt == 3;
  const int nframes = tm ? gp.M / (gp.Tr * gp.Wr) : gp.M / (gp.Hr * gp.Wr);      // (n, h) slabs | (n, t) frames
  long long nwg = ((long long)nframes + 7) / 8 * 8;
  if (nwg > ncu) nwg = ncu;
  const int fpw = (int)((nframes + nwg - 1) / nwg);
  const bool dg = mode == VLFB_CONV_DGRAD;
#define VLFB_ROWS64(T_, DG_, TM_) launch_rows64(conv_rows64_kernel<T_, DG_, TM_>, gp, TM_ ? 6 : 18, nframes, fpw, (unsigned)nwg, s)
  if (dtype == VLFB_F16) {
    if (tm) return dg ? VLFB_ROWS64(f16_t, true, true) : VLFB_ROWS64(f16_t, false, true);
    return dg ? VLFB_ROWS64(f16_t, true, false) : VLFB_ROWS64(f16_t, false, false);
  }
  if (tm) return dg ? VLFB_ROWS64(bf16_t, true, true) : VLFB_ROWS64(bf16_t, false, true);
  return dg ? VLFB_ROWS64(bf16_t, true, false) : VLFB_ROWS64(bf16_t, false, false);
#undef VLFB_ROWS64
}

}  // namespace vlfb
