// Weight-resident streaming implicit-GEMM kernel for the HBM-bound conv layers (gfx950 only).
//
// The res2 / res3 bottleneck convolutions of the 8-clip step have 0.8 M (res2) / 0.1-0.4 M (res3) output
// positions and 64-512 channels: 13-60 FLOP per HBM byte against a machine balance of ~400, so they are
// bound by HBM, not by the matrix cores (profiles/r02_per_launch_*.txt: 2.3-4.7 TB/s with the tiled
// kernels, whose workgroup-wide k-tile barriers serialise "load a tile / compute / write a tile").  Their
// whole weight matrix is at most 128 KiB.  So here
//   * one persistent 8-wave workgroup per CU copies the WHOLE weight operand into LDS once (DMA, 128-byte
//     k-tile rows XOR-swizzled like the tiled kernels), then never synchronises again;
//   * every wave streams its own blocks of 16 / 32 output positions: the activation fragments go straight
//     from HBM / L2 into MFMA operand registers (lane l holds 8 consecutive k of position l & 15 -- the
//     MFMA layout is also a legal global access pattern, 64 contiguous bytes per position and
//     instruction), the next chunk of k-tiles is requested before the MFMAs of the current one, and the
//     residual / ReLU-mask rows of a block are requested BEFORE its MFMAs, so a wave keeps 10-20 KiB in
//     flight and eight independent waves per CU cover the HBM latency without any barrier;
//   * weight rows sit in LDS in a permuted order (LDS row q*32 + e*16 + s*4 + r holds channel
//     q*32 + s*8 + e*4 + r), so that the accumulators of two neighbouring 16-channel fragments give a lane
//     EIGHT consecutive channels of one position: bias / residual / ReLU / mask and the 16-byte store
//     happen in registers, no LDS round trip of the output tile;
//   * the XCD that owns a workgroup owns one contiguous stripe of positions, so the neighbouring
//     positions a 3x3 / 3x1x1 tap reaches were fetched into the same L2 shortly before.
// k is accumulated in ascending order, 32 k per v_mfma_f32_16x16x32, fp32 -- the order of the 128x128 and
// 256-row kernels -- and the epilogue applies alpha, bias, residual, ReLU, mask in their order, so all
// three families are bit-identical (tests/test_stream_gpu.py).
#include "vlfb_gemm_common.h"

namespace vlfb {
namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_v;

__device__ __forceinline__ u32x4_v bufld16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
  return __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 0);
}
__device__ __forceinline__ void bufst16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, u32x4_v v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)voff, 0, 0);
}

// NF: 16-channel fragments (NF * 16 = output channels: 64, 128 or 256);
// MF: 16-position fragments per row block; UK: k-tiles (64 k) per load chunk; GATHER: conv taps (FPROP of
// any stride, unit-stride DGRAD) instead of plain rows; NW: waves per workgroup (8, or 16 where 128 VGPRs are enough).
template <typename T, int NF, int MF, int UK, bool GATHER, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_nts_kernel(const GP p, const int dgrad, const int lgN) {
  typedef typename V16<T>::V vec_t;
  constexpr int RB = 16 * MF;                    // positions per block
  constexpr int NQ = NF / 2;                     // 8-channel groups per lane
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int N = p.Ncols;
  const int nkt = p.K >> 6;
  const int nchunk = nkt / UK;

  // ---- the weight operand -> LDS, once per workgroup: [k-tile][permuted channel][128 B, chunk ^ (row & 7)] ----
  {
    const auto rsB = make_rsrc(p.B, p.b_bytes);
    const int total = nkt << (lgN + 3);          // 16-byte pieces
    for (int base = wave * 64; base < total; base += NW * 64) {
      const int piece = base + lane;
      const int s = piece & 7, r = (piece >> 3) & (N - 1), kt = piece >> (lgN + 3);
      const int c = (r & ~31) | (((r >> 2) & 3) << 3) | (((r >> 4) & 1) << 2) | (r & 3);
      const unsigned off = (unsigned)(c * p.ldb + kt * 64 + ((s ^ (r & 7)) << 3)) * 2u;
      bufglds16(rsB, piece < total ? off : kOOB, 0, smem + base * 16);
    }
    float* bl = reinterpret_cast<float*>(smem + ((size_t)nkt << (lgN + 7)));
    if (tid < N) bl[tid] = p.bias_mode == VLFB_BIAS_COL ? p.bias[tid] : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  const char* bias_lds = smem + ((size_t)nkt << (lgN + 7));

  const auto rsA = make_rsrc(p.A, p.a_bytes);
  const unsigned obytes = (unsigned)p.M * (unsigned)p.ldo * 2u, rbytes = (unsigned)p.M * (unsigned)p.ldr * 2u;
  const auto rsO = make_rsrc(p.O, obytes);
  const auto rsR = make_rsrc(p.R ? p.R : p.O, p.R ? rbytes : obytes);
  const auto rsM = make_rsrc(p.Mask ? p.Mask : p.O, p.Mask ? rbytes : obytes);
  const bool hasR = p.R != nullptr, hasM = p.Mask != nullptr;

  // ---- which blocks: XCD x owns stripe x of the positions, its workgroups' waves take the blocks round-robin --
  const int nblk = (p.M + RB - 1) / RB;
  const int xcd = blockIdx.x & 7, wx = blockIdx.x >> 3, nwx = gridDim.x >> 3;
  const int S = (nblk + 7) >> 3;
  const int stride = nwx * NW;
  const int blk0 = xcd * S;
  int li = wx * NW + wave;

  // per block row: byte offset of its first tap (wraps for padding rows) and, per filter dimension, one validity
  // bit per tap index (a tap exists iff its three bits are set) -- kt + kh + kw compares per row, not kt * kh * kw
  struct Rows { unsigned aoff[MF]; unsigned vt[MF], vh[MF], vw[MF]; };
  struct Cur { int a, b, c, ci; };
  const int sgn = dgrad ? -1 : 1;
  // row -> (n, t, h, w) by reciprocal multiplication (exact below 2^24 rows after the +-1 fix-up)
  const int hw = p.Hr * p.Wr, thw = p.Tr * hw;
  const float inv_thw = 1.0f / (float)thw, inv_hw = 1.0f / (float)hw, inv_w = 1.0f / (float)p.Wr;
  auto divmod = [](int x, int d, float inv, int& rem) {
    int q = (int)((float)x * inv);
    rem = x - q * d;
    if (rem < 0) { --q; rem += d; }
    else if (rem >= d) { ++q; rem -= d; }
    return q;
  };

  auto decode = [&](int blk, bool live, Rows& rw) {
#pragma unroll
    for (int f = 0; f < MF; ++f) {
      const int m = blk * RB + f * 16 + l15;
      const bool ok = live && m < p.M;
      if (!GATHER) {
        rw.aoff[f] = ok ? (unsigned)(m * p.lda + g * 8) * 2u : kOOB;
        rw.vt[f] = ok ? 1u : 0u;
        rw.vh[f] = rw.vw[f] = 1u;
      } else {
        int rem, rem2, w;
        const int n = divmod(ok ? m : 0, thw, inv_thw, rem);
        int t = divmod(rem, hw, inv_hw, rem2);
        int h = divmod(rem2, p.Wr, inv_w, w);
        if (!dgrad) { t = t * p.st - p.pt; h = h * p.sh - p.ph; w = w * p.sw - p.pw; }
        else { t += p.pt; h += p.ph; w += p.pw; }
        const int pix = ((n * p.Ts + t) * p.Hs + h) * p.Ws + w;
        rw.aoff[f] = (unsigned)(pix * p.lda + g * 8) * 2u;          // wraps for padding rows (masked by the bits)
        unsigned bt = 0, bh = 0, bw = 0;
#pragma clang loop vectorize(disable) unroll(disable)
        for (int a = 0; a < p.kt; ++a) bt |= ((unsigned)(t + sgn * a * p.dt) < (unsigned)p.Ts ? 1u : 0u) << a;
#pragma clang loop vectorize(disable) unroll(disable)
        for (int b = 0; b < p.kh; ++b) bh |= ((unsigned)(h + sgn * b * p.dh) < (unsigned)p.Hs ? 1u : 0u) << b;
#pragma clang loop vectorize(disable) unroll(disable)
        for (int c = 0; c < p.kw; ++c) bw |= ((unsigned)(w + sgn * c * p.dw) < (unsigned)p.Ws ? 1u : 0u) << c;
        rw.vt[f] = ok ? bt : 0u;
        rw.vh[f] = bh;
        rw.vw[f] = bw;
      }
    }
  };
  auto load_chunk = [&](vec_t (&dst)[MF][UK][2], const Rows& rw, Cur& cu) {
#pragma unroll
    for (int u = 0; u < UK; ++u) {
      unsigned dbyte;
      if (GATHER) {
        const int pix = sgn * ((cu.a * p.dt * p.Hs + cu.b * p.dh) * p.Ws + cu.c * p.dw);
        dbyte = (unsigned)(pix * p.lda * 2 + cu.ci);
      } else {
        dbyte = (unsigned)cu.ci;
      }
#pragma unroll
      for (int f = 0; f < MF; ++f) {
        const bool ok = GATHER ? (((rw.vt[f] >> cu.a) & (rw.vh[f] >> cu.b) & (rw.vw[f] >> cu.c)) & 1u) != 0u : rw.vt[f] != 0u;
        const unsigned voff = ok ? rw.aoff[f] + dbyte : kOOB;
        dst[f][u][0] = __builtin_bit_cast(vec_t, bufld16(rsA, voff));
        dst[f][u][1] = __builtin_bit_cast(vec_t, bufld16(rsA, voff + 64u));
      }
      // cursor advance by selects (a branch between the loads would make the compiler's wait counts pessimistic)
      if (GATHER) {
        const bool wc = cu.ci + 128 >= p.Cs * 2;
        cu.ci = wc ? 0 : cu.ci + 128;
        const int c1 = cu.c + (wc ? 1 : 0);
        const bool wb = c1 == p.kw;
        cu.c = wb ? 0 : c1;
        const int b1 = cu.b + (wb ? 1 : 0);
        const bool wa = b1 == p.kh;
        cu.b = wa ? 0 : b1;
        cu.a += wa ? 1 : 0;
      } else {
        cu.ci += 128;
      }
    }
  };

  // fragment addressing in the weight image
  const int key = l15 & 7;
  const int kof0 = ((0 + g) ^ key) << 4, kof1 = ((4 + g) ^ key) << 4;
  const int ktile_bytes = N << 7;

  Rows rw_cur, rw_nxt;
  Cur cu = {0, 0, 0, 0};
  vec_t cur[MF][UK][2], nxt[MF][UK][2];
  bool live = li < S && blk0 + li < nblk;
  decode(blk0 + li, live, rw_cur);
  load_chunk(cur, rw_cur, cu);

  while (live) {
    const int blk = blk0 + li;
    const int nli = li + stride;
    const bool nlive = nli < S && blk0 + nli < nblk;
    decode(blk0 + nli, nlive, rw_nxt);
    {
      // residual / mask rows of this block: requested now, used after the MFMAs
      // (always issued: an absent operand is read at the out-of-range offset, which returns zeros without a
      // memory access -- no branches around loads, so the compiler can count what is in flight)
      u32x4_v rreg[MF][NQ], mreg[MF][NQ];
#pragma unroll
      for (int f = 0; f < MF; ++f) {
        const int m = blk * RB + f * 16 + l15;
        const unsigned ro = m < p.M ? (unsigned)(m * p.ldr + g * 8) * 2u : kOOB;
        const unsigned ror = hasR ? ro : kOOB, rom = hasM ? ro : kOOB;
#pragma unroll
        for (int q = 0; q < NQ; ++q) rreg[f][q] = bufld16(rsR, ror + q * 64u);
#pragma unroll
        for (int q = 0; q < NQ; ++q) mreg[f][q] = bufld16(rsM, rom + q * 64u);
      }
      f32x4_v acc[MF][NF];
#pragma unroll
      for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[f][j] = f32x4_v{0.f, 0.f, 0.f, 0.f};
      const char* wrow = smem + ((l15) << 7);
      for (int ch = 0; ch < nchunk; ++ch) {
        // The next chunk of this wave's stream goes in flight first: the next k-tiles of this block, or the
        // first ones of the wave's next block.  ONE load site, no branch around it (a second, conditional site
        // makes the compiler drain the queue before re-using the registers).
        {
          const bool same = ch + 1 < nchunk;
          Rows sel;
#pragma unroll
          for (int f = 0; f < MF; ++f) {
            sel.aoff[f] = same ? rw_cur.aoff[f] : rw_nxt.aoff[f];
            sel.vt[f] = same ? rw_cur.vt[f] : rw_nxt.vt[f];
            sel.vh[f] = same ? rw_cur.vh[f] : rw_nxt.vh[f];
            sel.vw[f] = same ? rw_cur.vw[f] : rw_nxt.vw[f];
          }
          if (!same) cu = Cur{0, 0, 0, 0};
          load_chunk(nxt, sel, cu);
        }
        // weight fragments: groups of four ds_read_b128, the next group requested before the MFMAs of this one
        const char* wk = wrow + (size_t)(ch * UK) * ktile_bytes;
        constexpr int JG = NF / 4;                // groups per k-step
        constexpr int NG = UK * 2 * JG;
        vec_t bq[2][4];
        auto read_group = [&](int gi, vec_t (&dst)[4]) {
          const int u = gi / (2 * JG), ks = (gi / JG) & 1, jg = gi % JG;
          const char* wf = wk + u * ktile_bytes + (ks ? kof1 : kof0) + jg * 4 * 2048;
#pragma unroll
          for (int j = 0; j < 4; ++j) dst[j] = *reinterpret_cast<const vec_t*>(wf + j * 2048);
        };
        read_group(0, bq[0]);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
          if (gi + 1 < NG) read_group(gi + 1, bq[(gi + 1) & 1]);
          const int u = gi / (2 * JG), ks = (gi / JG) & 1, jg = gi % JG;
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int f = 0; f < MF; ++f) acc[f][jg * 4 + j] = V16<T>::mma(bq[gi & 1][j], cur[f][u][ks], acc[f][jg * 4 + j]);
        }
#pragma unroll
        for (int f = 0; f < MF; ++f)
#pragma unroll
          for (int u = 0; u < UK; ++u) { cur[f][u][0] = nxt[f][u][0]; cur[f][u][1] = nxt[f][u][1]; }
      }
      // ---- epilogue in registers: lane = position l15, channels q*32 + g*8 .. +7 -----------------------
#pragma unroll
      for (int f = 0; f < MF; ++f) {
        const int m = blk * RB + f * 16 + l15;
        const unsigned oo = m < p.M ? (unsigned)(m * p.ldo + g * 8) * 2u : kOOB;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[e * 4 + r] = __fmul_rn(acc[f][2 * q + e][r], p.alpha);
          if (p.bias_mode == VLFB_BIAS_COL) {
            const float4 b0 = *reinterpret_cast<const float4*>(bias_lds + ((q * 32 + g * 8) << 2));
            const float4 b1 = *reinterpret_cast<const float4*>(bias_lds + ((q * 32 + g * 8 + 4) << 2));
            v[0] = __fadd_rn(v[0], b0.x); v[1] = __fadd_rn(v[1], b0.y); v[2] = __fadd_rn(v[2], b0.z); v[3] = __fadd_rn(v[3], b0.w);
            v[4] = __fadd_rn(v[4], b1.x); v[5] = __fadd_rn(v[5], b1.y); v[6] = __fadd_rn(v[6], b1.z); v[7] = __fadd_rn(v[7], b1.w);
          }
          if (hasR) {
            float r[8];
            const uint4 t = make_uint4(rreg[f][q].x, rreg[f][q].y, rreg[f][q].z, rreg[f][q].w);
            unpack_elems<T, 8>(t, r);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = __fadd_rn(v[e], r[e]);
          }
          if (p.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          if (hasM) {
            float r[8];
            const uint4 t = make_uint4(mreg[f][q].x, mreg[f][q].y, mreg[f][q].z, mreg[f][q].w);
            unpack_elems<T, 8>(t, r);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = r[e] > 0.f ? v[e] : 0.f;
          }
          u32x4_v o;
          o.x = Elem<T>::pack2(v[0], v[1]); o.y = Elem<T>::pack2(v[2], v[3]);
          o.z = Elem<T>::pack2(v[4], v[5]); o.w = Elem<T>::pack2(v[6], v[7]);
          bufst16(rsO, oo + q * 64u, o);
        }
      }
    }
    rw_cur = rw_nxt;
    li = nli;
    live = nlive;
  }
}

template <typename K>
int launch_s(K kernel, const GP& gp, int dgrad, int lgN, unsigned nwg, int nw, size_t lds, hipStream_t s) {
  static bool configured = false;   // per template instance
  if (!configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    configured = true;
  }
  hipLaunchKernelGGL(kernel, dim3(nwg), dim3(nw * 64), lds, s, gp, dgrad, lgN);
  return check_launch("conv kernel (weight-resident streaming)");
}

template <typename T, int NF, int MF, int NW>
int launch_nts_shape(const GP& gp, int mode, int uk, int lgN, unsigned nwg, size_t lds, hipStream_t s) {
  const int dgrad = mode == 2;
  if (mode == 0) {
    if (uk == 1) return launch_s(gemm_nts_kernel<T, NF, MF, 1, false, NW>, gp, 0, lgN, nwg, NW, lds, s);
    if (uk == 2) return launch_s(gemm_nts_kernel<T, NF, MF, 2, false, NW>, gp, 0, lgN, nwg, NW, lds, s);
    return launch_s(gemm_nts_kernel<T, NF, MF, 4, false, NW>, gp, 0, lgN, nwg, NW, lds, s);
  }
  if (uk == 3) return launch_s(gemm_nts_kernel<T, NF, MF, 3, true, NW>, gp, dgrad, lgN, nwg, NW, lds, s);
  return launch_s(gemm_nts_kernel<T, NF, MF, 4, true, NW>, gp, dgrad, lgN, nwg, NW, lds, s);
}

template <typename T>
int launch_nts_t(const GP& gp, int mode, int uk, int lgN, unsigned nwg, size_t lds, hipStream_t s) {
  if (gp.Ncols == 64) return launch_nts_shape<T, 4, 1, 16>(gp, mode, uk, lgN, nwg, lds, s);
  if (gp.Ncols == 128) return launch_nts_shape<T, 8, 1, 8>(gp, mode, uk, lgN, nwg, lds, s);
  return launch_nts_shape<T, 16, 1, 8>(gp, mode, uk, lgN, nwg, lds, s);
}

}  // namespace

int nts_chunk(int mode, long long K) {
  const long long nkt = K / 64;
  if (K % 64) return 0;
  if (mode == 0) return nkt % 4 == 0 ? 4 : nkt % 2 == 0 ? 2 : 1;
  return nkt % 3 == 0 ? 3 : nkt % 4 == 0 ? 4 : 0;
}

int launch_nts(const GP& gp, int mode, int dtype, hipStream_t s) {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, n = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
    ncu = n / 8 * 8;
  }
  const int uk = nts_chunk(mode, gp.K);
  int lgN = 0;
  while ((1 << lgN) < gp.Ncols) ++lgN;
  const int rb = 16, nw = gp.Ncols == 64 ? 16 : 8;
  const long long nblk = ((long long)gp.M + rb - 1) / rb;
  long long nwg = (nblk + nw - 1) / nw;                 // every wave of a workgroup wants a block
  nwg = (nwg + 7) / 8 * 8;
  if (nwg > ncu) nwg = ncu;
  const size_t lds = (size_t)gp.Ncols * gp.K * 2 + (size_t)gp.Ncols * 4;
  if (dtype == VLFB_F16) return launch_nts_t<f16_t>(gp, mode, uk, lgN, (unsigned)nwg, lds, s);
  return launch_nts_t<bf16_t>(gp, mode, uk, lgN, (unsigned)nwg, lds, s);
}

}  // namespace vlfb
