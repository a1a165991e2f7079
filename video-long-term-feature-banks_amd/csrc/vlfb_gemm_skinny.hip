// NT products with a handful of rows: O[m][n] = sum_k A[m][k] * W[n][k], M <= 64 -- the 1x1x1 convs of the FBO head on
// one row per RoI (lfb_helper.py:170-263: theta / out / reduc of the feature-bank operator, M = R ~ 33 in a training
// batch, K = 512 .. 2048, 512 columns).  On the 128 x 128 tiles such a launch is 4 workgroups walking K serially
// (20-35 us of a mostly idle chip, 10 launches per step); here a workgroup owns 16 output COLUMNS and all rows, its four
// waves split K and read their MFMA operand fragments straight from global memory (a fragment is 16 contiguous bytes
// per lane, no LDS staging), and the four partial tiles meet in LDS.  Plain rows only, 16-bit operands (bf16 / f16).
#include "vlfb_gemm_common.h"

namespace vlfb {
namespace {

template <typename T, typename OutT>
__global__ __launch_bounds__(256) void gemm_skinny_nt_kernel(const GP p) {
  constexpr int FM = 4;                               // 64 rows
  __shared__ float4 part[4][FM][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  typedef typename V16<T>::V Vv;
  const Vv zero = {};
  const T* A = reinterpret_cast<const T*>(p.A);
  const T* W = reinterpret_cast<const T*>(p.B);
  const int n = n0 + l15;
  const bool nok = n < p.Ncols;
  const T* wrow = W + (long long)(nok ? n : 0) * p.ldb + g * 8;
  const T* arow[FM];
  bool aok[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = i * 16 + l15;
    aok[i] = m < p.M;
    arow[i] = A + (long long)(aok[i] ? m : 0) * p.lda + g * 8;
  }
  f32x4_v acc[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) acc[i] = f32x4_v{0.f, 0.f, 0.f, 0.f};
  const int ksteps = p.K >> 5;
  // two k-steps in flight per wave (the loads of a step are independent of every MFMA)
  for (int ks = wave; ks < ksteps; ks += 8) {
    const int k0 = ks * 32, k1 = (ks + 4) * 32;
    const bool two = ks + 4 < ksteps;
    Vv w0 = nok ? *reinterpret_cast<const Vv*>(wrow + k0) : zero;
    Vv w1 = (nok && two) ? *reinterpret_cast<const Vv*>(wrow + k1) : zero;
    Vv x0[FM], x1[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      x0[i] = aok[i] ? *reinterpret_cast<const Vv*>(arow[i] + k0) : zero;
      x1[i] = (aok[i] && two) ? *reinterpret_cast<const Vv*>(arow[i] + k1) : zero;
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i] = V16<T>::mma(w0, x0[i], acc[i]);
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i] = V16<T>::mma(w1, x1[i], acc[i]);
  }
#pragma unroll
  for (int i = 0; i < FM; ++i) part[wave][i][lane] = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
  __syncthreads();
  // thread (wave = row fragment, lane) folds the four partial tiles in wave order and runs the epilogue of its 4 values
  const int i = wave;
  float4 s = part[0][i][lane];
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const float4 t = part[w][i][lane];
    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
  }
  const int m = i * 16 + l15;
  const int nb = n0 + g * 4;
  if (m >= p.M || nb >= p.Ncols) return;
  const int cnt = (p.Ncols - nb) < 4 ? (p.Ncols - nb) : 4;
  const float a4[4] = {s.x, s.y, s.z, s.w};
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float x = a4[r] * p.alpha;
    if (r < cnt) {
      if (p.bias_mode == VLFB_BIAS_COL) x += p.bias[nb + r];
      else if (p.bias_mode == VLFB_BIAS_ROW) x += p.bias[m];
      if (p.R) x += ld_elem<T>(p.R, (long long)m * p.ldr + nb + r);
      if (p.relu) x = fmaxf(x, 0.f);
      if (p.Mask) x = ld_elem<T>(p.Mask, (long long)m * p.ldr + nb + r) > 0.f ? x : 0.f;
    }
    v[r] = x;
  }
  store4<OutT>(p.O, (long long)m * p.ldo + nb, v, cnt, (p.ldo & 3) == 0);
}

// The same for split-bf16 math on fp32 storage (VLFB_MATH_BF16X3: the fp32 head of the "mix" and "split" paths -- theta / out /
// reduc of the feature-bank operator on one row per RoI, forward and DGRAD).  A rows are fp32: a lane's 8 values of a k-step
// (two 16-byte loads) are split into two bf16 terms in registers; W arrives as bf16 term planes GP::b_ps apart; a product is
// m.h + h.m + h.h on v_mfma_f32_16x16x32_bf16, small terms first.  fp32 output (+ its fp16 copy, o_planes = 1), fp32 residual / mask.
__device__ __forceinline__ void split2_frag(const float4 lo, const float4 hi, bf16x8_v& h, bf16x8_v& m) {
  float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  uint32_t wh[4], wm[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t hp = pack_bf2(x[2 * i], x[2 * i + 1]);
    wh[i] = hp;
    wm[i] = pack_bf2(x[2 * i] - __uint_as_float(hp << 16), x[2 * i + 1] - __uint_as_float(hp & 0xffff0000u));
  }
  h = __builtin_bit_cast(bf16x8_v, make_uint4(wh[0], wh[1], wh[2], wh[3]));
  m = __builtin_bit_cast(bf16x8_v, make_uint4(wm[0], wm[1], wm[2], wm[3]));
}

__global__ __launch_bounds__(512) void gemm_skinny_nt_sp_kernel(const GP p) {
  constexpr int FM = 4;                               // 64 rows
  constexpr int NWV = 8;                              // waves that split K (two k-steps in flight each)
  __shared__ float4 part[NWV][FM][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const bf16x8_v zero = {};
  const float* A = reinterpret_cast<const float*>(p.A);
  const bf16_t* W = reinterpret_cast<const bf16_t*>(p.B);
  const int n = n0 + l15;
  const bool nok = n < p.Ncols;
  const bf16_t* wrow = W + (long long)(nok ? n : 0) * p.ldb + g * 8;
  const float* arow[FM];
  bool aok[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = i * 16 + l15;
    aok[i] = m < p.M;
    arow[i] = A + (long long)(aok[i] ? m : 0) * p.lda + g * 8;
  }
  f32x4_v acc[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) acc[i] = f32x4_v{0.f, 0.f, 0.f, 0.f};
  const int ksteps = p.K >> 5;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int ks = wave; ks < ksteps; ks += 2 * NWV) {
    const int k0 = ks * 32, k1 = (ks + NWV) * 32;
    const bool two = ks + NWV < ksteps;
    const bf16x8_v wh0 = nok ? *reinterpret_cast<const bf16x8_v*>(wrow + k0) : zero;
    const bf16x8_v wm0 = nok ? *reinterpret_cast<const bf16x8_v*>(wrow + p.b_ps + k0) : zero;
    const bf16x8_v wh1 = (nok && two) ? *reinterpret_cast<const bf16x8_v*>(wrow + k1) : zero;
    const bf16x8_v wm1 = (nok && two) ? *reinterpret_cast<const bf16x8_v*>(wrow + p.b_ps + k1) : zero;
    float4 lo0[FM], hi0[FM], lo1[FM], hi1[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      lo0[i] = aok[i] ? *reinterpret_cast<const float4*>(arow[i] + k0) : z4;
      hi0[i] = aok[i] ? *reinterpret_cast<const float4*>(arow[i] + k0 + 4) : z4;
      lo1[i] = (aok[i] && two) ? *reinterpret_cast<const float4*>(arow[i] + k1) : z4;
      hi1[i] = (aok[i] && two) ? *reinterpret_cast<const float4*>(arow[i] + k1 + 4) : z4;
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      bf16x8_v xh, xm;
      split2_frag(lo0[i], hi0[i], xh, xm);
      acc[i] = V16<bf16_t>::mma(wm0, xh, acc[i]);
      acc[i] = V16<bf16_t>::mma(wh0, xm, acc[i]);
      acc[i] = V16<bf16_t>::mma(wh0, xh, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      bf16x8_v xh, xm;
      split2_frag(lo1[i], hi1[i], xh, xm);
      acc[i] = V16<bf16_t>::mma(wm1, xh, acc[i]);
      acc[i] = V16<bf16_t>::mma(wh1, xm, acc[i]);
      acc[i] = V16<bf16_t>::mma(wh1, xh, acc[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < FM; ++i) part[wave][i][lane] = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
  __syncthreads();
  if (wave >= FM) return;                              // waves 0-3 fold the partial tiles (in wave order) and run the epilogue
  const int i = wave;
  float4 s = part[0][i][lane];
#pragma unroll
  for (int w = 1; w < NWV; ++w) {
    const float4 t = part[w][i][lane];
    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
  }
  const int m = i * 16 + l15;
  const int nb = n0 + g * 4;
  if (m >= p.M || nb >= p.Ncols) return;
  const int cnt = (p.Ncols - nb) < 4 ? (p.Ncols - nb) : 4;
  const float a4[4] = {s.x, s.y, s.z, s.w};
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float x = a4[r] * p.alpha;
    if (r < cnt) {
      if (p.bias_mode == VLFB_BIAS_COL) x += p.bias[nb + r];
      else if (p.bias_mode == VLFB_BIAS_ROW) x += p.bias[m];
      if (p.R) x += ld_elem<float>(p.R, (long long)m * p.ldr + nb + r);
      if (p.relu) x = fmaxf(x, 0.f);
      if (p.Mask) x = ld_elem<float>(p.Mask, (long long)m * p.ldr + nb + r) > 0.f ? x : 0.f;
    }
    v[r] = x;
  }
  store4<float>(p.O, (long long)m * p.ldo + nb, v, cnt, (p.ldo & 3) == 0);
  if (p.op_n == 1) {        // the fp16 copy of the output that a 16-bit backward reads (vlfb_conv_desc.o_planes = 1; positive stays positive)
    unsigned short* oh = reinterpret_cast<unsigned short*>(p.OP) + (long long)m * p.ldo + nb;
    for (int r = 0; r < cnt; ++r) oh[r] = f2h_pos(v[r]);
  }
}

}  // namespace

// (plain fp32 rows, two-term weight planes, fp32 output without a copy: what the launch must be; make_plan checks the rest)
bool skinny_nt_split_ok(const GP& gp, long long batch, bool ident) {
  return ident && batch == 1 && !gp.accumulate && gp.M <= 64 && gp.K % 32 == 0 && gp.K >= 128 && gp.lda % 8 == 0 && gp.ldb % 8 == 0;
}
int launch_skinny_nt_split(const GP& gp, hipStream_t s) {
  const dim3 grid((unsigned)((gp.Ncols + 15) / 16));
  hipLaunchKernelGGL(gemm_skinny_nt_sp_kernel, grid, dim3(512), 0, s, gp);
  return check_launch("conv (skinny rows, split-bf16) kernel");
}

bool skinny_nt_ok(const GP& gp, int dtype, long long batch, bool ident) {
  // (accumulate is a WGRAD-only epilogue: the NT epilogue here never reads O, so a descriptor that asks for it is not taken)
  return ident && batch == 1 && is16(dtype) && !gp.accumulate && gp.M <= 64 && gp.K % 32 == 0 && gp.K >= 128 && gp.lda % 8 == 0 && gp.ldb % 8 == 0;
}

int launch_skinny_nt(const GP& gp, int dtype, bool out_f32, hipStream_t s) {
  const dim3 grid((unsigned)((gp.Ncols + 15) / 16));
  if (dtype == VLFB_BF16) {
    if (out_f32) hipLaunchKernelGGL((gemm_skinny_nt_kernel<bf16_t, float>), grid, dim3(256), 0, s, gp);
    else hipLaunchKernelGGL((gemm_skinny_nt_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, gp);
  } else {
    if (out_f32) hipLaunchKernelGGL((gemm_skinny_nt_kernel<f16_t, float>), grid, dim3(256), 0, s, gp);
    else hipLaunchKernelGGL((gemm_skinny_nt_kernel<f16_t, f16_t>), grid, dim3(256), 0, s, gp);
  }
  return check_launch("conv (skinny rows) kernel");
}

}  // namespace vlfb
