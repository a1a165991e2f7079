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

}  // namespace

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
