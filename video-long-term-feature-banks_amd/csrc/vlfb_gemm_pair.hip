// Two-plane fp16 implicit GEMM (vlfb_conv_desc.math = VLFB_MATH_F16X3): the FORWARD contractions of the "mix" path.
//
// A value is stored as two fp16 planes, v = hi + lo with hi = fp16(v), lo = fp16(v - hi) (~22 significant bits; the
// fp16 MFMA keeps subnormal operands -- probed on MI355X, scratch/r6/pl_probe.py -- so small activations lose nothing
// but the 2^-24 absolute floor of the lo term).  A product is hi.hi + hi.lo + lo.hi on v_mfma_f32_16x16x32_f16 with
// fp32 accumulation: three MFMAs per product like the split-bf16 form (vlfb_gemm_split.hip), but ~16x more exact per
// product (2^-21 against 2^-17) and with NOTHING to convert in the k-loop: the planes are what the producing epilogue
// stored (O = hi, O_lo = lo of vlfb_conv_args) and what vlfb_weight_prep wrote (VLFB_MIXH*, scaled by 2^10 so that
// the lo plane of a weight stays in the normal range; alpha carries 2^-10).  The kernel is the tuned 128-row NT kernel
// of the 16-bit paths (vlfb_gemm_nt.h, PAIR): LDS-DMA of 128-byte rows = 32 k of hi | the same 32 k of lo, one bare
// barrier per k-tile, XOR-swizzled rows, the LDS-staged 16-byte epilogue with two-term residual / output.
#include "vlfb_gemm_nt.h"
#include "vlfb_gemm_nt8.h"

namespace vlfb {
namespace {

template <typename K>
void launch_pair_k(K kernel, dim3 grid, int threads, size_t lds, const GP& gp, hipStream_t s) {
  static bool configured = false;  // per template instance
  if (!configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    configured = true;
  }
  hipLaunchKernelGGL(kernel, grid, dim3(threads), lds, s, gp);
}

template <typename OutT, int BN, bool PRE>
void launch_pair_shape(const GP& gp, bool ident, dim3 grid, size_t lds, hipStream_t s) {
  constexpr int NW = 8;
  if (ident) launch_pair_k(gemm_nt_kernel<f16_t, OutT, 128, BN, true, false, false, 128, PRE, NW, 2, false, true>, grid, 64 * NW, lds, gp, s);
  else launch_pair_k(gemm_nt_kernel<f16_t, OutT, 128, BN, false, false, false, 128, PRE, NW, 2, true, true>, grid, 64 * NW, lds, gp, s);
}

// (a non-type template parameter: one `configured` flag per KERNEL -- every kernel here has the same function type)
template <auto Kernel>
void launch_pair8_k(const GP& gp, dim3 grid, size_t lds, hipStream_t s) {
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    configured = true;
  }
  hipLaunchKernelGGL(Kernel, grid, dim3(512), lds, s, gp);
}
template <typename OutT, int BN, int RV>
void launch_pair8_bn(const GP& gp, int mode, dim3 grid, hipStream_t s) {
  constexpr size_t lds = (BN == 256 ? 2 : 3) * (size_t)(2 + BN / 128) * 16384;
  if (mode == 0) launch_pair8_k<gemm_nt8_kernel<f16_t, OutT, BN, 0, false, RV, true>>(gp, grid, lds, s);
  else launch_pair8_k<gemm_nt8_kernel<f16_t, OutT, BN, 1, false, RV, true>>(gp, grid, lds, s);
}
template <typename OutT>
void launch_pair8_t(const GP& gp, int mode, int bm, int bn, dim3 grid, hipStream_t s) {
  if (bn == 256) {
    if (bm == 196) launch_pair8_bn<OutT, 256, 98>(gp, mode, grid, s); else launch_pair8_bn<OutT, 256, 128>(gp, mode, grid, s);
  } else {
    if (bm == 196) launch_pair8_bn<OutT, 128, 98>(gp, mode, grid, s); else launch_pair8_bn<OutT, 128, 128>(gp, mode, grid, s);
  }
}

}  // namespace

// the 256-row phase-pipelined form (vlfb_gemm_nt8.h): mode 0 plain rows, 1 gathered FPROP; bm 256 | 196, bn 256 | 128
int launch_nt8_pair(const GP& gp, int bm, int bn, int mode, bool out_f32, hipStream_t s) {
  const dim3 grid((unsigned)(gp.tiles_m * gp.tiles_n), 1, 1);
  if (out_f32) launch_pair8_t<float>(gp, mode, bm, bn, grid, s); else launch_pair8_t<f16_t>(gp, mode, bm, bn, grid, s);
  return check_launch("conv kernel (fp16 planes, 256-row pipelined)");
}

int launch_nt_pair(const GP& gp, int bn, bool ident, bool pre, bool out_f32, dim3 grid, size_t lds, hipStream_t s) {
  if (out_f32) {
    if (bn == 64) launch_pair_shape<float, 64, false>(gp, ident, grid, lds, s);
    else launch_pair_shape<float, 128, false>(gp, ident, grid, lds, s);
  } else if (pre) {
    if (bn == 64) launch_pair_shape<f16_t, 64, true>(gp, ident, grid, lds, s);
    else launch_pair_shape<f16_t, 128, true>(gp, ident, grid, lds, s);
  } else {
    if (bn == 64) launch_pair_shape<f16_t, 64, false>(gp, ident, grid, lds, s);
    else launch_pair_shape<f16_t, 128, false>(gp, ident, grid, lds, s);
  }
  return check_launch("conv kernel (fp16 planes)");
}

}  // namespace vlfb
