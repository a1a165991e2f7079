// r6 microbenchmark: bytes per clock per CU of the two ways a GEMM operand can reach the MFMA on gfx950, every CU busy, two 512-thread
// workgroups per CU (the occupancy of the 128-row kernels):  (a) buffer_load_dwordx4 ... lds (LDS-DMA, 1 KiB per wave instruction),
// (b) buffer_load_dwordx4 into VGPRs.  Footprints: 2 MiB shared by all workgroups (L2 hits: a weight operand), 64 MiB, 1 GiB (HBM).
// Access pattern of both: per instruction a wave reads 8 rows of 128 contiguous bytes (the NT kernels' tile rows), rows 4 KiB apart.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I video-long-term-feature-banks_amd/csrc -I include scratch/r6/dma_bw_probe.hip -o scratch/r6/dma_bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "vlfb_gemm_common.h"
using namespace vlfb;

template <int MODE>
__global__ __launch_bounds__(512) void probe(const char* src, unsigned bytes, int iters, unsigned* sink, int pieces) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const auto rs = make_rsrc(src, bytes);
  // lane -> (row = lane / 8, 16-byte chunk = lane % 8) of an 8-row x 128-byte piece; rows 4 KiB apart
  const unsigned lane_off = (unsigned)(lane >> 3) * 4096u + (unsigned)(lane & 7) * 16u;
  unsigned base = ((unsigned)blockIdx.x * 8u + (unsigned)wave) * 32768u;       // 32 KiB per wave and step
  uint4 acc = make_uint4(0, 0, 0, 0);
  const unsigned mask = bytes - 1;                                               // (power-of-two footprint)
  for (int it = 0; it < iters; ++it) {
#pragma unroll 1
    for (int p = 0; p < pieces; ++p) {
      const unsigned off = ((base + (unsigned)p * 128u) & mask & ~32767u) + ((unsigned)p * 128u & 4095u) + lane_off;
      if (MODE == 0) {
        bufglds16(rs, off, 0, smem + (wave * 4 + (p & 3)) * 1024);
      } else {
        const uint4 v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
      }
    }
    if (MODE == 0) { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); }
    base += gridDim.x * 8u * 32768u + 37u * 32768u;       // (a different 32-KiB block every step: no L1 re-use at any footprint)
  }
  if (MODE == 0) {
    __syncthreads();
    acc = *reinterpret_cast<const uint4*>(smem + tid * 16);
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

int main() {
  const size_t cap = 1ull << 30;
  char* buf; unsigned* sink;
  hipMalloc(&buf, cap); hipMalloc(&sink, 64);
  hipMemset(buf, 1, cap);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const double clk = prop.clockRate * 1e3;        // Hz (nominal)
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  printf("%d CUs, nominal clock %.0f MHz; 2 workgroups of 8 waves per CU, 1 KiB per wave instruction\n", cus, clk / 1e6);
  const size_t foot[3] = {2ull << 20, 64ull << 20, 1ull << 30};
  // in-flight sweep (L2-resident footprint, LDS-DMA): `pieces` KiB per wave between two waits = 16 x pieces KiB in flight per CU;
  // the 128-row GEMM kernels run at 4 (one 32-KiB tile per 8-wave workgroup, two workgroups per CU)
  for (int pieces : {2, 4, 6, 8, 12, 16, 32}) {
    const int grid = cus * 2, iters = 6400 / pieces;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(512), 64 * 1024, 0, buf, (unsigned)foot[0], iters, sink, pieces);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * 8 * pieces * 1024.0 * iters;
    printf("L2-resident, LDS-DMA, %2d KiB per wave (= %3d KiB per CU) in flight between waits: %7.2f TB/s  %6.1f B/clk/CU\n", pieces, 16 * pieces,
           bytes / ms / 1e9, bytes / (ms * 1e-3) / clk / cus);
  }
  for (int f = 0; f < 3; ++f)
    for (int mode = 0; mode < 2; ++mode) {
      const int grid = cus * 2, pieces = 32, iters = 200;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(512), 64 * 1024, 0, buf, (unsigned)foot[f], iters, sink, pieces);
        else hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(512), 64 * 1024, 0, buf, (unsigned)foot[f], iters, sink, pieces);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)grid * 8 * pieces * 1024.0 * iters;
      printf("footprint %5zu MiB  %-28s %8.3f ms  %7.2f TB/s  %6.1f B/clk/CU (nominal clock)\n", foot[f] >> 20,
             mode == 0 ? "buffer_load_dwordx4 ... lds" : "buffer_load_dwordx4 -> VGPR", ms, bytes / ms / 1e9, bytes / (ms * 1e-3) / clk / cus);
    }
  return 0;
}
