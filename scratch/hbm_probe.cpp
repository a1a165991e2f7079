// Development probe: what the chip gives to elementwise kernels with the traffic mix of the thin-K conv layers
// (read a + read b, write y) vs a plain copy, to tell a kernel-structure problem from the HBM ceiling.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "vlfb.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
__global__ void copy16(const uint4* __restrict__ a, uint4* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = a[i];
}
__global__ void add2(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    uint4 u = a[i], v = b[i];
    y[i] = make_uint4(u.x ^ v.x, u.y ^ v.y, u.z ^ v.z, u.w ^ v.w);
  }
}
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  const long long elems = 802816ll * 256;       // one res2 activation tensor, bf16: 411 MB
  void *a, *b, *y, *m;
  CK(hipMalloc(&a, elems * 2)); CK(hipMalloc(&b, elems * 2)); CK(hipMalloc(&y, elems * 2)); CK(hipMalloc(&m, elems * 2));
  CK(hipMemset(a, 0x3c, elems * 2)); CK(hipMemset(b, 0x3d, elems * 2)); CK(hipMemset(m, 0x3e, elems * 2));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, double bytes, auto fn) {
    fn(); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 10; ++r) fn();
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.1f us  %6.2f TB/s\n", name, ms * 100.0, bytes / (ms * 1e-4) * 1e-12);
  };
  const long long n16 = elems * 2 / 16;
  for (int grid : {2048, 8192, 65536}) {
    char nm[96];
    snprintf(nm, sizeof nm, "copy 411 MB -> 411 MB, grid %d", grid);
    timeit(nm, 2.0 * elems * 2, [&] { hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, s, (const uint4*)a, (uint4*)y, n16); });
    snprintf(nm, sizeof nm, "2 reads + 1 write (3 x 411 MB), grid %d", grid);
    timeit(nm, 3.0 * elems * 2, [&] { hipLaunchKernelGGL(add2, dim3(grid), dim3(256), 0, s, (const uint4*)a, (const uint4*)b, (uint4*)y, n16); });
  }
  timeit("vlfb_add a+b -> y", 3.0 * elems * 2, [&] { vlfb_add(a, b, y, nullptr, VLFB_BF16, elems, 0, s); });
  timeit("vlfb_add a+b -> y, masked (4 tensors)", 4.0 * elems * 2, [&] { vlfb_add(a, b, y, m, VLFB_BF16, elems, 0, s); });
  // the conv layer with the same epilogue traffic: res2 2c 64 -> 256, + R, ReLU  (A 103 MB + R 411 MB + O 411 MB)
  vlfb_conv_desc d; vlfb_conv_desc_init(&d);
  d.mode = VLFB_CONV_FPROP; d.N = 8; d.Tr = 32; d.Hr = 56; d.Wr = 56; d.Ts = 32; d.Hs = 56; d.Ws = 56; d.Cs = 64; d.Cn = 256; d.relu = 1;
  void *x64, *w; CK(hipMalloc(&x64, 802816ll * 64 * 2)); CK(hipMalloc(&w, 256 * 64 * 2));
  CK(hipMemset(x64, 0x3c, 802816ll * 64 * 2)); CK(hipMemset(w, 0x2c, 256 * 64 * 2));
  timeit("conv res2 2c 64->256 + R + relu (925 MB)", 802816.0 * (64 + 256 + 256) * 2, [&] { vlfb_conv_run(&d, x64, w, nullptr, y, nullptr, nullptr, b, nullptr, nullptr, 0, s); });
  timeit("conv res2 2c 64->256 no R (514 MB)", 802816.0 * (64 + 256) * 2, [&] { vlfb_conv_run(&d, x64, w, nullptr, y, nullptr, nullptr, nullptr, nullptr, nullptr, 0, s); });
  d.mode = VLFB_CONV_DGRAD; d.Cs = 256; d.Cn = 64;     // 2c dgrad: G 256 ch -> dX 64 ch, mask
  void* w2; CK(hipMalloc(&w2, 256 * 64 * 2)); CK(hipMemset(w2, 0x2c, 256 * 64 * 2));
  timeit("conv res2 2c dgrad 256->64 + mask (617 MB)", 802816.0 * (256 + 64 + 64) * 2, [&] { vlfb_conv_run(&d, a, w2, nullptr, x64, nullptr, nullptr, nullptr, x64, nullptr, 0, s); });
  return 0;
}
