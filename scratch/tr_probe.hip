// probe of ds_read_b64_tr_b16 semantics: prints what each lane receives
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s4;
__global__ void k(s4* out, int rowstride_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  short* m = reinterpret_cast<short*>(smem);
  // matrix [64 rows][rowstride/2 cols], value = row*100 + col
  const int cols = rowstride_bytes / 2;
  for (int i = threadIdx.x; i < 64 * cols; i += blockDim.x) m[i] = (short)((i / cols) * 100 + (i % cols));
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, p = lane & 15;
  // hypothesis: 16-lane group g reads block rows 8g..8g+3, cols 0..15; lane p supplies row 8g + (p>>2), cols 4*(p&3)..
  auto ptr = (__attribute__((address_space(3))) s4*)(smem + (8 * g + (p >> 2)) * rowstride_bytes + (p & 3) * 8);
  out[threadIdx.x] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
}
int main() {
  s4* d; hipMalloc(&d, 64 * sizeof(s4));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 64 * 256, 0, d, 256);
  s4 h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l][0], h[l][1], h[l][2], h[l][3]);
  return 0;
}
