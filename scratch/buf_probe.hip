// probe: does `buffer_load_dwordx4 ... offen lds` write ZEROS for out-of-range lanes? is soffset part of
// the range check?  prints per-case verdicts.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__global__ void k(const char* src, uint32_t* dst, int nbytes, int mode) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = 0xDEADBEEFu;
  __syncthreads();
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  int voff = threadIdx.x * 16;
  int soff = 0;
  if (mode == 1 && (threadIdx.x & 1)) voff = 0x80000000;        // OOB constant
  if (mode == 2) { voff = threadIdx.x * 16 - 4096; soff = 4096; }   // negative voffset + positive soffset
  if (mode == 3) { voff = threadIdx.x * 16; soff = nbytes; }    // soffset pushes past the end
  if (mode == 4 && (threadIdx.x & 1)) voff = -16;               // 0xFFFFFFF0
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(smem), 16, voff, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += blockDim.x) dst[i] = reinterpret_cast<uint32_t*>(smem)[i];
}
int main() {
  const int n = 8192;
  std::vector<uint32_t> h(n / 4);
  for (int i = 0; i < n / 4; ++i) h[i] = 0x1000 + i;
  char* d; uint32_t* o;
  hipMalloc(&d, n); hipMalloc(&o, 1024);
  hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
  for (int mode = 0; mode <= 4; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, o, mode == 3 ? 1024 : n, mode);
    std::vector<uint32_t> r(256);
    hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    printf("mode %d: lane0 %08x %08x  lane1 %08x %08x  lane2 %08x  lane63 %08x\n", mode, r[0], r[1], r[4], r[5], r[8], r[252]);
  }
  return 0;
}
