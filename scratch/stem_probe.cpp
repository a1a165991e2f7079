// Development probe (not product, not a test): the packed stem FPROP through vlfb_conv_run with algo = TILE128 (the
// 128x64 tiled kernel) and algo = AUTO (the direct-convolution kernel of vlfb_stem.hip); bit-compare and time.
//   hipcc --offload-arch=gfx950 -O2 -Iinclude scratch/stem_probe.cpp -o scratch/stem_probe \
//         -Lvideo-long-term-feature-banks_amd/lib/vlfb -lvlfb_hip -Wl,-rpath,'$ORIGIN/../video-long-term-feature-banks_amd/lib/vlfb'
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "vlfb.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
__global__ void fill_bf16(unsigned short* p, long long n, unsigned seed, float scale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    float f = ((float)(h & 0xffffff) / 8388608.0f - 1.0f) * scale;
    unsigned u = __float_as_uint(f); u += 0x7fffu + ((u >> 16) & 1u);
    p[i] = (unsigned short)(u >> 16);
  }
}
__global__ void fill_f32(float* p, long long n, unsigned seed) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    p[i] = (float)((i * 37 + seed) % 17) * 0.05f - 0.4f;
}
__global__ void count_diff(const unsigned* a, const unsigned* b, long long n, unsigned long long* out) {
  unsigned long long c = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) c += a[i] != b[i];
  if (c) atomicAdd(out, c);
}
int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 8, T = argc > 2 ? atoi(argv[2]) : 32, H = argc > 3 ? atoi(argv[3]) : 224;
  const int W = 224, wpad = 4, Wp = W + 2 * wpad, Ho = (H + 6 - 7) / 2 + 1, Wo = 112;
  vlfb_conv_desc d; vlfb_conv_desc_init(&d);
  d.mode = VLFB_CONV_FPROP; d.dtype = VLFB_BF16; d.out_dtype = VLFB_BF16;
  d.N = N; d.Tr = T; d.Hr = Ho; d.Wr = Wo; d.Ts = T; d.Hs = H; d.Ws = Wp; d.Cs = 4; d.Cn = 64;
  d.kt = 5; d.kh = 7; d.kw = 7; d.st = 1; d.sh = 2; d.sw = 2; d.pt = 2; d.ph = 3; d.pw = 3 - wpad; d.dt = d.dh = d.dw = 1;
  d.pack_w = 8; d.relu = 1; d.bias_mode = VLFB_BIAS_COL;
  const long long nin = (long long)N * T * H * Wp * 4, K = 5 * 7 * 32, M = (long long)N * T * Ho * Wo;
  unsigned short *X, *Wt; float* bias; char *O1, *O2; unsigned long long* dcnt;
  CK(hipMalloc(&X, nin * 2)); CK(hipMalloc(&Wt, 64 * K * 2)); CK(hipMalloc(&bias, 256)); CK(hipMalloc(&O1, M * 128)); CK(hipMalloc(&O2, M * 128));
  CK(hipMalloc(&dcnt, 8));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, s, X, nin, 11u, 1.0f);
  hipLaunchKernelGGL(fill_bf16, dim3(256), dim3(256), 0, s, Wt, 64 * K, 22u, 0.05f);
  hipLaunchKernelGGL(fill_f32, dim3(1), dim3(64), 0, s, bias, 64ll, 3u);
  CK(hipMemsetAsync(O1, 0xff, M * 128, s)); CK(hipMemsetAsync(O2, 0xee, M * 128, s));
  float t[2];
  for (int algo = 0; algo < 2; ++algo) {
    d.algo = algo == 0 ? VLFB_ALGO_TILE128 : VLFB_ALGO_AUTO;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 6; ++rep) {
      if (rep == 1) CK(hipEventRecord(e0, s));
      int rc = vlfb_conv_run(&d, X, Wt, nullptr, algo == 0 ? O1 : O2, bias, nullptr, nullptr, nullptr, nullptr, 0, s);
      if (rc) { printf("error: %s\n", vlfb_last_error()); return 1; }
    }
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&t[algo], e0, e1)); t[algo] *= 1e3f / 5;
  }
  CK(hipMemsetAsync(dcnt, 0, 8, s));
  hipLaunchKernelGGL(count_diff, dim3(2048), dim3(256), 0, s, (const unsigned*)O1, (const unsigned*)O2, M * 32, dcnt);
  unsigned long long h = 0; CK(hipMemcpyAsync(&h, dcnt, 8, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
  const double gf = 2.0 * M * 64 * 5 * 7 * 7 * 3 / 1e9;
  printf("stem fprop N=%d T=%d H=%d: tiled %.1f us (%.0f TF/s) | direct %.1f us (%.0f TF/s) | %s (%llu words differ)\n", N, T, H, t[0],
         gf / t[0] * 1e3, t[1], gf / t[1] * 1e3, h ? "MISMATCH" : "bit-exact", h);
  return h ? 1 : 0;
}
