// Development probe (not product, not a test): runs the HBM-bound layer shapes of the 8-clip ava_r50_lfb_nl
// step through vlfb_conv_run with algo = TILE128 and algo = STREAM (the weight-resident streaming kernel),
// checks that both give bit-identical outputs and prints their times and the algorithmic HBM rate.
//   hipcc --offload-arch=gfx950 -O2 -Iinclude scratch/nts_probe.cpp -o scratch/nts_probe \
//         -Lvideo-long-term-feature-banks_amd/lib/vlfb -lvlfb_hip -Wl,-rpath,'$ORIGIN/../video-long-term-feature-banks_amd/lib/vlfb'
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "vlfb.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void fill_bf16(unsigned short* p, long long n, unsigned seed, float scale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    float f = ((float)(h & 0xffffff) / 8388608.0f - 1.0f) * scale;
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    p[i] = (unsigned short)(u >> 16);
  }
}
__global__ void fill_f32(float* p, long long n, unsigned seed, float scale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    p[i] = ((float)(h & 0xffffff) / 8388608.0f - 1.0f) * scale;
  }
}
__global__ void count_diff(const unsigned* a, const unsigned* b, long long n, unsigned long long* out) {
  unsigned long long c = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    c += a[i] != b[i];
  if (c) atomicAdd(out, c);
}

__global__ void count_far(const float* a, const float* b, long long n, unsigned long long* out) {
  unsigned long long c = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float d = fabsf(a[i] - b[i]);
    c += !(d <= 1e-3f * fmaxf(fabsf(a[i]), fabsf(b[i])) + 2e-2f);
  }
  if (c) atomicAdd(out, c);
}

struct Case {
  const char* name;
  int mode;            // 0 fprop, 1 dgrad
  int N, Tr, Hr, Wr;   // row space
  int Ts, Hs, Ws, Cs;  // source
  int Cn;
  int kt, kh, kw, st, sh, sw, pt, ph, pw, dh;
  int batch;           // > 1: plain batched GEMM
  int flags;           // 1 R, 2 relu, 4 bias, 8 mask, 16 fp32 out
};

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  const char* only = argc > 2 ? argv[2] : nullptr;
  const int second_algo = argc > 3 ? atoi(argv[3]) : 3;      // 3 = STREAM, 0 = AUTO (whatever the library picks)
  std::vector<Case> cases = {
    // res2 (8 clips: 32 x 56 x 56 per clip)
    {"res2 2c 64->256 +R relu", 0, 8, 32, 56, 56, 32, 56, 56, 64, 256, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 7},
    {"res2 b1 64->256", 0, 8, 32, 56, 56, 32, 56, 56, 64, 256, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 4},
    {"res2 2c dgrad 256->64 mask", 1, 8, 32, 56, 56, 32, 56, 56, 256, 64, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 8},
    {"res2 2a 3x1x1 256->64", 0, 8, 32, 56, 56, 32, 56, 56, 256, 64, 3, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, 6},
    {"res2 2a 3x1x1 dgrad +R mask", 1, 8, 32, 56, 56, 32, 56, 56, 64, 256, 3, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, 9},
    {"res2 2b 3x3 64->64", 0, 8, 32, 56, 56, 32, 56, 56, 64, 64, 1, 3, 3, 1, 1, 1, 0, 1, 1, 1, 1, 6},
    {"res2 2b 3x3 dgrad mask", 1, 8, 32, 56, 56, 32, 56, 56, 64, 64, 1, 3, 3, 1, 1, 1, 0, 1, 1, 1, 1, 8},
    {"res2_0 2a 3x1x1 64->64", 0, 8, 32, 56, 56, 32, 56, 56, 64, 64, 3, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, 6},
    {"res2_0 2a 3x1x1 dgrad", 1, 8, 32, 56, 56, 32, 56, 56, 64, 64, 3, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, 0},
    // res3 (16 x 28 x 28)
    {"res3 2c 128->512 +R relu", 0, 8, 16, 28, 28, 16, 28, 28, 128, 512, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 7},
    {"res3 2c dgrad 512->128 mask", 1, 8, 16, 28, 28, 16, 28, 28, 512, 128, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 8},
    {"res3 2a 512->128", 0, 8, 16, 28, 28, 16, 28, 28, 512, 128, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 6},
    {"res3 2a dgrad 128->512 +R mask", 1, 8, 16, 28, 28, 16, 28, 28, 128, 512, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 9},
    {"res3_0 2b 3x3 s2 128->128", 0, 8, 16, 28, 28, 16, 56, 56, 128, 128, 1, 3, 3, 1, 2, 2, 0, 1, 1, 1, 1, 6},
    // res4 (16 x 14 x 14): weights that still fit
    {"res4 2c dgrad 1024->256 (no)", 1, 8, 16, 14, 14, 16, 14, 14, 1024, 256, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 8},
    // stride-2 dgrads (rows = conv input positions)
    {"s2 res3_0 2b 3x3 dgrad mask", 1, 8, 16, 56, 56, 16, 28, 28, 128, 128, 1, 3, 3, 1, 2, 2, 0, 1, 1, 1, 1, 8},
    {"s2 res3_0 b1 dgrad 512->256 +R", 1, 8, 16, 56, 56, 16, 28, 28, 512, 256, 1, 1, 1, 1, 2, 2, 0, 0, 0, 1, 1, 1},
    {"s2 res4_0 2b 3x3 dgrad mask", 1, 8, 16, 28, 28, 16, 14, 14, 256, 256, 1, 3, 3, 1, 2, 2, 0, 1, 1, 1, 1, 8},
    {"s2 res4_0 b1 dgrad 1024->512 +R", 1, 8, 16, 28, 28, 16, 14, 14, 1024, 512, 1, 1, 1, 1, 2, 2, 0, 0, 0, 1, 1, 1},
    {"nl3 g/phi 512->256 pooled", 0, 8, 16, 14, 14, 16, 14, 14, 512, 128, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 4},
  };
  hipStream_t s;
  CK(hipStreamCreate(&s));
  unsigned long long* dcnt;
  CK(hipMalloc(&dcnt, 8));
  printf("%-32s %8s %9s %9s %8s %8s %7s  %s\n", "case", "MB", "t128_us", "tstr_us", "TB/s128", "TB/sStr", "speedup", "check");
  double tot128 = 0, tot256 = 0, totfl = 0;
  for (const Case& c : cases) {
    if (only && !strstr(c.name, only)) continue;
    vlfb_conv_desc d;
    vlfb_conv_desc_init(&d);
    d.mode = c.mode; d.dtype = VLFB_BF16; d.out_dtype = (c.flags & 16) ? VLFB_F32 : VLFB_BF16;
    d.N = c.N; d.Tr = c.Tr; d.Hr = c.Hr; d.Wr = c.Wr; d.Ts = c.Ts; d.Hs = c.Hs; d.Ws = c.Ws; d.Cs = c.Cs; d.Cn = c.Cn;
    d.kt = c.kt; d.kh = c.kh; d.kw = c.kw; d.st = c.st; d.sh = c.sh; d.sw = c.sw; d.pt = c.pt; d.ph = c.ph; d.pw = c.pw;
    d.dt = 1; d.dh = c.dh; d.dw = c.dh;
    d.relu = (c.flags & 2) ? 1 : 0;
    d.bias_mode = (c.flags & 4) ? VLFB_BIAS_COL : VLFB_BIAS_NONE;
    const long long M = (long long)c.N * c.Tr * c.Hr * c.Wr;
    const long long src = (long long)c.N * c.Ts * c.Hs * c.Ws;
    const long long K = (long long)c.kt * c.kh * c.kw * c.Cs;
    const bool wg = c.mode == 2;
    if (c.batch > 1) {
      d.batch = c.batch;
      d.a_bstride = M * c.Cs; d.b_bstride = (long long)c.Cn * K; d.o_bstride = wg ? (long long)c.Cn * K : M * c.Cn;
      d.r_bstride = M * c.Cn; d.p_bstride = M * c.Cn;
      if (wg) d.splits = 1;
    }
    const int B = c.batch > 1 ? c.batch : 1;
    const size_t osz = (c.flags & 16) ? 4 : 2;
    unsigned short *A, *W, *R, *Mk;
    float* bias;
    char *O1, *O2;
    CK(hipMalloc(&A, (size_t)B * src * c.Cs * 2));
    CK(hipMalloc(&W, (size_t)B * c.Cn * K * 2));
    CK(hipMalloc(&R, (size_t)B * M * c.Cn * 2));
    CK(hipMalloc(&Mk, (size_t)B * M * c.Cn * 2));
    CK(hipMalloc(&bias, (size_t)c.Cn * 4));
    const size_t oelems = wg ? (size_t)B * c.Cn * K : (size_t)B * M * c.Cn;
    CK(hipMalloc(&O1, oelems * osz));
    CK(hipMalloc(&O2, oelems * osz));
    float* wsp = nullptr;
    long long wsb = 0;
    hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, s, A, (long long)B * src * c.Cs, 11u, 1.0f);
    hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, s, W, (long long)B * c.Cn * K, 22u, 0.05f);
    hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, s, R, (long long)B * M * c.Cn, 33u, 1.0f);
    hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, s, Mk, (long long)B * M * c.Cn, 44u, 1.0f);
    hipLaunchKernelGGL(fill_f32, dim3(64), dim3(256), 0, s, bias, (long long)c.Cn, 55u, 0.5f);
    CK(hipMemsetAsync(O1, 0xff, oelems * osz, s));
    CK(hipMemsetAsync(O2, 0xee, oelems * osz, s));
    const void* Rp = (c.flags & 1) ? R : nullptr;
    const void* Mp = (c.flags & 8) ? Mk : nullptr;
    const float* bp = (c.flags & 4) ? bias : nullptr;
    float t[3] = {0, 0, 0};
    bool ok256 = true;
    for (int algo = 1; algo <= 2; ++algo) {
      d.algo = algo == 1 ? 1 : second_algo;
      char* O = algo == 1 ? O1 : O2;
      if (wg) {
        const long long need = vlfb_conv_workspace_bytes(&d);
        if (need > wsb) { if (wsp) hipFree(wsp); CK(hipMalloc(&wsp, (size_t)need)); wsb = need; }
      }
      // WGRAD: P = the gradient operand (stored in R's buffer: [M][Cn])
      int rc = wg ? vlfb_conv_run(&d, A, nullptr, R, O, nullptr, nullptr, nullptr, nullptr, wsp, wsb, s)
                  : vlfb_conv_run(&d, A, W, nullptr, O, bp, nullptr, Rp, Mp, nullptr, 0, s);
      if (rc != 0) { printf("%-28s algo %d: %s\n", c.name, algo, vlfb_last_error()); ok256 = false; continue; }
      CK(hipStreamSynchronize(s));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, s));
      for (int r = 0; r < reps; ++r) {
        if (wg) vlfb_conv_run(&d, A, nullptr, R, O, nullptr, nullptr, nullptr, nullptr, wsp, wsb, s);
        else vlfb_conv_run(&d, A, W, nullptr, O, bp, nullptr, Rp, Mp, nullptr, 0, s);
      }
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&t[algo], e0, e1));
      t[algo] = t[algo] * 1e3f / reps;
    }
    unsigned long long cnt = 0;
    CK(hipMemsetAsync(dcnt, 0, 8, s));
    const bool approx = wg && c.batch <= 1;     // split counts differ between the kernels: fp32 sums in another order
    if (approx) hipLaunchKernelGGL(count_far, dim3(1024), dim3(256), 0, s, (const float*)O1, (const float*)O2, (long long)oelems, dcnt);
    else hipLaunchKernelGGL(count_diff, dim3(1024), dim3(256), 0, s, (const unsigned*)O1, (const unsigned*)O2,
                       (long long)(oelems * osz / 4), dcnt);
    CK(hipMemcpyAsync(&cnt, dcnt, 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    const double fl = 2.0 * M * c.Cn * K * B;
    if (ok256) { tot128 += t[1]; tot256 += t[2]; totfl += fl; }
    const double by = 2.0 * ((double)src * c.Cs + (double)M * c.Cn * (1 + ((c.flags & 1) ? 1 : 0) + ((c.flags & 8) ? 1 : 0)));
    printf("%-32s %8.1f %9.1f %9.1f %8.2f %8.2f %7.2f  %s\n", c.name, by * 1e-6, t[1], t[2], by / (t[1] * 1e-6) * 1e-12,
           by / (t[2] * 1e-6) * 1e-12, t[1] / t[2], !ok256 ? "SKIPPED" : cnt == 0 ? (approx ? "close" : "bit-exact") : "MISMATCH");
    if (cnt) printf("   mismatching 32-bit words: %llu of %lld\n", cnt, (long long)(oelems * osz / 4));
    if (wsp) hipFree(wsp);
    fflush(stdout);
    hipFree(A); hipFree(W); hipFree(R); hipFree(Mk); hipFree(bias); hipFree(O1); hipFree(O2);
  }
  printf("TOTAL: %.1f GFLOP  128-tile %.1f us (%.0f TF/s)   streaming %.1f us (%.0f TF/s)\n", totfl * 1e-9, tot128,
         totfl / tot128 * 1e-6, tot256, totfl / tot256 * 1e-6);
  return 0;
}
