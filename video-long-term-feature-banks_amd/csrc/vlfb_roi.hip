// RoIAlign (legacy Caffe2 semantics, sampling_ratio = 0) fused with the 7x7 max pool that
// follows it in the RoI head (lib/models/head_helper.py:88-123, lib/models/lfb_helper.py:130-152).
//
// Build note: this file is compiled with -ffp-contract=off.  The integer decisions of RoIAlign
// (sampling-grid size, bilinear corner indices, the out-of-range predicate) depend on the exact
// fp32 operation order of the Caffe2 operator; fused multiply-adds would
// change roundings and therefore indices, and the parity bar for those is bit-exact.
#include "vlfb_common.h"
#include <math.h>

namespace vlfb {
namespace {

struct RoiGeom {
  int batch;
  float start_w, start_h, bin_h, bin_w;
  int grid_h, grid_w;
};

__device__ __forceinline__ RoiGeom roi_geom(const float* roi, float spatial_scale, int pooled) {
  RoiGeom g;
  g.batch = (int)roi[0];
  g.start_w = roi[1] * spatial_scale;
  g.start_h = roi[2] * spatial_scale;
  const float end_w = roi[3] * spatial_scale;
  const float end_h = roi[4] * spatial_scale;
  const float roi_w = fmaxf(end_w - g.start_w, 1.0f);
  const float roi_h = fmaxf(end_h - g.start_h, 1.0f);
  g.bin_h = roi_h / (float)pooled;
  g.bin_w = roi_w / (float)pooled;
  g.grid_h = (int)ceilf(roi_h / (float)pooled);
  g.grid_w = (int)ceilf(roi_w / (float)pooled);
  return g;
}

struct Bilin {
  int y_low, x_low, y_high, x_high;
  float w1, w2, w3, w4;
  bool inside;
};

__device__ __forceinline__ Bilin bilinear(float y, float x, int height, int width) {
  Bilin b;
  b.inside = !(y < -1.0f || y > (float)height || x < -1.0f || x > (float)width);
  if (!b.inside) {
    b.y_low = b.x_low = b.y_high = b.x_high = -1;
    b.w1 = b.w2 = b.w3 = b.w4 = 0.f;
    return b;
  }
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  b.y_low = (int)y;
  b.x_low = (int)x;
  if (b.y_low >= height - 1) { b.y_high = b.y_low = height - 1; y = (float)b.y_low; }
  else b.y_high = b.y_low + 1;
  if (b.x_low >= width - 1) { b.x_high = b.x_low = width - 1; x = (float)b.x_low; }
  else b.x_high = b.x_low + 1;
  const float ly = y - (float)b.y_low, lx = x - (float)b.x_low;
  const float hy = 1.0f - ly, hx = 1.0f - lx;
  b.w1 = hy * hx; b.w2 = hy * lx; b.w3 = ly * hx; b.w4 = ly * lx;
  return b;
}

// one block per RoI; threads = 16-byte channel chunks x groups of pooled rows
template <typename T>
__global__ void roi_align_max_fwd_kernel(const T* __restrict__ feat, const float* __restrict__ rois,
                                         T* __restrict__ out, uint8_t* __restrict__ argbin,
                                         int32_t* __restrict__ dbg, int H, int W, int C, int pooled,
                                         float spatial_scale) {
  constexpr int V = Vec16<T>::N;
  const int r = blockIdx.x;
  const RoiGeom g = roi_geom(rois + (long long)r * 5, spatial_scale, pooled);
  const float count = (float)(g.grid_h * g.grid_w);
  const T* fb = feat + (long long)g.batch * H * W * C;

  if (dbg && threadIdx.x == 0) {
    for (int ph = 0; ph < pooled; ++ph)
      for (int pw = 0; pw < pooled; ++pw) {
        const float y = g.start_h + (float)ph * g.bin_h + (0.5f * g.bin_h) / (float)g.grid_h;
        const float x = g.start_w + (float)pw * g.bin_w + (0.5f * g.bin_w) / (float)g.grid_w;
        const Bilin b = bilinear(y, x, H, W);
        int32_t* d = dbg + (((long long)r * pooled + ph) * pooled + pw) * 8;
        d[0] = g.batch; d[1] = g.grid_h; d[2] = g.grid_w;
        d[3] = b.y_low; d[4] = b.x_low; d[5] = b.y_high; d[6] = b.x_high; d[7] = b.inside ? 1 : 0;
      }
  }

  // threads = channel chunks x GR row groups: group q scans pooled rows [q*rows_per, ...) and the groups
  // are folded through LDS in bin order with a strict >, so the first maximal bin wins as in a serial scan
  extern __shared__ char roi_sm[];
  const int nch = C / V;                               // 16-byte channel chunks (host: C % V == 0)
  const int GR = blockDim.x / nch > 0 ? blockDim.x / nch : 1;
  const int chunk = threadIdx.x % nch, q = threadIdx.x / nch;
  const int rows_per = (pooled + GR - 1) / GR;
  float* sbest = reinterpret_cast<float*>(roi_sm);     // [GR][C]
  uint8_t* sarg = reinterpret_cast<uint8_t*>(sbest + (long long)GR * C);
  if (q < GR && chunk < nch) {
    const int c0 = chunk * V;
    float best[V];
    int arg[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { best[k] = -INFINITY; arg[k] = 0; }
    const int ph_end = min(pooled, (q + 1) * rows_per);
    for (int ph = q * rows_per; ph < ph_end; ++ph)
      for (int pw = 0; pw < pooled; ++pw) {
        float acc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] = 0.f;
        for (int iy = 0; iy < g.grid_h; ++iy) {
          const float y = g.start_h + (float)ph * g.bin_h + (((float)iy + 0.5f) * g.bin_h) / (float)g.grid_h;
          for (int ix = 0; ix < g.grid_w; ++ix) {
            const float x = g.start_w + (float)pw * g.bin_w + (((float)ix + 0.5f) * g.bin_w) / (float)g.grid_w;
            const Bilin b = bilinear(y, x, H, W);
            if (!b.inside) continue;
            float v1[V], v2[V], v3[V], v4[V];
            Vec16<T>::load(fb + ((long long)b.y_low * W + b.x_low) * C + c0, v1);
            Vec16<T>::load(fb + ((long long)b.y_low * W + b.x_high) * C + c0, v2);
            Vec16<T>::load(fb + ((long long)b.y_high * W + b.x_low) * C + c0, v3);
            Vec16<T>::load(fb + ((long long)b.y_high * W + b.x_high) * C + c0, v4);
#pragma unroll
            for (int k = 0; k < V; ++k) acc[k] += b.w1 * v1[k] + b.w2 * v2[k] + b.w3 * v3[k] + b.w4 * v4[k];
          }
        }
        const int bin = ph * pooled + pw;
#pragma unroll
        for (int k = 0; k < V; ++k) {
          const float v = acc[k] / count;
          if (v > best[k]) { best[k] = v; arg[k] = bin; }
        }
      }
#pragma unroll
    for (int k = 0; k < V; ++k) {
      sbest[(long long)q * C + c0 + k] = best[k];
      sarg[(long long)q * C + c0 + k] = (uint8_t)arg[k];
    }
  }
  __syncthreads();
  if (q == 0 && chunk < nch) {
    const int c0 = chunk * V;
    float best[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
      float bv = sbest[c0 + k];
      int ba = sarg[c0 + k];
      for (int j = 1; j < GR; ++j) {
        const float v = sbest[(long long)j * C + c0 + k];
        if (v > bv) { bv = v; ba = sarg[(long long)j * C + c0 + k]; }
      }
      best[k] = bv;
      argbin[(long long)r * C + c0 + k] = (uint8_t)ba;
    }
    Vec16<T>::store(out + (long long)r * C + c0, best);
  }
}

// Backward of RoIAlign + max: the gradient of (roi, channel) goes through the selected bin's bilinear samples to at most
// 4 x grid_h x grid_w pixels of the RoI's clip.  DETERMINISTIC: one thread owns a (clip, channel) column of dfeat and adds
// the contributions of that clip's RoIs in RoI order, sample order, corner order with plain read-modify-writes -- no two
// threads ever touch the same address (the reference operator scatters with atomics, whose order varies from run to run:
// overlapping RoIs of a clip hit the same pixels).  R is a few tens, so a thread scanning all RoIs costs nothing.
template <typename T>
__global__ void roi_align_max_bwd_kernel(const T* __restrict__ dout, const float* __restrict__ rois,
                                         const uint8_t* __restrict__ argbin, float* __restrict__ dfeat,
                                         long long N, long long R, int H, int W, int C, int pooled,
                                         float spatial_scale) {
  const long long total = N * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / C;
    const int c = (int)(i - n * C);
    float* fb = dfeat + n * H * W * C + c;
    for (long long r = 0; r < R; ++r) {
      const RoiGeom g = roi_geom(rois + r * 5, spatial_scale, pooled);
      if ((long long)g.batch != n) continue;
      const float count = (float)(g.grid_h * g.grid_w);
      const int bin = argbin[r * C + c];
      const int ph = bin / pooled, pw = bin - ph * pooled;
      const float gv = Elem<T>::ld(dout + r * C + c) / count;
      for (int iy = 0; iy < g.grid_h; ++iy) {
        const float y = g.start_h + (float)ph * g.bin_h + (((float)iy + 0.5f) * g.bin_h) / (float)g.grid_h;
        for (int ix = 0; ix < g.grid_w; ++ix) {
          const float x = g.start_w + (float)pw * g.bin_w + (((float)ix + 0.5f) * g.bin_w) / (float)g.grid_w;
          const Bilin b = bilinear(y, x, H, W);
          if (!b.inside) continue;
          float* p1 = fb + ((long long)b.y_low * W + b.x_low) * C;
          float* p2 = fb + ((long long)b.y_low * W + b.x_high) * C;
          float* p3 = fb + ((long long)b.y_high * W + b.x_low) * C;
          float* p4 = fb + ((long long)b.y_high * W + b.x_high) * C;
          *p1 += gv * b.w1;          // (the four corners may coincide at the border: sequential adds, in this order)
          *p2 += gv * b.w2;
          *p3 += gv * b.w3;
          *p4 += gv * b.w4;
        }
      }
    }
  }
}

}  // namespace
}  // namespace vlfb

using namespace vlfb;

extern "C" int vlfb_roi_align_max_fwd(const void* feat, int dtype, const float* rois, void* out,
                                      uint8_t* argbin, int32_t* dbg, int64_t n, int64_t h, int64_t w,
                                      int64_t c, int64_t r, int pooled, float spatial_scale,
                                      vlfb_stream_t stream) {
  VLFB_REQUIRE(feat && rois && out && argbin && n > 0 && h > 0 && w > 0 && c > 0 && r > 0, "roi_align_fwd: bad args");
  VLFB_REQUIRE(pooled > 0 && pooled * pooled <= 255, "roi_align_fwd: pooled resolution out of range");
  const int v = dtype == VLFB_F32 ? 4 : 8;
  VLFB_REQUIRE(c % v == 0, "roi_align_fwd: C must be a multiple of %d", v);
  const int nch = (int)(c / v);
  VLFB_REQUIRE(nch <= 1024, "roi_align_fwd: at most %d channels", 1024 * v);
  int gr = 1024 / nch;                      // row groups per RoI (few RoIs: parallelism has to come from the bins)
  if (gr > pooled) gr = pooled;
  if (gr < 1) gr = 1;
  const int threads = nch * gr;
  const size_t lds = (size_t)gr * c * 5;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == VLFB_F32)
    hipLaunchKernelGGL(roi_align_max_fwd_kernel<float>, dim3((unsigned)r), dim3(threads), lds, s, (const float*)feat, rois, (float*)out, argbin, dbg, (int)h, (int)w, (int)c, pooled, spatial_scale);
  else if (is16(dtype))
    VLFB_WITH_T16(dtype, hipLaunchKernelGGL(roi_align_max_fwd_kernel<T16>, dim3((unsigned)r), dim3(threads), lds, s, (const T16*)feat, rois, (T16*)out, argbin, dbg, (int)h, (int)w, (int)c, pooled, spatial_scale));
  else return set_error(VLFB_ERR_ARG, "roi_align_fwd: bad dtype");
  return check_launch("roi_align_max_fwd");
}

// Integer decisions of EVERY bilinear sample (test hook, never on the hot path): the same roi_geom / bilinear device
// functions and the same coordinate expressions as the loops above, one thread per (RoI, bin).
namespace vlfb { namespace {
__global__ void roi_decisions_kernel(const float* __restrict__ rois, int32_t* __restrict__ dbg, int R, int H, int W,
                                     int pooled, float spatial_scale, int max_grid) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= R * pooled * pooled) return;
  const int r = id / (pooled * pooled), ph = (id / pooled) % pooled, pw = id % pooled;
  const RoiGeom g = roi_geom(rois + (long long)r * 5, spatial_scale, pooled);
  int32_t* d0 = dbg + (long long)id * max_grid * max_grid * 8;
  for (int iy = 0; iy < max_grid; ++iy)
    for (int ix = 0; ix < max_grid; ++ix) {
      int32_t* d = d0 + (iy * max_grid + ix) * 8;
      if (iy >= g.grid_h || ix >= g.grid_w) {
        d[0] = g.batch; d[1] = g.grid_h; d[2] = g.grid_w; d[3] = d[4] = d[5] = d[6] = -2; d[7] = -1;   // no such sample
        continue;
      }
      const float y = g.start_h + (float)ph * g.bin_h + (((float)iy + 0.5f) * g.bin_h) / (float)g.grid_h;
      const float x = g.start_w + (float)pw * g.bin_w + (((float)ix + 0.5f) * g.bin_w) / (float)g.grid_w;
      const Bilin b = bilinear(y, x, H, W);
      d[0] = g.batch; d[1] = g.grid_h; d[2] = g.grid_w;
      d[3] = b.y_low; d[4] = b.x_low; d[5] = b.y_high; d[6] = b.x_high; d[7] = b.inside ? 1 : 0;
    }
}
} }

extern "C" int vlfb_roi_align_decisions(const float* rois, int32_t* dbg, int64_t h, int64_t w, int64_t r, int pooled,
                                        float spatial_scale, int max_grid, vlfb_stream_t stream) {
  VLFB_REQUIRE(rois && dbg && r > 0 && pooled > 0 && max_grid > 0, "roi_align_decisions: bad args");
  const int n = (int)r * pooled * pooled;
  hipLaunchKernelGGL(vlfb::roi_decisions_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, rois, dbg, (int)r,
                     (int)h, (int)w, pooled, spatial_scale, max_grid);
  return vlfb::check_launch("roi_align_decisions");
}

extern "C" int vlfb_roi_align_max_bwd(const void* dout, int dtype, const float* rois,
                                      const uint8_t* argbin, float* dfeat, int64_t n, int64_t h,
                                      int64_t w, int64_t c, int64_t r, int pooled, float spatial_scale,
                                      vlfb_stream_t stream) {
  VLFB_REQUIRE(dout && rois && argbin && dfeat && n > 0 && h > 0 && w > 0 && c > 0 && r > 0, "roi_align_bwd: bad args");
  int grid = grid_for(n * c, 64);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == VLFB_F32)
    hipLaunchKernelGGL(roi_align_max_bwd_kernel<float>, dim3(grid), dim3(64), 0, s, (const float*)dout, rois, argbin, dfeat, (long long)n, (long long)r, (int)h, (int)w, (int)c, pooled, spatial_scale);
  else if (is16(dtype))
    VLFB_WITH_T16(dtype, hipLaunchKernelGGL(roi_align_max_bwd_kernel<T16>, dim3(grid), dim3(64), 0, s, (const T16*)dout, rois, argbin, dfeat, (long long)n, (long long)r, (int)h, (int)w, (int)c, pooled, spatial_scale));
  else return set_error(VLFB_ERR_ARG, "roi_align_bwd: bad dtype");
  return check_launch("roi_align_max_bwd");
}
