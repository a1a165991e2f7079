// Clip preprocessing on the device: decoded uint8 BGR frames ->
// short-side scale (bilinear) -> crop -> horizontal flip -> /255 -> (x - mean) / std -> RGB, written
// straight into the model's `data` input in its device layout [T][crop][wl + crop + wr][c_pad].
// Replaces the per-frame cv2 / NumPy chain of lib/datasets/data_input_helper.py:70-139
// (images_and_boxes_preprocessing) and lib/datasets/image_processor.py:80-251.
//
// The resize is OpenCV's 8-bit INTER_LINEAR fixed-point algorithm (cfg.INTERPOLATION, config.py:238):
// the host computes the per-column / per-row source indices and 11-bit coefficients exactly as
// cv::resize does and the kernel does integer arithmetic only, so the result does not depend on
// floating-point contraction or rounding modes.  Built with -ffp-contract=off for the fp32
// normalisation tail (same operation order as the NumPy code: x / 255, - mean, / std).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vlfb.h"
#include "vlfb_common.h"

namespace vlfb {
namespace {

struct ClipP {
  const uint8_t* src;     // [T][Hs][Ws][3]
  const int32_t* xofs;    // [Wr] left source column      (resized image column -> source)
  const int16_t* xcoef;   // [Wr][2] 11-bit weights of columns xofs, xofs + 1
  const int32_t* yofs;    // [Hr]
  const int16_t* ycoef;   // [Hr][2]
  int T, Hs, Ws, Hr, Wr;
  int resize;             // 0: Hr == Hs and Wr == Ws, frames are used as they are
  int crop_h, crop_w, y0, x0, flip;
  float mean[3], stdv[3]; // in the SOURCE channel order (BGR)
  int to_rgb;
  int wl, wtot, c_pad;    // destination row: wl zero pixels, crop_w pixels, rest zero; c_pad channels
};

__device__ __forceinline__ int resized_u8(const ClipP& p, const uint8_t* frame, int y, int x, int c) {
  if (!p.resize) return frame[((long long)y * p.Ws + x) * 3 + c];
  const int sx = p.xofs[x], sy = p.yofs[y];
  const int a0 = p.xcoef[2 * x], a1 = p.xcoef[2 * x + 1];
  const int b0 = p.ycoef[2 * y], b1 = p.ycoef[2 * y + 1];
  const int sx1 = min(sx + 1, p.Ws - 1), sy1 = min(sy + 1, p.Hs - 1);
  const uint8_t* r0 = frame + (long long)sy * p.Ws * 3;
  const uint8_t* r1 = frame + (long long)sy1 * p.Ws * 3;
  // horizontal pass (HResizeLinear: int = u8 * coef + u8 * coef), vertical pass with the staged
  // shifts of VResizeLinear<uchar>: ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
  const int h0 = r0[sx * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;
  const int h1 = r1[sx * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
  const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

template <typename T>
__global__ void clip_preprocess_kernel(ClipP p, T* __restrict__ dst) {
  const long long total = (long long)p.T * p.crop_h * p.crop_w;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % p.crop_w);
    const long long ty = i / p.crop_w;
    const int y = (int)(ty % p.crop_h);
    const int t = (int)(ty / p.crop_h);
    const int xs = p.flip ? p.x0 - x : p.x0 + x;   // x0 = resized-frame column of output column 0; a flip walks left
    const int ys = p.y0 + y;
    const uint8_t* frame = p.src + (long long)t * p.Hs * p.Ws * 3;
    T* d = dst + ((long long)(t * p.crop_h + y) * p.wtot + p.wl + x) * p.c_pad;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float u = (float)resized_u8(p, frame, ys, xs, c);
      float v = u / 255.0f;
      v = v - p.mean[c];
      v = v / p.stdv[c];
      Elem<T>::st(d + (p.to_rgb ? 2 - c : c), v);
    }
  }
}

}  // namespace
}  // namespace vlfb

using namespace vlfb;

extern "C" int vlfb_clip_preprocess(const vlfb_clip_desc* d, const uint8_t* frames, const int32_t* xofs,
                                    const int16_t* xcoef, const int32_t* yofs, const int16_t* ycoef,
                                    void* dst, int dst_dtype, vlfb_stream_t stream) {
  VLFB_REQUIRE(d && frames && dst, "clip_preprocess: NULL buffer");
  VLFB_REQUIRE(d->frames > 0 && d->src_h > 0 && d->src_w > 0 && d->crop_h > 0 && d->crop_w > 0, "clip_preprocess: empty geometry");
  const bool resize = d->resized_h != d->src_h || d->resized_w != d->src_w;
  VLFB_REQUIRE(!resize || (xofs && xcoef && yofs && ycoef), "clip_preprocess: resize tables are required");
  VLFB_REQUIRE(d->y0 >= 0 && d->y0 + d->crop_h <= d->resized_h && d->x0 >= 0 && d->x0 < d->resized_w &&
                   (d->flip ? d->x0 - (d->crop_w - 1) >= 0 : d->x0 + d->crop_w <= d->resized_w),
               "clip_preprocess: crop window leaves the resized frame");
  VLFB_REQUIRE(d->c_pad >= 3 && d->w_left >= 0 && d->w_total >= d->w_left + d->crop_w, "clip_preprocess: bad destination row");
  VLFB_REQUIRE(dst_dtype == VLFB_F32 || is16(dst_dtype), "clip_preprocess: dst dtype must be f32 or bf16");
  ClipP p;
  p.src = frames; p.xofs = xofs; p.xcoef = xcoef; p.yofs = yofs; p.ycoef = ycoef;
  p.T = d->frames; p.Hs = d->src_h; p.Ws = d->src_w; p.Hr = d->resized_h; p.Wr = d->resized_w;
  p.resize = resize ? 1 : 0;
  p.crop_h = d->crop_h; p.crop_w = d->crop_w; p.y0 = d->y0; p.x0 = d->x0; p.flip = d->flip;
  for (int c = 0; c < 3; ++c) { p.mean[c] = d->mean[c]; p.stdv[c] = d->std[c]; }
  p.to_rgb = d->to_rgb; p.wl = d->w_left; p.wtot = d->w_total; p.c_pad = d->c_pad;
  const long long total = (long long)p.T * p.crop_h * p.crop_w;
  const int grid = grid_for(total, 256);
  if (dst_dtype == VLFB_F32)
    hipLaunchKernelGGL(clip_preprocess_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, (float*)dst);
  else
    VLFB_WITH_T16(dst_dtype, hipLaunchKernelGGL(clip_preprocess_kernel<T16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, (T16*)dst));
  return check_launch("clip_preprocess");
}
