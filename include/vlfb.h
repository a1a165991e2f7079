/*
 * vlfb.h -- C ABI of libvlfb_hip.so, the MI355X (gfx950) kernel library behind the
 * R50/R101-I3D-NL + long-term-feature-bank training path.
 *
 * The reference (facebookresearch/video-long-term-feature-banks) reaches its GPU code
 * through the Caffe2 operator registry; Caffe2 is gone, so this header IS the plug-in
 * boundary.  Every entry point cites the reference interface it replaces.
 *
 * Conventions
 *   - plain pointers + sizes only; no torch / HIP types in signatures (vlfb_stream_t is a
 *     hipStream_t passed as void*).
 *   - the caller owns every buffer; kernels are asynchronous on `stream`; the library
 *     allocates nothing and keeps no mutable global state besides a thread-local error string.
 *   - every function returns VLFB_OK (0) or a negative error code; vlfb_last_error() gives text.
 *   - "NCTHW" = the reference's blob layout (kwargs['order']='NCHW',
 *     lib/models/model_builder_video.py:69).  "NTHWC" = this library's internal
 *     channels-last activation layout (rows = N*T*H*W positions, C contiguous).
 *   - element types: VLFB_BF16 / VLFB_F16 (throughput paths: 16-bit storage, bf16 / fp16 MFMA with fp32
 *     accumulation) and VLFB_F32 -- with vlfb_conv_desc.math = 0 the exact-fp32 MFMA, with math = BF16X6 /
 *     BF16X3 the parity-grade path: fp32 storage, every product as split-bf16 products on the bf16 matrix cores.
 */
#ifndef VLFB_H_
#define VLFB_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vlfb_stream_t;

enum { VLFB_F32 = 0, VLFB_BF16 = 1, VLFB_F16 = 2 };
/* Weight-operand format of the split-bf16 path (vlfb_conv_desc.math != 0; accepted by vlfb_weight_prep* only): the
 * FPROP copy is THREE bf16 planes [3][Cout][taps][Cin] (h = bf16(w), m = bf16(w - h), l = bf16(w - h - m)), the DGRAD
 * copy TWO planes [2][Cin][taps][Cout] (h, m). */
enum { VLFB_SPLIT = 3 };
/* Weight-operand formats of the "mix" path (split-bf16 FORWARD products on fp32 storage, fp16 BACKWARD; accepted by
 * vlfb_weight_prep* only): the FPROP copy as VLFB_SPLIT's three bf16 planes; the DGRAD copy
 *   VLFB_MIX     plain fp16 [Cin][taps][Cout]
 *   VLFB_MIX_W2  two fp16 terms of (w * s) * VLFB_MIX_W2_SCALE, [Cin][term][taps][Cout]: the weight of the same
 *                convolution with a doubled OUTERMOST tap dimension of dilation 0 (vlfb_conv_desc: kt' = 2 kt... with
 *                dt = 0; alpha carries 1 / VLFB_MIX_W2_SCALE), so that the plain fp16 DGRAD kernels contract the fp16
 *                gradient with 22 significant bits of W: dX = dY . Wh + dY . Wl */
enum { VLFB_MIX = 4, VLFB_MIX_W2 = 5 };
#define VLFB_MIX_W2_SCALE 1024.0f
/* ... of the two-plane fp16 forward (vlfb_conv_desc.math = VLFB_MATH_F16X3): the FPROP copy as the two fp16 terms of
 * (w * s) * VLFB_MIX_W2_SCALE, planes [term][Cout][taps][Cin] (alpha of the launch carries 1 / VLFB_MIX_W2_SCALE); the
 * DGRAD copy as VLFB_MIX (VLFB_MIXH) or VLFB_MIX_W2 (VLFB_MIXH_W2) */
enum { VLFB_MIXH = 6, VLFB_MIXH_W2 = 7 };
/* ... with the two-term DGRAD copy INTERLEAVED per 64-channel k-tile: [Cin][taps][Cout / 64][term][64] (Cout % 64 == 0),
 * the weight operand of vlfb_conv_desc.math = VLFB_MATH_F16W2 -- the two terms of a tap's 64 channels are consecutive
 * k-tiles, contracted with ONE gradient tile (fetched into LDS and read from it once).  FPROP copy as VLFB_MIX (three bf16
 * planes; VLFB_MIX_W2I) or as VLFB_MIXH (two fp16 planes; VLFB_MIXH_W2I). */
enum { VLFB_MIX_W2I = 9, VLFB_MIXH_W2I = 10 };
/* A TWO-PLANE fp16 tensor: [2][numel] fp16, value = plane 0 (hi = fp16(v)) + plane 1 (lo = fp16(v - hi)), ~22 significant
 * bits.  The storage format of the forward activations of the "mix" path: plane 0 alone is the fp16 tensor the fp16 backward
 * reads (WGRAD operand, ReLU mask).  vlfb_pool_desc.dtype of vlfb_maxpool_fwd (x, y two-plane) and vlfb_avgpool_fwd (x
 * two-plane, y fp32); produced by convolutions through O / O_lo of vlfb_conv_args, by vlfb_pair_split from fp32. */
enum { VLFB_F16PAIR = 8 };
/* vlfb_conv_desc.math: how the contraction is evaluated when dtype == VLFB_F32.
 *   VLFB_MATH_NATIVE  v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 vector rate)
 *   VLFB_MATH_BF16X6  operands expanded into three bf16 terms each, six bf16 MFMAs per product (hh hm mh mm hl lh):
 *                     fp32-grade results at 1/6 of the bf16 matrix rate (forward products of the parity path)
 *   VLFB_MATH_BF16X3  two terms each, three MFMAs (hh hm mh): ~2^-17 per product (backward products)
 * With math != 0 the B operand of FPROP / DGRAD launches is pre-split: bf16 planes (3 for BF16X6, 2 for BF16X3)
 * `b_pstride` elements apart, each laid out as the fp32 B operand would be (vlfb_weight_prep* with VLFB_SPLIT,
 * vlfb_split_planes); A, P, R, Mask and O stay fp32. */
enum { VLFB_MATH_NATIVE = 0, VLFB_MATH_BF16X3 = 3, VLFB_MATH_BF16X6 = 6,
       /* dtype VLFB_F16, FPROP / plain NT products: BOTH operands are two fp16 planes (A: a_pstride elements apart, B:
        * b_pstride, 0 = Cn * ldb), a product is hi.hi + hi.lo + lo.hi on the fp16 MFMA (~2^-21 per product, nothing is
        * converted in the k-loop; fp16 MFMA operands keep subnormals).  out_dtype VLFB_F16: the output is written as two
        * planes O / O_lo, the residual read as R / R_lo (vlfb_conv_args); out_dtype VLFB_F32: a plain fp32 output.
        * The split-bf16 maths accept out_dtype VLFB_F16 in the same sense (fp32 A operand, two-plane output / residual). */
       VLFB_MATH_F16X3 = 13,
       /* dtype VLFB_F16 / VLFB_BF16, unit-stride DGRAD with Cs % 64 == 0: the weight operand holds TWO 16-bit terms per
        * value, rows [tap][Cs / 64][term][64] (vlfb_weight_prep* VLFB_MIX_W2I; ldb = 2 K), dX = dY . (Wh + Wl) with every
        * gradient tile contracted with both terms (alpha carries 1 / VLFB_MIX_W2_SCALE).  The same product as the
        * doubled-tap form of VLFB_MIX_W2 on 3/4 of its LDS DMA bytes and 2/3 of its LDS fragment reads. */
       VLFB_MATH_F16W2 = 12 };

enum {
  VLFB_OK = 0,
  VLFB_ERR_ARG = -1,
  VLFB_ERR_LAUNCH = -2,
  VLFB_ERR_UNSUPPORTED = -3,
  VLFB_ERR_WORKSPACE = -4
};

const char* vlfb_last_error(void);
int vlfb_version(void);
/* number of bytes per element of a VLFB_* dtype (0 if unknown) */
int vlfb_dtype_size(int dtype);

/* ------------------------------------------------------------------------------------------
 * AffineNd / AffineNdGradient -- the reference's only native op, on its own layout.
 * Replaces REGISTER_CUDA_OPERATOR(AffineNd, ...) / (AffineNdGradient, ...)
 *   caffe2_customized_ops/video/affine_nd_op.cu:31-44,61-83  (y = x*s[c] + b[c])
 *   caffe2_customized_ops/video/affine_nd_op.cu:46-58,85-104 (dx = dy*s[c]; no ds/db)
 * x,y: fp32 NCTHW with `inner` = T*H*W elements per (n,c); in-place allowed (x==y, dy==dx)
 * as in the schema (affine_nd_op.cc:35-43).  numel must be < 2^31 as in the reference.
 * ------------------------------------------------------------------------------------------ */
int vlfb_affine_nd_fwd(const float* x, const float* scale, const float* bias, float* y,
                       int64_t n, int64_t c, int64_t inner, vlfb_stream_t stream);
int vlfb_affine_nd_bwd(const float* dy, const float* scale, float* dx,
                       int64_t n, int64_t c, int64_t inner, vlfb_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM 3-D convolution family (MFMA), channels-last.
 * Replaces every cuDNN `Conv` the builders emit (ModelBuilder.ConvNd call sites:
 * lib/models/model_builder_video.py:181,211; resnet_video.py:169; nonlocal_helper.py:36,58,68,131;
 * lib/models/lfb_helper.py:175,184,194,244,303,323), its auto-generated ConvGradient
 * (dgrad + wgrad), and the cuBLAS BatchMatMul calls (nonlocal_helper.py:94,121) -- a plain or
 * batched GEMM is the 1x1x1 special case.
 *
 * mode FPROP:  O[m][n]  = sum_{tap,c} A[src(m,tap)][c] * B[n][tap][c]
 *              rows m enumerate the OUTPUT positions (N,Tr,Hr,Wr);
 *              src(m,tap) = m*stride - pad + tap*dil, inside (Ts,Hs,Ws) else 0.
 * mode DGRAD:  same contraction, rows m enumerate the conv's INPUT positions (N,Tr,Hr,Wr) and
 *              A is the output-gradient (N,Ts,Hs,Ws,Cs): src = (m + pad - tap*dil)/stride when
 *              divisible and in range else 0; B is the weight in [Cin][tap][Cout] order.
 * mode WGRAD:  O[p][tap][c] = sum_m P[m][p] * A[src(m,tap)][c]
 *              (m = output positions; P = output gradient [M][Cn]; A = the conv input).
 *              If the launch is split along m, fp32 partial slabs go to `workspace` and
 *              vlfb_conv_wgrad_reduce must follow (vlfb_conv_run does it when it is given
 *              a workspace).
 * pack_w:      stem mode (conv1, Cs = 4 padded RGB): the kw taps are packed with the channel
 *              dim into the contiguous K axis, K = kt*kh*(kw_pad*Cs); requires dw == 1 and an
 *              input whose W rows are zero-padded (vlfb_ncthw_to_nthwc_wpad) so that
 *              w*sw - pw + [0, kw_pad) is always inside [0, Ws).
 * epilogue (FPROP/DGRAD): v = alpha*acc + bias + R[m][n]; relu; then v = (Mask[m][n] > 0) ? v : 0
 * epilogue (WGRAD):       v = alpha * rowscale[p] * acc (+ O if accumulate)
 * ------------------------------------------------------------------------------------------ */
enum { VLFB_CONV_FPROP = 0, VLFB_CONV_DGRAD = 1, VLFB_CONV_WGRAD = 2 };
enum { VLFB_BIAS_NONE = 0, VLFB_BIAS_COL = 1, VLFB_BIAS_ROW = 2 };
enum { VLFB_ALGO_AUTO = 0, VLFB_ALGO_TILE128 = 1, VLFB_ALGO_PIPE256 = 2, VLFB_ALGO_STREAM = 3,
       VLFB_ALGO_CLASSES = 4 /* split-math DGRAD of a (1,2,2)-strided conv: force the parity-class walk (AUTO takes it for kh*kw > 1) */,
       /* 16-bit DGRAD of a (1,2,2)-strided 1x1x1 conv (the projection shortcut of res3_0 / res4_0, resnet_helper.py:86-103) as
        * an ACCUMULATE over the rows it touches: only the input positions the conv reads (even h, even w: a quarter of the
        * rows) are computed and written, every other row of O is LEFT AS IT IS.  For a caller that passes R = O holding an
        * earlier contribution to the same gradient (in-place accumulate; R_lo = O_lo likewise), or has zero-filled O. */
       VLFB_ALGO_CLASS0 = 5 };

typedef struct vlfb_conv_desc {
  int32_t mode;
  int32_t dtype;      /* operand element type of A, B(,P), R, Mask */
  int32_t out_dtype;  /* VLFB_F32 or == dtype */
  /* row space (see mode) */
  int32_t N, Tr, Hr, Wr;
  /* gather-source extents and channels (per tap K extent = Cs, or kw_pad*Cs with pack_w) */
  int32_t Ts, Hs, Ws, Cs;
  int32_t kt, kh, kw;
  int32_t st, sh, sw;
  int32_t pt, ph, pw;
  int32_t dt, dh, dw;
  int32_t pack_w;     /* 0, or kw_pad (padded kw, power of two) */
  /* FPROP/DGRAD: number of output columns; WGRAD: channels of P (output rows) */
  int32_t Cn;
  /* leading dimensions in elements (0 = dense default) */
  int32_t lda;        /* A position stride            (default Cs)            */
  int32_t ldb;        /* B row stride                 (default K)             */
  int32_t ldo;        /* O row stride                 (default Cn, WGRAD: K)  */
  int32_t ldr;        /* R / Mask row stride          (default ldo)           */
  int32_t ldp;        /* WGRAD: P position stride     (default Cn)            */
  /* batched GEMM (blockIdx.z); strides in elements */
  int32_t batch;
  int64_t a_bstride, b_bstride, o_bstride, r_bstride, p_bstride;
  /* epilogue */
  float alpha;
  int32_t relu;
  int32_t bias_mode;
  int32_t accumulate; /* WGRAD only: O += ... */
  int32_t splits;     /* WGRAD only: 0 = library chooses */
  int32_t algo;       /* kernel family: 0 = library chooses, 1 = 128x128 tiles (one barrier per k-tile),
                         2 = 256-row phase-pipelined tiles (bf16, MFMA-bound shapes; error if the
                         problem is not eligible), 3 = weight-resident streaming kernel (FPROP / DGRAD of
                         HBM-bound layers whose weight operand fits LDS; error if not eligible).  The
                         tile families (1, 2, 3, the direct-convolution kernels AUTO may pick) give bit-identical
                         results; the one exception under AUTO is the skinny kernel for plain NT products of at
                         most 64 rows (the FBO head on one row per RoI), whose four waves split K: same products,
                         another fp32 summation order (VLFB_SKINNY=0 or algo = 1 restores the tile order). */
  int32_t math;       /* VLFB_MATH_* (dtype VLFB_F32 only) */
  int64_t b_pstride;  /* math != 0, FPROP / DGRAD: elements between the bf16 term planes of B (0 = batch * Cn * ldb) */
  /* math != 0: activations / gradients that some earlier launch already wrote as bf16 term planes next to their fp32
   * values (o_planes below) can be handed in pre-split -- the kernel then spends no VALU on the expansion:
   *   a_planes  0 = A is fp32; 2 (BF16X3) = A points at [plane][position][lda] bf16 planes, a_pstride elements apart
   *   p_planes  WGRAD only: the same for P (ldp); WGRAD takes both operands as planes or neither
   *   o_planes  FPROP / DGRAD: 2 = additionally write the first two bf16 terms of every output value to the
   *             O_planes argument of vlfb_conv_run_planes, [plane][row][ldo], o_pstride elements apart;
   *             1 = additionally write an fp16 COPY of the output to O_planes, [row][ldo] (batch stride o_bstride): what
   *             the fp16 backward of the "mix" path reads (positive values stay positive in the copy) */
  int32_t a_planes, p_planes, o_planes;
  int32_t wgrad_bias; /* WGRAD only: 1 = the launch also produces the bias gradient db[p] = alpha * rowscale[p] * sum_m P[m][p]
                         (vlfb_conv_run_wgrad_bias); counted in vlfb_conv_workspace_bytes */
  int64_t a_pstride, p_pstride, o_pstride;
} vlfb_conv_desc;

/* fills the desc with zeros and safe defaults (1x1x1, stride 1, alpha 1, batch 1) */
void vlfb_conv_desc_init(vlfb_conv_desc* d);
/* bytes of fp32 workspace vlfb_conv_run needs for this desc (0 unless a split WGRAD) */
int64_t vlfb_conv_workspace_bytes(const vlfb_conv_desc* d);

/* Text description of what vlfb_conv_run launches for this descriptor -- kernel family, element type, tile, split count
 * ("nt8 bf16 256x256", "tn_tr f16 128x128 splits=24", ...).  The planner is a pure function of the descriptor.  buf: at
 * least 96 bytes. */
int vlfb_conv_plan_describe(const vlfb_conv_desc* d, char* buf, int64_t buf_bytes);

/* One scratch query for every entry point that takes caller-owned scratch (the caller owns every
 * buffer, the library never allocates).  `arg` is the op's descriptor or dimension list; returns bytes, < 0 on error.
 *   VLFB_WS_CONV           arg = const vlfb_conv_desc*            `workspace` of vlfb_conv_run (= vlfb_conv_workspace_bytes)
 *   VLFB_WS_MAXPOOL_ARGMAX arg = const vlfb_pool_desc*            the whole `argmax` tensor of vlfb_maxpool_fwd / _bwd
 *   VLFB_WS_FBO_ATTN_BWD   arg = const int64_t[2] {r, k}          `ds_ws` of vlfb_fbo_attn_bwd (r RoIs x k bank rows, fp32)
 *   VLFB_WS_ATTN_SCORES    arg = const int64_t[3] {b, l1, l2}     fp32 score matrix between the scores GEMM and
 *                                                                vlfb_softmax_fwd / _bwd when the fused kernels are not used
 *   VLFB_WS_BN             arg = const int64_t[3] {dtype, rows, C} `workspace` of vlfb_bn_fwd / _bwd (= vlfb_bn_workspace_bytes) */
enum { VLFB_WS_CONV = 0, VLFB_WS_MAXPOOL_ARGMAX = 1, VLFB_WS_FBO_ATTN_BWD = 2, VLFB_WS_ATTN_SCORES = 3, VLFB_WS_BN = 4 };
int64_t vlfb_query_workspace(int op, const void* arg);
/* A: activation / gradient operand; B: weight operand (FPROP/DGRAD) or unused (WGRAD);
 * P: WGRAD output-gradient operand; O: output; bias/rowscale: fp32 vectors or NULL;
 * R, Mask: optional, dtype elements. */
int vlfb_conv_run(const vlfb_conv_desc* d, const void* A, const void* B, const void* P, void* O,
                  const float* bias, const float* rowscale, const void* R, const void* Mask,
                  void* workspace, int64_t workspace_bytes, vlfb_stream_t stream);
/* the same with the destination of the output's term planes (desc.o_planes > 0; NULL otherwise) */
int vlfb_conv_run_planes(const vlfb_conv_desc* d, const void* A, const void* B, const void* P, void* O,
                         const float* bias, const float* rowscale, const void* R, const void* Mask,
                         void* workspace, int64_t workspace_bytes, void* O_planes, vlfb_stream_t stream);

/* Every operand of a launch in one struct (what the three entry points above pass, plus):
 *   R_lo / O_lo  16-bit FPROP / DGRAD launches with 16-bit outputs: a TWO-TERM residual / output.  The epilogue computes
 *                v = alpha * acc + bias + R + R_lo; relu; mask  and stores O = T(v), O_lo = T(v - O): a running sum that is
 *                re-rounded by every launch of a chain (the residual-stream gradient of the bottleneck stack) keeps ~22
 *                significant bits while its leading term O stays a plain 16-bit MFMA operand.  Either may be NULL.
 *                With an fp32 output (out_dtype VLFB_F32 on 16-bit operands) O_lo alone is accepted as well: it receives the
 *                output ROUNDED to the operand type -- the copy the next 16-bit launch reads, without a cast pass. */
typedef struct vlfb_conv_args {
  const void* A; const void* B; const void* P; void* O;
  const float* bias; const float* rowscale;
  const void* R; const void* Mask;
  void* workspace; int64_t workspace_bytes;
  void* O_planes; float* dbias;
  const void* R_lo; void* O_lo;
} vlfb_conv_args;
int vlfb_conv_run_args(const vlfb_conv_desc* d, const vlfb_conv_args* a, vlfb_stream_t stream);

/* WGRAD with desc.wgrad_bias = 1: weight gradient O and bias gradient dbias (fp32 [Cn]) of a conv that carries a bias
 * (nonlocal_helper.py:36-77, lfb_helper.py:175-200) from ONE pass over the output gradient P: the 16-bit transposed-read
 * kernel sums the P tiles it stages anyway (per-split partial rows, folded in split order: deterministic); for every other
 * kernel family the library runs a column-sum pass behind the launch -- the caller does not have to know which. */
int vlfb_conv_run_wgrad_bias(const vlfb_conv_desc* d, const void* A, const void* P, void* O, float* dbias,
                             const float* rowscale, void* workspace, int64_t workspace_bytes, vlfb_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Layout / dtype movers at the boundary.
 * ------------------------------------------------------------------------------------------ */
/* fp32 NCTHW (the `data` blob, lib/datasets/ava_data_input.py:143-161) -> NTHWC with C padded
 * to c_pad (zeros) in `dtype`. */
int vlfb_ncthw_to_nthwc(const float* src, void* dst, int dtype, int64_t n, int64_t c, int64_t thw,
                        int64_t c_pad, vlfb_stream_t stream);
/* The stem's input format: as above with `wpad_left` / (w_total - wpad_left - w) zero pixels on the
 * two sides of every W row, [N][rows = T*H][w_total][c_pad].  With it the packed conv1 kernels
 * (pack_w) never test w bounds; pass Ws = w_total and pw = pad_w - wpad_left in the conv desc. */
int vlfb_ncthw_to_nthwc_wpad(const float* src, void* dst, int dtype, int64_t n, int64_t c, int64_t rows,
                             int64_t w, int64_t c_pad, int64_t wpad_left, int64_t w_total,
                             vlfb_stream_t stream);
/* NTHWC `dtype` -> fp32 NCTHW (for FetchBlob of activations / gradients) */
int vlfb_nthwc_to_ncthw(const void* src, float* dst, int dtype, int64_t n, int64_t c, int64_t thw,
                        vlfb_stream_t stream);
/* fp32 -> two-plane fp16 (VLFB_F16PAIR: dst[0..n) = hi, dst[n..2n) = lo) and back; n % 8 == 0 */
int vlfb_pair_split(const float* src, void* dst_pair, int64_t n, vlfb_stream_t stream);
int vlfb_pair_join(const void* src_pair, float* dst, int64_t n, vlfb_stream_t stream);
/* generic cast fp32 <-> dtype, n elements */
int vlfb_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n,
              vlfb_stream_t stream);
/* fp32 -> fp16 copy whose sign pattern equals the source's (a positive value below the fp16 subnormal range becomes the
 * smallest subnormal, not +0): the activations the "mix" path's fp16 backward reads as WGRAD operands and ReLU masks */
int vlfb_half_copy(const float* src, void* dst_f16, int64_t n, vlfb_stream_t stream);
/* batched 2-D transpose of dtype elements: dst[b][j][i] = src[b][i][j], src is rows x cols */
int vlfb_transpose2d(const void* src, void* dst, int dtype, int64_t batch, int64_t rows,
                     int64_t cols, vlfb_stream_t stream);
/* strided 2-D copy of dtype elements (Concat axis=1 and its gradient, head_helper.py:55,82) */
int vlfb_copy2d(const void* src, int64_t lds, void* dst, int64_t ldd, int dtype, int64_t rows,
                int64_t cols, vlfb_stream_t stream);
int vlfb_zero_f32(float* p, int64_t n, vlfb_stream_t stream);
/* fp32 [batch][rows][cols] -> `nplanes` (2 | 3) bf16 term planes [plane][batch][rows][cols] (transpose = 0) or
 * [plane][batch][cols][rows] (transpose = 1): the B operands of split-bf16 launches that are activations (the
 * attention products, nonlocal_helper.py:94-121) */
int vlfb_split_planes(const float* src, void* dst, int nplanes, int64_t batch, int64_t rows, int64_t cols,
                      int transpose, vlfb_stream_t stream);
/* Weight preparation: fp32 master W[Cout][taps][Cin] (kernel K-order) and the frozen affine
 * scale s[Cout] (may be NULL = 1) -> operand copies in `dtype`:
 *   w_fprop[Cout][taps][Cin] = W*s          (B operand of FPROP)
 *   w_dgrad[Cin][taps][Cout] = W*s          (B operand of DGRAD; may be NULL)
 * Folding the frozen scale is exact algebra: affine(conv(x,W)) = conv(x, W*s) + b
 * (affine_nd_op.cu:31-44 has no gradient for s, affine_nd_op.cc:45-53). */
int vlfb_weight_prep(const float* w, const float* scale, void* w_fprop, void* w_dgrad, int dtype,
                     int64_t cout, int64_t taps, int64_t cin, vlfb_stream_t stream);

/* The same for every convolution of the model in ONE launch.  `items_dev` is a device array sorted
 * by tile_begin; item i owns tiles [tile_begin, tile_begin + taps*ceil(cout/32)*ceil(cin/32)). */
typedef struct vlfb_wprep_item {
  const void* w;        /* fp32 master [cout][taps][cin] */
  const void* scale;    /* fp32 [cout] or NULL */
  void* w_fprop;        /* dtype [cout][taps][cin] or NULL */
  void* w_dgrad;        /* dtype [cin][taps][cout] or NULL */
  int32_t cout, taps, cin;
  int32_t tile_begin;
} vlfb_wprep_item;
int vlfb_weight_prep_batched(const vlfb_wprep_item* items_dev, int n_items, int total_tiles,
                             int dtype, vlfb_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Pools (channels-last).  Replace Caffe2 MaxPool / AveragePool:
 * resnet_video.py:190,219; nonlocal_helper.py:49; head_helper.py:37,92,113; lfb_helper.py:112,124.
 * ------------------------------------------------------------------------------------------ */
typedef struct vlfb_pool_desc {
  int32_t dtype;
  int32_t N, Ti, Hi, Wi, C;
  int32_t To, Ho, Wo;
  int32_t kt, kh, kw, st, sh, sw, pt, ph, pw;
} vlfb_pool_desc;
/* bytes per arg-max element: 1 (window <= 255 taps) or 2 (the FBO-max head pools 300 bank rows) */
int vlfb_pool_argmax_bytes(const vlfb_pool_desc* d);
/* y[N,To,Ho,Wo,C]; argmax (tap index inside the window, first maximum in t,h,w scan order; element
 * size = vlfb_pool_argmax_bytes) may be NULL in inference. */
int vlfb_maxpool_fwd(const vlfb_pool_desc* d, const void* x, void* y, void* argmax,
                     vlfb_stream_t stream);
/* dx = (accumulate ? add : 0) + scatter(dy) ; then dx = (mask > 0) ? dx : 0 when mask != NULL.
 * `add` may alias dx. Gather formulation: deterministic, no atomics. */
int vlfb_maxpool_bwd(const vlfb_pool_desc* d, const void* dy, const void* argmax, void* dx,
                     const void* add, const void* mask, vlfb_stream_t stream);
/* The same backward for a max-pool whose INPUT is a ReLU output and has no other consumer (pool1 after conv1, pool2
 * after res2: resnet_video.py:183-194, 229-236): the selected element is the pooled value itself, so the ReLU mask
 * of the input is `y > 0` read at the window's output position -- bit-identical to vlfb_maxpool_bwd(mask = x) and
 * kt*kh*kw / (st*sh*sw) times less mask traffic. */
int vlfb_maxpool_relu_bwd(const vlfb_pool_desc* d, const void* dy, const void* argmax, const void* y, void* dx,
                          vlfb_stream_t stream);
/* average over the window (pad 0 only, as every AveragePool in the reference) */
int vlfb_avgpool_fwd(const vlfb_pool_desc* d, const void* x, void* y, vlfb_stream_t stream);
int vlfb_avgpool_bwd(const vlfb_pool_desc* d, const void* dy, void* dx, const void* add,
                     const void* mask, vlfb_stream_t stream);

/* The average-pool backward from an fp32 pooled gradient dy into a TWO-TERM 16-bit input gradient (d->dtype = VLFB_F16 /
 * VLFB_BF16): dx = (mask > 0 ? sum dy / window : 0), dx_hi = round(dx), dx_lo = round(dx - dx_hi) -- see vlfb_conv_args.O_lo.
 * Where the "mix" path's fp32 head gradient (head_helper.py:37-40, 92-98: the pools over res5) re-enters the 16-bit backward:
 * every position of a channel receives the same value, so a one-term rounding is an error common to all positions. */
int vlfb_avgpool_bwd_two_term(const vlfb_pool_desc* d, const float* dy, void* dx_hi, void* dx_lo, const void* mask,
                              vlfb_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Row softmax with pre-scale: P[r][:] = softmax(scale * S[r][:]).
 * Replaces Scale + Softmax(axis=2, engine=CUDNN): nonlocal_helper.py:98-104, lfb_helper.py:227-230.
 * S is fp32 (the affinity is kept in fp32 on both paths); P is `dtype`.
 * bwd: dS[r][:] = scale * P .* (dP - sum(dP .* P)), dP fp32, dS `dtype`.
 * ------------------------------------------------------------------------------------------ */
int vlfb_softmax_fwd(const float* s, void* p, int dtype, int64_t rows, int64_t cols, float scale,
                     vlfb_stream_t stream);
int vlfb_softmax_bwd(const float* dp, const void* p, void* ds, int dtype, int64_t rows,
                     int64_t cols, float scale, vlfb_stream_t stream);
/* bwd with the probabilities in fp32 and ds in a 16-bit type (cols % 4 == 0, cols <= 2048): the "mix" path */
int vlfb_softmax_bwd_p32(const float* dp, const float* p, void* ds, int ds_dtype, int64_t rows, int64_t cols,
                         float scale, vlfb_stream_t stream);
/* The same two operators FUSED with the batched product that feeds them (nonlocal_helper.py:94-121), so the fp32
 * score matrix never exists in memory:
 *   fwd:  prob[b][l1][l2] = softmax_l2(scale * sum_c theta[b][l1][c] * phi[b][l2][c])
 *   bwd:  ds[b][l1][l2]   = scale * prob o (dP - sum_l2(dP o prob)),  dP = sum_c dy[b][l1][c] * g[b][l2][c]
 * operands / outputs are `dtype` (16-bit types only), row-major as indexed above.  Supported shapes:
 * 512 <= l2 <= 1024 with l2 % 8 == 0, ci % 64 == 0; otherwise VLFB_ERR_UNSUPPORTED and the caller composes
 * vlfb_conv_run + vlfb_softmax_*.  vlfb_attn_scores_supported returns 0 or a mask of the flags below: whether the
 * shape can run at all, and for which direction the fused kernel was MEASURED faster than the composed one on
 * MI355X (the engine fuses a direction only then). */
enum { VLFB_ATTN_CAN_RUN = 1, VLFB_ATTN_FWD_FASTER = 2, VLFB_ATTN_BWD_FASTER = 4 };
int vlfb_attn_scores_supported(int dtype, int64_t l1, int64_t l2, int64_t ci);
int vlfb_attn_scores_fwd(const void* theta, const void* phi, void* prob, int dtype, int64_t batch, int64_t l1,
                         int64_t l2, int64_t ci, float scale, vlfb_stream_t stream);
int vlfb_attn_scores_bwd(const void* dy, const void* g, const void* prob, void* ds, int dtype, int64_t batch,
                         int64_t l1, int64_t l2, int64_t ci, float scale, vlfb_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Small elementwise / reduction ops (rows x cols matrices of `dtype`, cols contiguous).
 * ------------------------------------------------------------------------------------------ */
/* y = a + b (optionally relu, optionally masked by mask>0); any of y,a,b may alias.
 * Replaces Sum (+Relu): resnet_helper.py:112-117; nonlocal_helper.py:170,201; lfb_helper.py:286 */
int vlfb_add(const void* a, const void* b, void* y, const void* mask, int dtype, int64_t n,
             int relu, vlfb_stream_t stream);
/* y = relu(x); dx = (y>0) ? dy : 0  (model_builder_video.py:169-174) */
int vlfb_relu_fwd(const void* x, void* y, int dtype, int64_t n, vlfb_stream_t stream);
int vlfb_relu_bwd(const void* dy, const void* y, void* dx, int dtype, int64_t n,
                  vlfb_stream_t stream);
/* colsum[c] (+)= sum_r g[r][c] -- bias gradients of the convs that carry a bias
 * (nonlocal_helper.py:36-77, lfb_helper.py:175-200, resnet_video.py:327).  Row slabs are folded with fp32 atomics (no scratch
 * argument): not bit-reproducible.  The engine takes its bias gradients from vlfb_conv_run_wgrad_bias, which is. */
int vlfb_colsum(const void* g, int dtype, int64_t rows, int64_t cols, int64_t ld, float* out,
                int accumulate, vlfb_stream_t stream);
/* ------------------------------------------------------------------------------------------
 * SpatialBN over channels-last rows x[rows][C] -- the graphs built with MODEL.USE_AFFINE False / NONLOCAL.USE_BN True
 * (model_builder_video.py:176-197 Conv3dBN, resnet_video.py:185-188, nonlocal_helper.py:146-155).  The operator is
 * Caffe2's spatial_batch_norm_op (a dependency, not in the reference tree); its algorithm:
 *   train (is_test = 0): mu / var = biased row moments; y = (x - mu) / sqrt(var + eps) * gamma + beta;
 *          running_mean = momentum * running_mean + (1 - momentum) * mu,
 *          running_var  = momentum * running_var  + (1 - momentum) * var * rows / (rows - 1);
 *          save_mean / save_inv_std [C] are kept for the backward pass
 *   test  (is_test = 1): y = (x - running_mean) / sqrt(running_var + eps) * gamma + beta; nothing else is written
 *   bwd:   dbeta = sum dy, dgamma = sum dy * xhat (each times grad_scale; either may be NULL),
 *          dx = gamma * inv_std * (dy - dbeta / rows - xhat * dgamma / rows)   (NULL: parameter gradients only)
 * Per-GPU statistics, as in the reference's data-parallel model.  Reductions are two-stage and ordered (no atomics).
 * x may alias y, dy may alias dx.  workspace: vlfb_bn_workspace_bytes (fp32 slab partials + coefficient rows).
 * ------------------------------------------------------------------------------------------ */
int64_t vlfb_bn_workspace_bytes(int dtype, int64_t rows, int64_t C);
int vlfb_bn_fwd(const void* x, void* y, const float* gamma, const float* beta, float* running_mean,
                float* running_var, float* save_mean, float* save_inv_std, void* workspace,
                int64_t workspace_bytes, int dtype, int64_t rows, int64_t C, float eps, float momentum,
                int is_test, vlfb_stream_t stream);
int vlfb_bn_bwd(const void* dy, const void* x, const float* gamma, const float* save_mean,
                const float* save_inv_std, void* dx, float* dgamma, float* dbeta, void* workspace,
                int64_t workspace_bytes, int dtype, int64_t rows, int64_t C, float grad_scale,
                vlfb_stream_t stream);

/* LayerNorm over each row, no learnable scale/bias, eps inside sqrt (lfb_helper.py:160-166,
 * 252-256; Caffe2 LayerNorm axis=1).  rstd[r] = 1/sqrt(var+eps) saved for backward. */
int vlfb_layernorm_fwd(const void* x, void* y, float* rstd, int dtype, int64_t rows, int64_t cols,
                       float eps, vlfb_stream_t stream);
int vlfb_layernorm_bwd(const void* dy, const void* y, const float* rstd, void* dx, int dtype,
                       int64_t rows, int64_t cols, vlfb_stream_t stream);
/* Dropout (lfb_helper.py:259,314,334; resnet_video.py:323): counter-based mask
 * keep(i) = u(seed, i) >= ratio with u from vlfb's mix32 generator over the REFERENCE-layout
 * linear index i; y = keep ? x/(1-ratio) : 0.  The mask is written as bytes for backward.
 * Logical tensor is (rows, ch, inner) in reference order [(r*ch + c)*inner + k] while the
 * data is stored [(r*inner + k)*ch + c] (channels-last). */
int vlfb_dropout_fwd(const void* x, void* y, uint8_t* mask, int dtype, int64_t rows, int64_t inner,
                     int64_t ch, float ratio, uint64_t seed, vlfb_stream_t stream);
/* the same with the seed read from device memory at run time: a step captured in a HIP graph replays
 * the same launch packet every iteration, so per-iteration values cannot be kernel arguments */
int vlfb_dropout_fwd_dev(const void* x, void* y, uint8_t* mask, int dtype, int64_t rows, int64_t inner,
                         int64_t ch, float ratio, const uint64_t* seed_dev, vlfb_stream_t stream);
int vlfb_dropout_bwd(const void* dy, const uint8_t* mask, void* dx, int dtype, int64_t n,
                     float ratio, vlfb_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * FC + loss.  Replace FC, Sigmoid, SigmoidCrossEntropyLoss (resnet_video.py:327-338).
 * ------------------------------------------------------------------------------------------ */
/* logits[r][k] = sum_c x[r][c] * w[k][c] + b[k]; x `dtype`, w/b fp32 master, logits fp32 */
int vlfb_fc_fwd(const void* x, int dtype, const float* w, const float* b, float* logits,
                int64_t rows, int64_t cin, int64_t cout, vlfb_stream_t stream);
/* dx[r][c] = sum_k dl[r][k]*w[k][c] ; dw[k][c] (+)= sum_r dl[r][k]*x[r][c]; db[k] (+)= sum_r dl */
int vlfb_fc_bwd(const void* x, int dtype, const float* w, const float* dlogits, void* dx,
                float* dw, float* db, int64_t rows, int64_t cin, int64_t cout, int accumulate,
                vlfb_stream_t stream);
/* prob = sigmoid(logits); loss = scale * sum(l) / max(#(t>=0),1) with
 * l = -x*(t - (x>=0)) + log(1 + exp(x - 2x*(x>=0))), t in {0,1} (or -1 = ignore);
 * dlogits = scale * (prob - t) / normalizer (0 where ignored).  loss: 1 float. */
int vlfb_sigmoid_ce(const float* logits, const int32_t* labels, float* prob, float* loss,
                    float* dlogits, int64_t rows, int64_t cols, float scale, vlfb_stream_t stream);
/* Single-label heads (EPIC-Kitchens; Softmax / SoftmaxWithLoss, resnet_video.py:339-347): prob = softmax over
 * `cols`; with labels [rows] (class indices): loss = scale * sum_r -log(max(prob[r][label_r], 1e-20)) / rows and
 * dlogits = scale * (prob - onehot) / rows (Caffe2 SoftmaxWithLoss without weights).  labels NULL = test mode. */
int vlfb_softmax_ce(const float* logits, const int32_t* labels, float* prob, float* loss, float* dlogits,
                    int64_t rows, int64_t cols, float scale, vlfb_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * RoI head.  Replaces RoIAlign + 7x7 MaxPool (head_helper.py:88-123, lfb_helper.py:130-152).
 * feat: [N,H,W,C] channels-last `dtype`; rois: fp32 (R,5) = [batch_idx, x1,y1,x2,y2];
 * out: [R,C] = max over the pooled x pooled RoIAlign bins; argbin: uint8 [R,C].
 * dbg (optional, int32 [R][pooled][pooled][8]) receives the integer decisions of the first
 * sample of every bin: {batch, grid_h, grid_w, y_low, x_low, y_high, x_high, inside}.
 * ------------------------------------------------------------------------------------------ */
int vlfb_roi_align_max_fwd(const void* feat, int dtype, const float* rois, void* out,
                           uint8_t* argbin, int32_t* dbg, int64_t n, int64_t h, int64_t w,
                           int64_t c, int64_t r, int pooled, float spatial_scale,
                           vlfb_stream_t stream);
/* Test hook: the integer decisions of EVERY bilinear sample of every bin, from the same device functions the operator
 * uses: dbg int32 [R][pooled][pooled][max_grid][max_grid][8] = {batch, grid_h, grid_w, y_low, x_low, y_high, x_high,
 * inside}; samples beyond the RoI's adaptive grid carry {.., -2, -2, -2, -2, -1}. */
int vlfb_roi_align_decisions(const float* rois, int32_t* dbg, int64_t h, int64_t w, int64_t r, int pooled,
                             float spatial_scale, int max_grid, vlfb_stream_t stream);
/* dfeat (fp32 [N,H,W,C], must be zeroed by the caller) += scatter of dout through argbin.  Deterministic: one thread owns a
 * (clip, channel) column and adds its RoIs' contributions in RoI / sample / corner order (the reference operator scatters
 * with atomics, whose order -- overlapping RoIs of a clip -- varies from run to run). */
int vlfb_roi_align_max_bwd(const void* dout, int dtype, const float* rois, const uint8_t* argbin,
                           float* dfeat, int64_t n, int64_t h, int64_t w, int64_t c, int64_t r,
                           int pooled, float spatial_scale, vlfb_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * FBO-NL attention core for one query per row (lfb_helper.py:170-263, num_feat1 == 1):
 *   aff[r][k] = scale * <theta[r], phi[r][k]>, p = softmax_k(aff), t[r] = sum_k p[r][k] g[r][k].
 * theta [R][D], phi/g [R][K][D] (row stride ld), p fp32 [R][K], t [R][D].
 * ------------------------------------------------------------------------------------------ */
int vlfb_fbo_attn_fwd(const void* theta, const void* phi, const void* g, float* p, void* t,
                      int dtype, int64_t r, int64_t k, int64_t d, int64_t ld, float scale,
                      vlfb_stream_t stream);
/* The same with SHARED banks (inference: tools/lfb_loader.py, tools/test_net.py): the data layer hands every RoI a COPY of its
 * clip's bank (lib/datasets/ava_data_input.py:191-192) and the graph projects each copy (lib/models/lfb_helper.py:320-338);
 * without dropout the projected banks of a clip's RoIs are identical, so phi / g are computed ONCE per clip -- [n_banks][K][D]
 * -- and row r attends to bank (int)owner[r * owner_stride] (the batch-index column of the `proposals` blob: owner = rois,
 * owner_stride = 5).  Same arithmetic per row as vlfb_fbo_attn_fwd on the duplicated banks: bit-identical outputs. */
int vlfb_fbo_attn_fwd_shared(const void* theta, const void* phi, const void* g, float* p, void* t,
                             int dtype, int64_t r, int64_t k, int64_t d, int64_t ld, float scale,
                             const float* owner, int64_t owner_stride, vlfb_stream_t stream);
/* ds_ws: fp32 scratch of r*k elements (the softmax-input gradient handed from the per-row
 * kernel to the chip-wide dphi/dg writer) */
int vlfb_fbo_attn_bwd(const void* dt, const void* theta, const void* phi, const void* g,
                      const float* p, void* dtheta, void* dphi, void* dg, float* ds_ws, int dtype,
                      int64_t r, int64_t k, int64_t d, int64_t ld, float scale, vlfb_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Solver.  Replaces WeightedSum + MomentumSGDUpdate(nesterov) per parameter
 * (model_builder_video.py:348-389) with one pass over a flat fp32 bucket:
 *   g += wd*p ; m' = mu*m + lr*g ; p -= (1+mu)*m' - mu*m   (nesterov)  |  p -= m' (plain)
 * ------------------------------------------------------------------------------------------ */
int vlfb_sgd_update(float* p, float* g, float* m, int64_t n, float lr, float wd, float mu,
                    int nesterov, vlfb_stream_t stream);
/* learning rate read from device memory (captured steps, see vlfb_dropout_fwd_dev) */
int vlfb_sgd_update_dev(float* p, float* g, float* m, int64_t n, const float* lr_dev, float wd, float mu,
                        int nesterov, vlfb_stream_t stream);
/* dst[i] = values[i], i < n <= 8, in stream order; the values are copied into the launch packet before the
 * call returns (per-iteration scalars of a captured step: learning rate bits, dropout seeds) */
int vlfb_store_scalars(uint64_t* dst, int n, const uint64_t* values, vlfb_stream_t stream);
int vlfb_scale_inplace(float* x, int64_t n, float s, vlfb_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Long-term feature bank resident on the device.
 * Replaces the host dict-of-lists bank of tools/lfb_loader.py:49-112 (construct_ava_lfb,
 * construct_frame_level_lfb) and the per-clip NumPy sampling of lib/datasets/ava.py:300-323 and
 * lib/datasets/charades.py:251-276.
 *   bank  [n_videos][n_steps][capacity][dim]  elements of `dtype`, caller-owned
 *   count [n_videos][n_steps] int32, caller-owned, zero = empty
 * AVA: step = second (minus a base), capacity = most boxes kept per second.
 * Charades: step = LFB frame slot, capacity = 1.
 * ------------------------------------------------------------------------------------------ */
typedef struct vlfb_lfb_desc {
  int32_t n_videos, n_steps, capacity, dim;
  int32_t dtype;
  int32_t step_base;   /* reference time key of step 0 (AVA seconds start at 902); only the sampling
                          draw depends on it, so that it is a function of the reference's own keys */
} vlfb_lfb_desc;
int64_t vlfb_lfb_bank_bytes(const vlfb_lfb_desc* d);
/* Append `rows` features (feats [rows][dim] of feat_dtype) under keys [rows][2] = {video, step}.
 * Rows of the same key keep their batch order behind what the bank already holds (list.append of
 * lfb_loader.py:100-104); rows with video < 0 are padding and skipped; rows that do not fit
 * (bad key / cell full) are counted in *dropped (optional device int32). */
int vlfb_lfb_append(const vlfb_lfb_desc* d, void* bank, int32_t* count, const void* feats,
                    int feat_dtype, const int32_t* keys, int64_t rows, int32_t* dropped,
                    vlfb_stream_t stream);
/* AVA window sampling (ava.py:300-323): query [rows][3] = {video, centre step, sample id};
 * out [rows][window*max_per_step][dim]; time slot j holds min(n, max_per_step) distinct features
 * of step centre - window/2 + j in random order, zeros elsewhere.  The draw is a pure function of
 * (seed, sample id, video, step): rows sharing a sample id get the same features (the reference
 * repeats the clip's sample for each of its RoIs, ava_data_input.py:191-192). */
int vlfb_lfb_sample_window(const vlfb_lfb_desc* d, const void* bank, const int32_t* count,
                           const int32_t* query, int64_t rows, int window, int max_per_step,
                           uint64_t seed, void* out, int out_dtype, vlfb_stream_t stream);
/* The AVA window from a HOST-DRAWN table (stream-equal with the reference: the host makes the np.random.choice calls of
 * ava.py:316-318, the device gathers): query [rows][2] = {video, centre step}; table [rows][window][max_per_step] = slot of
 * step (centre - window/2 + j) that becomes output row j*max_per_step + k, or -1 for zeros. */
int vlfb_lfb_gather_slots(const vlfb_lfb_desc* d, const void* bank, const int32_t* count, const int32_t* query,
                          const int32_t* table, int64_t rows, int window, int max_per_step, void* out, int out_dtype,
                          vlfb_stream_t stream);
/* Frame-level sampling (charades.py:251-276): query [rows][3] = {video, first step, last step};
 * out [rows][window][dim] = the first `window` occupied steps of [first, last] packed to the
 * front, zeros behind. */
int vlfb_lfb_sample_compact(const vlfb_lfb_desc* d, const void* bank, const int32_t* count,
                            const int32_t* query, int64_t rows, int window, void* out,
                            int out_dtype, vlfb_stream_t stream);
/* The same with up to `max_per_step` features per step (slot order), packed in step order until `window`
 * rows are filled: EPIC-Kitchens verb banks (one clip feature per step, lib/datasets/epic.py:310-331,
 * max_per_step = 1) and noun banks (detector features per frame, epic.py:338-374, max_per_step =
 * EPIC.MAX_NUM_FEATS_PER_NOUN_LFB_FRAME). */
int vlfb_lfb_sample_packed(const vlfb_lfb_desc* d, const void* bank, const int32_t* count,
                           const int32_t* query, int64_t rows, int window, int max_per_step, void* out,
                           int out_dtype, vlfb_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Clip preprocessing on the device.  Replaces the per-frame cv2 / NumPy chain of
 * lib/datasets/data_input_helper.py:70-139 (images_and_boxes_preprocessing) with one kernel:
 * uint8 BGR frames [frames][src_h][src_w][3] -> bilinear resize to resized_h x resized_w (OpenCV's
 * 8-bit INTER_LINEAR fixed-point algorithm; xofs/xcoef/yofs/ycoef are cv::resize's per-column / per-row
 * source index and 11-bit weight tables, may be NULL when no resize happens) -> crop: output pixel
 * (y, x) takes resized pixel (y0 + y, x0 + x), or (y0 + y, x0 - x) when flip is set (x0 is then the
 * RIGHT edge of the window: train flips after the crop, test before it) -> x / 255 -> (x - mean[c]) / std[c] (mean / std in
 * the source channel order) -> optional BGR -> RGB, written to dst rows of w_total pixels x c_pad
 * channels starting at pixel w_left (the W-padded, channel-padded layout of the model's data input;
 * padding is not touched).  dst points at frame 0 of the destination clip.
 * ------------------------------------------------------------------------------------------ */
typedef struct vlfb_clip_desc {
  int32_t frames, src_h, src_w, resized_h, resized_w;
  int32_t crop_h, crop_w, y0, x0, flip;
  int32_t to_rgb, w_left, w_total, c_pad;
  float mean[3], std[3];
} vlfb_clip_desc;
int vlfb_clip_preprocess(const vlfb_clip_desc* d, const uint8_t* frames, const int32_t* xofs,
                         const int16_t* xcoef, const int32_t* yofs, const int16_t* ycoef, void* dst,
                         int dst_dtype, vlfb_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VLFB_H_ */
