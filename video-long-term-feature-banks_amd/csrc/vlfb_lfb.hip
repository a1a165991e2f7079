// Long-term feature bank resident in HBM.
//
// The reference keeps the bank on the host as dict-of-lists (tools/lfb_loader.py:79-112), samples
// a window per clip with NumPy (lib/datasets/ava.py:300-323, lib/datasets/charades.py:251-276) and
// ships (R, K, 2048) fp32 through the blob queue every iteration.  Here the bank is one dense
// tensor  bank[video][step][slot][dim]  with  count[video][step],  appended to and sampled from by
// kernels, so the sampled (R, K, dim) block is written straight into the model's `lfb` input.
//
// Byte-moving, HBM-bound kernels: one workgroup per destination row group, 16-byte accesses when
// source and destination share the element type.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vlfb.h"
#include "vlfb_common.h"

namespace vlfb {
namespace {

__host__ __device__ __forceinline__ uint32_t lfb_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}
// sort key of feature i of (video, step) for the draw `sample_id` (oracle/lfb.py: choice_key)
__host__ __device__ __forceinline__ uint32_t lfb_key(uint64_t seed, uint32_t sample_id, uint32_t video,
                                                     uint32_t step, uint32_t i) {
  uint32_t h = lfb_mix32((uint32_t)seed ^ sample_id);
  h = lfb_mix32(h + 0x9e3779b9u + ((uint32_t)(seed >> 32) ^ video));
  h = lfb_mix32(h + 0x85ebca6bu + step);
  h = lfb_mix32(h + 0xc2b2ae35u + i);
  return h;
}

__device__ __forceinline__ float ld_any(const void* p, int dtype, long long i) {
  return dtype == VLFB_F32 ? reinterpret_cast<const float*>(p)[i]
         : dtype == VLFB_F16 ? h2f(reinterpret_cast<const unsigned short*>(p)[i]) : bf2f(reinterpret_cast<const bf16_t*>(p)[i]);
}
__device__ __forceinline__ void st_any(void* p, int dtype, long long i, float v) {
  if (dtype == VLFB_F32) reinterpret_cast<float*>(p)[i] = v;
  else if (dtype == VLFB_F16) reinterpret_cast<unsigned short*>(p)[i] = f2h(v);
  else reinterpret_cast<bf16_t*>(p)[i] = f2bf(v);
}
__device__ __forceinline__ int esize(int dtype) { return dtype == VLFB_F32 ? 4 : 2; }

// copy / convert / zero one row of `dim` elements with the whole workgroup
__device__ void row_copy(void* dst, int ddt, long long doff, const void* src, int sdt, long long soff, int dim) {
  if (src != nullptr && ddt == sdt && ((dim * esize(ddt)) & 15) == 0) {
    const uint4* s = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(src) + soff * esize(sdt));
    uint4* d = reinterpret_cast<uint4*>(reinterpret_cast<char*>(dst) + doff * esize(ddt));
    for (int i = threadIdx.x; i < dim * esize(ddt) / 16; i += blockDim.x) d[i] = s[i];
    return;
  }
  if (src == nullptr && ((dim * esize(ddt)) & 15) == 0) {
    uint4* d = reinterpret_cast<uint4*>(reinterpret_cast<char*>(dst) + doff * esize(ddt));
    for (int i = threadIdx.x; i < dim * esize(ddt) / 16; i += blockDim.x) d[i] = make_uint4(0, 0, 0, 0);
    return;
  }
  for (int i = threadIdx.x; i < dim; i += blockDim.x)
    st_any(dst, ddt, doff + i, src ? ld_any(src, sdt, soff + i) : 0.f);
}

struct BankP {
  int n_videos, n_steps, capacity, dim, dtype, step_base;
};

__device__ __forceinline__ bool key_ok(const BankP& b, int video, int step) {
  return video >= 0 && video < b.n_videos && step >= 0 && step < b.n_steps;
}

// row r goes to slot count[key] + #(earlier rows of this batch with the same key): the append order
// of the reference's list.append (lfb_loader.py:103) is kept, whatever order workgroups run in.
__global__ void lfb_append_kernel(BankP b, char* bank, const int32_t* __restrict__ count, const void* feats,
                                  int fdt, const int32_t* __restrict__ keys, int rows, int32_t* dropped) {
  const int r = blockIdx.x;
  const int video = keys[2 * r], step = keys[2 * r + 1];
  if (!key_ok(b, video, step)) {          // padding rows of a partial batch carry video = -1
    if (threadIdx.x == 0 && video >= 0 && dropped) atomicAdd(dropped, 1);
    return;
  }
  __shared__ int s_before;
  if (threadIdx.x == 0) s_before = 0;
  __syncthreads();
  int mine = 0;
  for (int q = threadIdx.x; q < r; q += blockDim.x) mine += (keys[2 * q] == video && keys[2 * q + 1] == step);
  if (mine) atomicAdd(&s_before, mine);
  __syncthreads();
  const long long cell = (long long)video * b.n_steps + step;
  const int slot = count[cell] + s_before;
  if (slot >= b.capacity) {
    if (threadIdx.x == 0 && dropped) atomicAdd(dropped, 1);
    return;
  }
  row_copy(bank, b.dtype, (cell * b.capacity + slot) * (long long)b.dim, feats, fdt, (long long)r * b.dim, b.dim);
}

// after the copies: the last row of every key publishes the new count
__global__ void lfb_commit_kernel(BankP b, int32_t* count, const int32_t* __restrict__ keys, int rows) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const int video = keys[2 * r], step = keys[2 * r + 1];
  if (!key_ok(b, video, step)) return;
  int same = 0;
  for (int q = 0; q < rows; ++q) {
    const bool eq = keys[2 * q] == video && keys[2 * q + 1] == step;
    if (eq && q > r) return;              // a later row owns the update
    same += eq;
  }
  const long long cell = (long long)video * b.n_steps + step;
  const int n = count[cell] + same;
  count[cell] = n < b.capacity ? n : b.capacity;
}

// AVA: out[r][j*K + k] = k-th of min(n, K) features of step (centre - window/2 + j) drawn without
// replacement, zeros elsewhere (ava.py:300-323).  grid = (window, rows).
__global__ void lfb_sample_window_kernel(BankP b, const char* __restrict__ bank, const int32_t* __restrict__ count,
                                         const int32_t* __restrict__ query, int window, int K, uint64_t seed,
                                         void* out, int odt) {
  const int j = blockIdx.x, r = blockIdx.y;
  const int video = query[3 * r], centre = query[3 * r + 1];
  const uint32_t sid = (uint32_t)query[3 * r + 2];
  const int step = centre - window / 2 + j;
  const bool ok = key_ok(b, video, step);
  const long long cell = ok ? (long long)video * b.n_steps + step : 0;
  const int n = ok ? count[cell] : 0;
  const int m = n < K ? n : K;
  __shared__ int sel[64];
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const uint32_t tkey = (uint32_t)(step + b.step_base);      // the reference's time key (e.g. the AVA second)
    const uint32_t hi = lfb_key(seed, sid, (uint32_t)video, tkey, (uint32_t)i);
    int rank = 0;
    for (int q = 0; q < n; ++q) {
      const uint32_t hq = lfb_key(seed, sid, (uint32_t)video, tkey, (uint32_t)q);
      rank += (hq < hi) || (hq == hi && q < i);
    }
    if (rank < m) sel[rank] = i;
  }
  __syncthreads();
  const long long obase = ((long long)r * window + j) * K;
  for (int k = 0; k < K; ++k) {
    if (k < m) row_copy(out, odt, (obase + k) * b.dim, bank, b.dtype, (cell * b.capacity + sel[k]) * (long long)b.dim, b.dim);
    else row_copy(out, odt, (obase + k) * b.dim, nullptr, 0, 0, b.dim);
  }
}

// The same output from a HOST-DRAWN table: table[r][j][k] = slot of step (centre - window/2 + j) that goes to output row
// j*K + k, or -1 (zeros).  For a run that has to reproduce the reference's sample STREAM (np.random.choice in
// ava.py:316-318), not just its distribution: the host draws exactly as the reference does, the device only gathers.
__global__ void lfb_gather_slots_kernel(BankP b, const char* __restrict__ bank, const int32_t* __restrict__ count,
                                        const int32_t* __restrict__ query, const int32_t* __restrict__ table, int window,
                                        int K, void* out, int odt) {
  const int j = blockIdx.x, r = blockIdx.y;
  const int video = query[2 * r], centre = query[2 * r + 1];
  const int step = centre - window / 2 + j;
  const bool ok = key_ok(b, video, step);
  const long long cell = ok ? (long long)video * b.n_steps + step : 0;
  const int n = ok ? count[cell] : 0;
  const long long obase = ((long long)r * window + j) * K;
  for (int k = 0; k < K; ++k) {
    const int slot = table[obase + k];
    if (slot >= 0 && slot < n) row_copy(out, odt, (obase + k) * b.dim, bank, b.dtype, (cell * b.capacity + slot) * (long long)b.dim, b.dim);
    else row_copy(out, odt, (obase + k) * b.dim, nullptr, 0, 0, b.dim);
  }
}

// frame-level banks (Charades, EPIC verb: one feature per step; EPIC noun: up to `max_per_step` detector
// features per step): walk the steps of [first, last] in order, take the first min(count, max_per_step)
// features of every occupied step, pack them to the front of the `window` output rows, zeros behind
// (charades.py:251-276, epic.py:310-374).  One workgroup per query.
__global__ void lfb_sample_packed_kernel(BankP b, const char* __restrict__ bank, const int32_t* __restrict__ count,
                                         const int32_t* __restrict__ query, int window, int max_per_step, void* out,
                                         int odt) {
  extern __shared__ int found[];        // [window] source rows (step * capacity + slot), then the number found
  const int r = blockIdx.x;
  const int video = query[3 * r];
  int first = query[3 * r + 1], last = query[3 * r + 2];
  if (threadIdx.x == 0) {
    int k = 0;
    if (video >= 0 && video < b.n_videos) {
      if (first < 0) first = 0;
      if (last > b.n_steps - 1) last = b.n_steps - 1;
      for (int s = first; s <= last && k < window; ++s) {
        int n = count[(long long)video * b.n_steps + s];
        if (n > max_per_step) n = max_per_step;
        for (int i = 0; i < n && k < window; ++i) found[k++] = s * b.capacity + i;
      }
    }
    found[window] = k;
  }
  __syncthreads();
  const int k = found[window];
  for (int i = 0; i < window; ++i) {
    const long long o = ((long long)r * window + i) * b.dim;
    if (i < k) row_copy(out, odt, o, bank, b.dtype, ((long long)video * b.n_steps * b.capacity + found[i]) * (long long)b.dim, b.dim);
    else row_copy(out, odt, o, nullptr, 0, 0, b.dim);
  }
}

int check_desc(const vlfb_lfb_desc* d, BankP* b) {
  VLFB_REQUIRE(d != nullptr, "lfb: descriptor is NULL");
  VLFB_REQUIRE(d->n_videos > 0 && d->n_steps > 0 && d->capacity > 0 && d->dim > 0, "lfb: empty bank geometry");
  VLFB_REQUIRE(dtype_ok(d->dtype), "lfb: bank dtype must be f32, bf16 or f16");
  b->n_videos = d->n_videos; b->n_steps = d->n_steps; b->capacity = d->capacity; b->dim = d->dim; b->dtype = d->dtype;
  b->step_base = d->step_base;
  return VLFB_OK;
}

}  // namespace
}  // namespace vlfb

using namespace vlfb;

extern "C" int64_t vlfb_lfb_bank_bytes(const vlfb_lfb_desc* d) {
  if (!d || d->n_videos <= 0 || d->n_steps <= 0 || d->capacity <= 0 || d->dim <= 0) return -1;
  return (int64_t)d->n_videos * d->n_steps * d->capacity * d->dim * (d->dtype == VLFB_F32 ? 4 : 2);
}

extern "C" int vlfb_lfb_append(const vlfb_lfb_desc* d, void* bank, int32_t* count, const void* feats,
                               int feat_dtype, const int32_t* keys, int64_t rows, int32_t* dropped,
                               vlfb_stream_t stream) {
  BankP b;
  int rc = check_desc(d, &b);
  if (rc != VLFB_OK) return rc;
  VLFB_REQUIRE(bank && count && feats && keys, "lfb_append: NULL buffer");
  VLFB_REQUIRE(dtype_ok(feat_dtype), "lfb_append: feature dtype must be f32, bf16 or f16");
  VLFB_REQUIRE(rows >= 0 && rows < (1 << 20), "lfb_append: rows out of range");
  if (rows == 0) return VLFB_OK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(lfb_append_kernel, dim3((unsigned)rows), dim3(256), 0, s, b, (char*)bank, count, feats,
                     feat_dtype, keys, (int)rows, dropped);
  hipLaunchKernelGGL(lfb_commit_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, b, count, keys, (int)rows);
  return check_launch("lfb_append");
}

extern "C" int vlfb_lfb_sample_window(const vlfb_lfb_desc* d, const void* bank, const int32_t* count,
                                      const int32_t* query, int64_t rows, int window, int max_per_step,
                                      uint64_t seed, void* out, int out_dtype, vlfb_stream_t stream) {
  BankP b;
  int rc = check_desc(d, &b);
  if (rc != VLFB_OK) return rc;
  VLFB_REQUIRE(bank && count && query && out, "lfb_sample_window: NULL buffer");
  VLFB_REQUIRE(dtype_ok(out_dtype), "lfb_sample_window: out dtype must be f32, bf16 or f16");
  VLFB_REQUIRE(window > 0 && max_per_step > 0 && max_per_step <= 64, "lfb_sample_window: need 0 < max_per_step <= 64, window > 0");
  VLFB_REQUIRE(rows >= 0 && rows < 65536, "lfb_sample_window: rows out of range");
  if (rows == 0) return VLFB_OK;
  hipLaunchKernelGGL(lfb_sample_window_kernel, dim3((unsigned)window, (unsigned)rows), dim3(256), 0,
                     (hipStream_t)stream, b, (const char*)bank, count, query, window, max_per_step, seed, out, out_dtype);
  return check_launch("lfb_sample_window");
}

extern "C" int vlfb_lfb_gather_slots(const vlfb_lfb_desc* d, const void* bank, const int32_t* count,
                                     const int32_t* query, const int32_t* table, int64_t rows, int window, int max_per_step,
                                     void* out, int out_dtype, vlfb_stream_t stream) {
  BankP b;
  int rc = check_desc(d, &b);
  if (rc != VLFB_OK) return rc;
  VLFB_REQUIRE(bank && count && query && table && out, "lfb_gather_slots: NULL buffer");
  VLFB_REQUIRE(dtype_ok(out_dtype), "lfb_gather_slots: out dtype must be f32, bf16 or f16");
  VLFB_REQUIRE(window > 0 && max_per_step > 0, "lfb_gather_slots: need window > 0, max_per_step > 0");
  VLFB_REQUIRE(rows >= 0 && rows < 65536, "lfb_gather_slots: rows out of range");
  if (rows == 0) return VLFB_OK;
  hipLaunchKernelGGL(lfb_gather_slots_kernel, dim3((unsigned)window, (unsigned)rows), dim3(256), 0, (hipStream_t)stream, b,
                     (const char*)bank, count, query, table, window, max_per_step, out, out_dtype);
  return check_launch("lfb_gather_slots");
}

extern "C" int vlfb_lfb_sample_packed(const vlfb_lfb_desc* d, const void* bank, const int32_t* count,
                                      const int32_t* query, int64_t rows, int window, int max_per_step, void* out,
                                      int out_dtype, vlfb_stream_t stream) {
  BankP b;
  int rc = check_desc(d, &b);
  if (rc != VLFB_OK) return rc;
  VLFB_REQUIRE(bank && count && query && out, "lfb_sample_packed: NULL buffer");
  VLFB_REQUIRE(dtype_ok(out_dtype), "lfb_sample_packed: out dtype must be f32, bf16 or f16");
  VLFB_REQUIRE(window > 0 && window <= 8192, "lfb_sample_packed: window out of range");
  VLFB_REQUIRE(max_per_step > 0, "lfb_sample_packed: max_per_step must be positive");
  VLFB_REQUIRE(rows >= 0 && rows < (1 << 20), "lfb_sample_packed: rows out of range");
  if (rows == 0) return VLFB_OK;
  hipLaunchKernelGGL(lfb_sample_packed_kernel, dim3((unsigned)rows), dim3(256), (size_t)(window + 1) * sizeof(int),
                     (hipStream_t)stream, b, (const char*)bank, count, query, window, max_per_step, out, out_dtype);
  return check_launch("lfb_sample_packed");
}

extern "C" int vlfb_lfb_sample_compact(const vlfb_lfb_desc* d, const void* bank, const int32_t* count,
                                       const int32_t* query, int64_t rows, int window, void* out,
                                       int out_dtype, vlfb_stream_t stream) {
  return vlfb_lfb_sample_packed(d, bank, count, query, rows, window, 1, out, out_dtype, stream);
}
