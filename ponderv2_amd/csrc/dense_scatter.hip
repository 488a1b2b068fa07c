// Sparse rows -> dense grid scatter (sum / mean) and its backward, for PonderIndoor.to_dense.
//
// Stands in for torch_scatter.scatter(src, index, dim=0, reduce=..., out=...) at
// ponder/models/ponder/ponder_indoor_base.py:214 and ponder_outdoor_base.py:204.
// One wave moves one source row: lanes run along the channel axis, so both the row read and the
// atomic adds into the (channels-last) dense grid are contiguous.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void scatter_add_kernel(const float* __restrict__ src,
                                                          const int64_t* __restrict__ index,
                                                          int64_t m, int c, float* __restrict__ out,
                                                          float* __restrict__ count, int64_t g) {
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < m; row += nwaves) {
    const int64_t dst = index[row];
    if (dst < 0 || dst >= g) continue;
    for (int ch = lane; ch < c; ch += 64)
      unsafeAtomicAdd(out + dst * c + ch, src[row * c + ch]);
    if (count != nullptr && lane == 0) unsafeAtomicAdd(count + dst, 1.0f);
  }
}

__global__ __launch_bounds__(256) void mean_finish_kernel(float* __restrict__ out,
                                                          const float* __restrict__ count,
                                                          int64_t g, int c) {
  const int64_t total = g * c;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const float n = count[i / c];
    if (n > 1.0f) out[i] = out[i] / n;
  }
}

__global__ __launch_bounds__(256) void scatter_bwd_kernel(const float* __restrict__ dout,
                                                          const int64_t* __restrict__ index,
                                                          const float* __restrict__ count,
                                                          int64_t m, int c, float* __restrict__ dsrc,
                                                          int64_t g) {
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < m; row += nwaves) {
    const int64_t dst = index[row];
    const bool ok = dst >= 0 && dst < g;
    float inv = 1.0f;
    if (ok && count != nullptr) inv = 1.0f / fmaxf(count[dst], 1.0f);
    for (int ch = lane; ch < c; ch += 64)
      dsrc[row * c + ch] = ok ? dout[dst * c + ch] * inv : 0.0f;
  }
}

}  // namespace

namespace {
// out[i, :] = table[index[i], :] for a SMALL table (it lives in L1): a pure 16-byte-vector write of the
// output.  torch's index_select on an int64 index ran this 134 MB fill at 0.6 TB/s.
__global__ __launch_bounds__(256) void gather_rows_kernel(const float4* __restrict__ table,
                                                         const int32_t* __restrict__ index, int64_t n,
                                                         int c4, float4* __restrict__ out) {
  const int64_t total = n * c4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += stride) {
    const int64_t i = q / c4;
    const int j = (int)(q - i * c4);
    out[q] = table[(int64_t)index[i] * c4 + j];
  }
}
}  // namespace

extern "C" {

int pv2_gather_rows(const float* table, const int32_t* index, int64_t n, int c, float* out,
                    pv2_stream_t stream) {
  PV2_REQUIRE(c >= 4 && (c % 4) == 0 && n >= 0, "pv2_gather_rows: c must be a multiple of 4");
  if (n == 0) return PV2_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(pv2::grid_for(n * (c / 4), 256)), dim3(256), 0,
                     (hipStream_t)stream, (const float4*)table, index, n, c / 4, (float4*)out);
  return pv2::check_launch("gather_rows");
}

int pv2_scatter_add(const float* src, const int64_t* index, int64_t m, int c, float* out,
                    float* count, int64_t g, pv2_stream_t stream) {
  PV2_REQUIRE(c >= 1 && g >= 1, "pv2_scatter_add: bad sizes");
  if (m == 0) return PV2_OK;
  hipLaunchKernelGGL(scatter_add_kernel, dim3(pv2::grid_for(m * 64, 256)), dim3(256), 0,
                     (hipStream_t)stream, src, index, m, c, out, count, g);
  return pv2::check_launch("scatter_add");
}

int pv2_scatter_mean_finish(float* out, const float* count, int64_t g, int c,
                            pv2_stream_t stream) {
  PV2_REQUIRE(c >= 1 && g >= 1, "pv2_scatter_mean_finish: bad sizes");
  hipLaunchKernelGGL(mean_finish_kernel, dim3(pv2::grid_for(g * c, 256)), dim3(256), 0,
                     (hipStream_t)stream, out, count, g, c);
  return pv2::check_launch("scatter_mean_finish");
}

int pv2_scatter_backward(const float* dout, const int64_t* index, const float* count, int64_t m,
                         int c, float* dsrc, int64_t g, pv2_stream_t stream) {
  PV2_REQUIRE(c >= 1, "pv2_scatter_backward: bad sizes");
  if (m == 0) return PV2_OK;
  hipLaunchKernelGGL(scatter_bwd_kernel, dim3(pv2::grid_for(m * 64, 256)), dim3(256), 0,
                     (hipStream_t)stream, dout, index, count, m, c, dsrc, g);
  return pv2::check_launch("scatter_backward");
}

}  // extern "C"
