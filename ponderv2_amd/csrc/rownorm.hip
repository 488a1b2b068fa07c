// Row-matrix reductions for the sparse backbone on gfx950: training-mode BatchNorm1d over the
// active-voxel feature matrix [N, C], fused with the ReLU and the residual add that follow it in
// SpUNet's blocks, and a column sum (bias gradients of the MLP heads).
//
// Stands in for the ATen batch_norm / relu / add kernels behind
// ponder/models/sparse_unet/spconv_unet_v1m1_base.py:70-83 (BasicBlock.forward) and :108,120-121
// (norm_fn = BatchNorm1d(eps=1e-3, momentum=0.01) + ReLU inside SparseSequential).  59 BN layers
// per forward make this launch-latency territory: the stock path costs ~5 launches and >50 us per
// layer per direction; here a layer is 3 short launches per direction: per-block partial
// statistics, a combine kernel that adds the partials in a fixed order (no atomics: bitwise
// reproducible, nothing to clear) and the elementwise apply.
//
// Layout trick: a 256-thread block views consecutive rows as one flat run of floats; thread t
// always sees column t % C (C <= 256) so every load is fully coalesced and the per-column partial
// sums stay in registers until one LDS pass and one store per column per block.
#include <cstdlib>

#include "common.h"

namespace {

constexpr int kThreads = 256;

// Statistics in two steps without atomics (bitwise reproducible, nothing to clear):
//   1. col_partials_kernel: block b reduces its run of rows and writes partial[b][0..2C);
//   2. col_combine_kernel: one block per 32 channels adds the partial rows in a FIXED order (double
//      precision) and publishes mean / invstd + running statistics, or the bias / weight gradients;
//   3. the elementwise apply kernel.
// sums[0..C) = sum_r f0(r,c);  sums[C..2C) = sum_r f1(r,c)
// MODE 0: f0 = x - s, f1 = (x - s)^2 with the shift s = x[0, c] (forward statistics: the shifted form
//         keeps E[x^2] - mean^2 accurate for columns whose mean is far larger than their spread)
// MODE 1: g = dy * (y > 0 if y else 1);  f0 = g, f1 = g * xhat       (backward reductions)
constexpr int kMaxPartialBlocks = 1024;
constexpr int kMaxChannels = 1024;

template <int MODE>
__global__ __launch_bounds__(kThreads) void col_partials_kernel(
    const float* __restrict__ a, const float* __restrict__ x, const float* __restrict__ y,
    const float* __restrict__ mean_invstd, int64_t n, int c, int64_t rows_per_block,
    float* __restrict__ partial) {
  __shared__ float s0[kThreads];
  __shared__ float s1[kThreads];
  const int tid = threadIdx.x;
  for (int cb = 0; cb < c; cb += kThreads) {  // column panels of <= 256
    const int cw = min(c - cb, kThreads);
    const int rpi = kThreads / cw;             // rows covered per iteration
    const int rr = tid / cw, cc = tid % cw;
    const bool active = rr < rpi;
    float acc0 = 0.f, acc1 = 0.f;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(n, r0 + rows_per_block);
    float mu = 0.f, is = 1.f;
    if (active) {
      if (MODE == 1) {
        mu = mean_invstd[cb + cc];
        is = mean_invstd[c + cb + cc];
      } else {
        mu = a[cb + cc];  // the shift: row 0 of this column
      }
      for (int64_t r = r0 + rr; r < r1; r += rpi) {
        const int64_t idx = r * c + cb + cc;
        if (MODE == 0) {
          const float v = a[idx] - mu;
          acc0 += v;
          acc1 += v * v;
        } else {
          float g = a[idx];
          if (y != nullptr && !(y[idx] > 0.f)) g = 0.f;
          acc0 += g;
          acc1 += g * (x[idx] - mu) * is;
        }
      }
    }
    s0[tid] = acc0;
    s1[tid] = acc1;
    __syncthreads();
    if (tid < cw) {
      float t0 = 0.f, t1 = 0.f;
      for (int q = 0; q < rpi; ++q) {
        t0 += s0[q * cw + tid];
        t1 += s1[q * cw + tid];
      }
      partial[(int64_t)blockIdx.x * 2 * c + cb + tid] = t0;
      partial[(int64_t)blockIdx.x * 2 * c + c + cb + tid] = t1;
    }
    __syncthreads();
  }
}

// Step 2: one block per panel of 32 channels adds the partial rows in a FIXED order (thread (q, ch)
// adds rows q, q + 8, ...; the 8 sub-sums are then added in order of q), in double precision, and
// turns the totals into the layer's statistics.
//   MODE 0: out[0..C) = mean, out[C..2C) = invstd (biased variance); running statistics updated
//   MODE 1: out[0..C) = sum g (= dbias), out[C..2C) = sum g*xhat (= dweight)
template <int MODE>
__global__ __launch_bounds__(kThreads) void col_combine_kernel(
    const float* __restrict__ partial, int nb, int c, const float* __restrict__ x0, int64_t n,
    float eps, float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
    float* __restrict__ out) {
  __shared__ double r0[kThreads];
  __shared__ double r1[kThreads];
  const int tid = threadIdx.x, q = tid >> 5, cc = tid & 31;
  const int ch = blockIdx.x * 32 + cc;
  double a0 = 0.0, a1 = 0.0;
  if (ch < c) {
#pragma unroll 8
    for (int b = q; b < nb; b += 8) {
      a0 += (double)partial[(int64_t)b * 2 * c + ch];
      a1 += (double)partial[(int64_t)b * 2 * c + c + ch];
    }
  }
  r0[tid] = a0;
  r1[tid] = a1;
  __syncthreads();
  if (q != 0 || ch >= c) return;
  double t0 = 0.0, t1 = 0.0;
  for (int k = 0; k < 8; ++k) {
    t0 += r0[k * 32 + cc];
    t1 += r1[k * 32 + cc];
  }
  if (MODE == 0) {
    const double inv_n = 1.0 / (double)n;
    const double d = t0 * inv_n;                 // mean - shift
    const double mean = (double)x0[ch] + d;
    double var = t1 * inv_n - d * d;
    if (var < 0.0) var = 0.0;
    out[ch] = (float)mean;
    out[c + ch] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean != nullptr) {
      const double unbiased = n > 1 ? var * (double)n / (double)(n - 1) : var;
      running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * mean);
      running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * unbiased);
    }
  } else {
    out[ch] = (float)t0;
    out[c + ch] = (float)t1;
  }
}

// y = [relu]( (x - mean) * invstd * w + b [+ residual] )
__global__ __launch_bounds__(kThreads) void bn_apply_kernel(
    const float* __restrict__ x, int64_t total, int c, const float* __restrict__ mean_invstd,
    const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ residual,
    int relu, float* __restrict__ y) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int ch = (int)(i % c);
    float v = (x[i] - mean_invstd[ch]) * mean_invstd[c + ch];
    v = v * (w ? w[ch] : 1.f) + (b ? b[ch] : 0.f);
    if (residual) v += residual[i];
    if (relu && !(v > 0.f)) v = 0.f;
    y[i] = v;
  }
}

// g = dy * mask;  dx = w * invstd * (g - mean(g) - xhat * mean(g * xhat));  dres = g
__global__ __launch_bounds__(kThreads) void bn_backward_apply_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
    const float* __restrict__ mean_invstd, const float* __restrict__ w,
    const float* __restrict__ gsum, int64_t n, int c, float* __restrict__ dx,
    float* __restrict__ dres) {
  const int64_t total = n * c;
  const float inv_n = 1.0f / (float)n;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int ch = (int)(i % c);
    float g = dy[i];
    if (y != nullptr && !(y[i] > 0.f)) g = 0.f;
    const float is = mean_invstd[c + ch];
    const float xh = (x[i] - mean_invstd[ch]) * is;
    const float mg = gsum[ch] * inv_n, mgx = gsum[c + ch] * inv_n;
    dx[i] = (w ? w[ch] : 1.f) * is * (g - mg - xh * mgx);
    if (dres) dres[i] = g;
  }
}

// out[c] += sum_r x[r, c]
__global__ __launch_bounds__(kThreads) void col_sum_kernel(const float* __restrict__ x, int64_t n,
                                                           int c, int64_t rows_per_block,
                                                           float* __restrict__ out) {
  __shared__ float s0[kThreads];
  const int tid = threadIdx.x;
  for (int cb = 0; cb < c; cb += kThreads) {
    const int cw = min(c - cb, kThreads);
    const int rpi = kThreads / cw;
    const int rr = tid / cw, cc = tid % cw;
    float acc = 0.f;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(n, r0 + rows_per_block);
    if (rr < rpi)
      for (int64_t r = r0 + rr; r < r1; r += rpi) acc += x[r * c + cb + cc];
    s0[tid] = acc;
    __syncthreads();
    if (tid < cw) {
      float t = 0.f;
      for (int q = 0; q < rpi; ++q) t += s0[q * cw + tid];
      unsafeAtomicAdd(&out[cb + tid], t);
    }
    __syncthreads();
  }
}

inline int64_t max_reduce_blocks() {
  static const int64_t v = [] {
    const char* e = getenv("PV2_BN_MAXBLOCKS");  // tuning knob (tools/bench_rownorm.py)
    const long x = e ? atol(e) : 0;
    return (int64_t)(x > 0 ? x : 1024);  // measured best of 128..2048, profiles/r01_bn_block_cap.txt
  }();
  return v;
}

inline void reduce_geometry(int64_t n, int c, int* blocks, int64_t* rows_per_block) {
  // enough blocks to cover the 256 CUs several times over (these matrices are only a few MB, the
  // reduction is latency-bound), at least 8 row-iterations per block
  const int rpi = c >= kThreads ? 1 : kThreads / c;
  int64_t want = (n + (int64_t)rpi * 8 - 1) / ((int64_t)rpi * 8);
  if (want < 1) want = 1;
  if (want > max_reduce_blocks()) want = max_reduce_blocks();
  *blocks = (int)want;
  *rows_per_block = (n + want - 1) / want;
}

// partial-sum blocks of the BatchNorm statistics: >= 16 KB of the matrix each, at most 1024 (the
// matrices are a few MB: the reduction is latency-bound and wants every CU)
inline void partial_geometry(int64_t n, int c, int* blocks, int64_t* rows_per_block) {
  const int rpi = c >= kThreads ? 1 : kThreads / c;
  int64_t want = (n * c * 4 + 16383) / 16384;
  if (want < 1) want = 1;
  if (want > kMaxPartialBlocks) want = kMaxPartialBlocks;
  int64_t rpb = (n + want - 1) / want;
  rpb = (rpb + rpi - 1) / rpi * rpi;  // whole iterations
  *rows_per_block = rpb;
  *blocks = (int)((n + rpb - 1) / rpb);
}

}  // namespace

extern "C" {

int64_t pv2_bn_workspace_floats(int c) { return (int64_t)kMaxPartialBlocks * 2 * c; }

int pv2_bn_forward(const float* x, int64_t n, int c, const float* weight, const float* bias,
                   const float* residual, int relu, float eps, float momentum,
                   float* running_mean, float* running_var, float* workspace,
                   float* mean_invstd, float* y, pv2_stream_t stream) {
  PV2_REQUIRE(n >= 1 && c >= 1, "pv2_bn_forward: empty input");
  PV2_REQUIRE(c <= kMaxChannels, "pv2_bn_forward: at most 1024 channels");
  hipStream_t s = (hipStream_t)stream;
  int blocks;
  int64_t rpb;
  partial_geometry(n, c, &blocks, &rpb);
  hipLaunchKernelGGL((col_partials_kernel<0>), dim3(blocks), dim3(kThreads), 0, s, x, nullptr,
                     nullptr, nullptr, n, c, rpb, workspace);
  hipLaunchKernelGGL((col_combine_kernel<0>), dim3((c + 31) / 32), dim3(kThreads), 0, s, workspace,
                     blocks, c, x, n, eps, momentum, running_mean, running_var, mean_invstd);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(pv2::grid_for(n * c, kThreads)), dim3(kThreads), 0, s,
                     x, n * c, c, mean_invstd, weight, bias, residual, relu, y);
  return pv2::check_launch("bn_forward");
}

int pv2_bn_backward(const float* dy, const float* x, const float* y_or_null,
                    const float* mean_invstd, const float* weight, int64_t n, int c,
                    float* workspace, float* gsum, float* dx, float* dresidual_or_null,
                    pv2_stream_t stream) {
  PV2_REQUIRE(n >= 1 && c >= 1, "pv2_bn_backward: empty input");
  PV2_REQUIRE(c <= kMaxChannels, "pv2_bn_backward: at most 1024 channels");
  hipStream_t s = (hipStream_t)stream;
  int blocks;
  int64_t rpb;
  partial_geometry(n, c, &blocks, &rpb);
  hipLaunchKernelGGL((col_partials_kernel<1>), dim3(blocks), dim3(kThreads), 0, s, dy, x, y_or_null,
                     mean_invstd, n, c, rpb, workspace);
  hipLaunchKernelGGL((col_combine_kernel<1>), dim3((c + 31) / 32), dim3(kThreads), 0, s, workspace,
                     blocks, c, nullptr, n, 0.f, 0.f, nullptr, nullptr, gsum);
  hipLaunchKernelGGL(bn_backward_apply_kernel, dim3(pv2::grid_for(n * c, kThreads)),
                     dim3(kThreads), 0, s, dy, x, y_or_null, mean_invstd, weight, gsum, n, c, dx,
                     dresidual_or_null);
  return pv2::check_launch("bn_backward");
}

int pv2_col_sum(const float* x, int64_t n, int c, float* out, pv2_stream_t stream) {
  PV2_REQUIRE(c >= 1 && n >= 0, "pv2_col_sum: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  if (int e = pv2::zero_words(out, c, s)) return e;
  if (n == 0) return PV2_OK;
  int blocks;
  int64_t rpb;
  reduce_geometry(n, c, &blocks, &rpb);
  hipLaunchKernelGGL(col_sum_kernel, dim3(blocks), dim3(kThreads), 0, s, x, n, c, rpb, out);
  return pv2::check_launch("col_sum");
}

}  // extern "C"
