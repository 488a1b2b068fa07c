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

// Element types of the feature matrices: fp32, or the 16-bit storage of the reduced-precision
// training mode (statistics and arithmetic stay fp32; only loads and stores change).
struct F32 { using type = float; };
struct B16 { using type = unsigned short; };  // bfloat16
struct H16 { using type = unsigned short; };  // IEEE half

template <typename E> __device__ __forceinline__ float ld(const typename E::type* p, int64_t i);
template <> __device__ __forceinline__ float ld<F32>(const float* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ float ld<B16>(const unsigned short* p, int64_t i) {
  return __uint_as_float((uint32_t)p[i] << 16);
}
template <> __device__ __forceinline__ float ld<H16>(const unsigned short* p, int64_t i) {
  return (float)__builtin_bit_cast(_Float16, p[i]);
}
template <typename E> __device__ __forceinline__ void st(typename E::type* p, int64_t i, float v);
template <> __device__ __forceinline__ void st<F32>(float* p, int64_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void st<B16>(unsigned short* p, int64_t i, float v) {
  p[i] = __builtin_bit_cast(unsigned short, (__bf16)v);
}
template <> __device__ __forceinline__ void st<H16>(unsigned short* p, int64_t i, float v) {
  p[i] = __builtin_bit_cast(unsigned short, (_Float16)v);
}

// Eight consecutive elements (i % 8 == 0, 16-byte aligned rows: c % 8 == 0) per thread: two 16-byte
// loads for fp32, one for the 16-bit types.
template <typename E> __device__ __forceinline__ void ld8(const typename E::type* p, int64_t i, float (&v)[8]);
template <> __device__ __forceinline__ void ld8<F32>(const float* p, int64_t i, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p + i);
  const float4 b = *reinterpret_cast<const float4*>(p + i + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void ld8<B16>(const unsigned short* p, int64_t i, float (&v)[8]) {
  const uint4 q = *reinterpret_cast<const uint4*>(p + i);
  const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[2 * j] = __uint_as_float(w[j] << 16);
    v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
  }
}
template <> __device__ __forceinline__ void ld8<H16>(const unsigned short* p, int64_t i, float (&v)[8]) {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  const h8 q = *reinterpret_cast<const h8*>(p + i);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (float)q[j];
}
template <typename E> __device__ __forceinline__ void st8(typename E::type* p, int64_t i, const float (&v)[8]);
template <> __device__ __forceinline__ void st8<F32>(float* p, int64_t i, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p + i) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + i + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void st8<B16>(unsigned short* p, int64_t i, const float (&v)[8]) {
  typedef __bf16 b8 __attribute__((ext_vector_type(8)));
  b8 q;
#pragma unroll
  for (int j = 0; j < 8; ++j) q[j] = (__bf16)v[j];
  *reinterpret_cast<b8*>(p + i) = q;
}
template <> __device__ __forceinline__ void st8<H16>(unsigned short* p, int64_t i, const float (&v)[8]) {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  h8 q;
#pragma unroll
  for (int j = 0; j < 8; ++j) q[j] = (_Float16)v[j];
  *reinterpret_cast<h8*>(p + i) = q;
}

// Statistics in two steps without atomics (bitwise reproducible, nothing to clear):
//   1. col_partials_kernel: block b reduces its run of rows and writes partial[b][0..2C);
//   2. col_combine_kernel: one block per 32 channels adds the partial rows in a FIXED order (double
//      precision) and publishes mean / invstd + running statistics, or the bias / weight gradients;
//   3. the elementwise apply kernel.
// sums[0..C) = sum_r f0(r,c);  sums[C..2C) = sum_r f1(r,c)
// MODE 0: f0 = x - s, f1 = (x - s)^2 with the shift s = x[0, c] (forward statistics: the shifted form
//         keeps E[x^2] - mean^2 accurate for columns whose mean is far larger than their spread)
// MODE 1: g = dy * (y > 0 if y else 1);  f0 = g, f1 = g * xhat       (backward reductions)
constexpr int kMaxPartialBlocks = PV2_BN_MAX_PARTIAL_BLOCKS;  // (the conv epilogues write up to this many)
constexpr int kMaxChannels = 1024;

// EA: element type of `a` (the input x in MODE 0, dy in MODE 1); EX / EY: of x and y in MODE 1.
template <int MODE, typename EA, typename EX, typename EY>
__global__ __launch_bounds__(kThreads) void col_partials_kernel(
    const typename EA::type* __restrict__ a, const typename EX::type* __restrict__ x,
    const typename EY::type* __restrict__ y, const float* __restrict__ mean_invstd, int64_t n, int c, int64_t rows_per_block,
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
        mu = ld<EA>(a, cb + cc);  // the shift: row 0 of this column
      }
      for (int64_t r = r0 + rr; r < r1; r += rpi) {
        const int64_t idx = r * c + cb + cc;
        if (MODE == 0) {
          const float v = ld<EA>(a, idx) - mu;
          acc0 += v;
          acc1 += v * v;
        } else {
          float g = ld<EA>(a, idx);
          if (y != nullptr && !(ld<EY>(y, idx) > 0.f)) g = 0.f;
          acc0 += g;
          acc1 += g * (ld<EX>(x, idx) - mu) * is;
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

// The same reduction for c % 8 == 0 (every layer of the backbone): a thread owns 8 consecutive
// channels of every rpi-th row of the block's run - 16-byte loads, two rows in flight per thread.
template <int MODE, typename EA, typename EX, typename EY>
__global__ __launch_bounds__(kThreads) void col_partials_vec_kernel(
    const typename EA::type* __restrict__ a, const typename EX::type* __restrict__ x,
    const typename EY::type* __restrict__ y, const float* __restrict__ mean_invstd, int64_t n, int c,
    int64_t rows_per_block, float* __restrict__ partial) {
  __shared__ double s0[kThreads * 8];
  __shared__ double s1[kThreads * 8];
  const int tid = threadIdx.x;
  const int cg = c >> 3;           // column groups of 8 (<= 128)
  const int rpi = kThreads / cg;   // rows covered per iteration (>= 2)
  const int rr = tid / cg, cc = (tid % cg) * 8;
  const bool active = rr < rpi;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(n, r0 + rows_per_block);
  // (double accumulators: a block's run of rows is summed exactly to within its final rounding; the
  // partial row itself stays fp32 - round 6)
  double acc0[8], acc1[8];
  float mu[8], is[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc0[j] = acc1[j] = 0.0;
  if (active) {
    if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        mu[j] = mean_invstd[cc + j];
        is[j] = mean_invstd[c + cc + j];
      }
    } else {
      ld8<EA>(a, cc, mu);  // the shift: row 0 of these columns
    }
    auto one = [&](int64_t r) {
      const int64_t idx = r * c + cc;
      float va[8];
      ld8<EA>(a, idx, va);
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = va[j] - mu[j];
          acc0[j] += v;
          acc1[j] += v * v;
        }
      } else {
        float vx[8], vy[8];
        ld8<EX>(x, idx, vx);
        if (y != nullptr) ld8<EY>(y, idx, vy);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float g = (y != nullptr && !(vy[j] > 0.f)) ? 0.f : va[j];
          acc0[j] += g;
          acc1[j] += g * (vx[j] - mu[j]) * is[j];
        }
      }
    };
    int64_t r = r0 + rr;
    for (; r + rpi < r1; r += 2 * rpi) {  // two rows per trip: their loads overlap
      one(r);
      one(r + rpi);
    }
    if (r < r1) one(r);
  }
  if (active) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s0[rr * c + cc + j] = acc0[j];
      s1[rr * c + cc + j] = acc1[j];
    }
  }
  __syncthreads();
  for (int t = tid; t < c; t += kThreads) {
    double t0 = 0.0, t1 = 0.0;
    for (int q = 0; q < rpi; ++q) {
      t0 += s0[q * c + t];
      t1 += s1[q * c + t];
    }
    partial[(int64_t)blockIdx.x * 2 * c + t] = (float)t0;
    partial[(int64_t)blockIdx.x * 2 * c + c + t] = (float)t1;
  }
}

// Step 2: one block per panel of 32 channels adds the partial rows in a FIXED order (thread (q, ch)
// adds rows q, q + 32, ...; the 32 sub-sums are then added in order of q), in double precision, and
// turns the totals into the layer's statistics.
//   MODE 0: out[0..C) = mean, out[C..2C) = invstd (biased variance); running statistics updated
//   MODE 1: out[0..C) = sum g (= dbias), out[C..2C) = sum g*xhat (= dweight)
// (kCombineThreads / 32 = 32 sub-sums per channel: with up to 2048 partial rows the loop is a chain of
// dependent loads - 8 sub-sums took 10 us per layer, 118 launches per step)
// Round 6: the panel is 8 channels wide when there are many partial rows (128 sub-sums per channel, four
// times the workgroups: 6.7 -> ~3 us per launch at ~1500 rows; the rows are read in 32-byte pieces).
constexpr int kCombineThreads = 1024;
constexpr int kCombineSubs = kCombineThreads / 32;   // (col_combine_blocks_kernel: 32-channel panels)
template <int MODE, typename EX>
__global__ __launch_bounds__(kCombineThreads) void col_combine_kernel(
    int panel, const float* __restrict__ partial, int nb, int c, const typename EX::type* __restrict__ x0,
    int64_t n, float eps, float momentum, float* __restrict__ running_mean,
    float* __restrict__ running_var, float* __restrict__ out, const float* __restrict__ aff_w = nullptr,
    const float* __restrict__ aff_b = nullptr, float* __restrict__ affine = nullptr,
    int64_t extra_zero_rows = 0, const float* __restrict__ extra_g0 = nullptr, int n_extra_g0 = 0,
    const float* __restrict__ mean_invstd_in = nullptr) {
  __shared__ double r0[kCombineThreads];
  __shared__ double r1[kCombineThreads];
  const int subs = kCombineThreads / panel;   // sub-sums per channel: 32 or 128
  const int tid = threadIdx.x, q = tid / panel, cc = tid % panel;
  const int ch = blockIdx.x * panel + cc;
  double a0 = 0.0, a1 = 0.0;
  if (ch < c) {
#pragma unroll 8
    for (int b = q; b < nb; b += subs) {
      a0 += (double)partial[(int64_t)b * 2 * c + ch];
      a1 += (double)partial[(int64_t)b * 2 * c + c + ch];
    }
  }
  r0[tid] = a0;
  r1[tid] = a1;
  __syncthreads();
  // the sub-sums in order of q, in two steps: eight threads per channel add subs / 8 each, one adds those
  const int per = subs / 8;
  double u0 = 0.0, u1 = 0.0;
  if (q < 8) {
    for (int k = q * per; k < (q + 1) * per; ++k) {
      u0 += r0[k * panel + cc];
      u1 += r1[k * panel + cc];
    }
  }
  __syncthreads();
  if (q < 8) {
    r0[tid] = u0;
    r1[tid] = u1;
  }
  __syncthreads();
  if (q != 0 || ch >= c) return;
  double t0 = 0.0, t1 = 0.0;
  for (int k = 0; k < 8; ++k) {
    t0 += r0[k * panel + cc];
    t1 += r1[k * panel + cc];
  }
  if (MODE == 0) {
    if (extra_zero_rows != 0) {
      // `extra_zero_rows` all-zero rows that were never stored (the empty cells of a dense grid whose
      // occupied cells are the rows of x): each adds (0 - shift) and (0 - shift)^2; n counts them.
      // Negative: x carries that many zero rows MORE than the matrix has (capacity-sized cell arrays)
      const double sft = (double)ld<EX>(x0, ch);
      t0 -= (double)extra_zero_rows * sft;
      t1 += (double)extra_zero_rows * sft * sft;
    }
    const double inv_n = 1.0 / (double)n;
    const double d = t0 * inv_n;                 // mean - shift
    const double mean = (double)ld<EX>(x0, ch) + d;
    double var = t1 * inv_n - d * d;
    if (var < 0.0) var = 0.0;
    out[ch] = (float)mean;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    out[c + ch] = invstd;
    if (affine != nullptr) {  // the normalisation as one multiply-add per element: x * scale + shift
      const float scale = (aff_w ? aff_w[ch] : 1.f) * invstd;
      affine[ch] = scale;
      affine[c + ch] = (aff_b ? aff_b[ch] : 0.f) - (float)mean * scale;
    }
    if (running_mean != nullptr) {
      const double unbiased = n > 1 ? var * (double)n / (double)(n - 1) : var;
      running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * mean);
      running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * unbiased);
    }
  } else {
    if (extra_g0 != nullptr) {
      // the rows that were never stored (see MODE 0) all have xhat = -mean * invstd; the sum of THEIR
      // gradients is (total - stored), with the total handed over in n_extra_g0 ordered pieces
      double g0 = 0.0;
      for (int k = 0; k < n_extra_g0; ++k) g0 += (double)extra_g0[(int64_t)k * c + ch];
      const double xh = -(double)mean_invstd_in[ch] * (double)mean_invstd_in[c + ch];
      t1 += xh * (g0 - t0);
      t0 = g0;
    }
    out[ch] = (float)t0;
    out[c + ch] = (float)t1;
  }
}

template <int MODE, typename EX, typename... A>
inline void launch_combine(hipStream_t s, int c, const float* partial, int nb, A... rest) {
  static const int forced = [] {
    const char* e = getenv("PV2_BN_COMBINE_PANEL");   // 8 / 32: one panel width for every launch (A / B)
    return e ? atoi(e) : 0;
  }();
  const int panel = forced == 8 || forced == 32 ? forced : (nb >= 256 ? 8 : 32);
  hipLaunchKernelGGL((col_combine_kernel<MODE, EX>), dim3((c + panel - 1) / panel), dim3(kCombineThreads), 0,
                     s, panel, partial, nb, rest...);
}

// The forward statistics from PER-BLOCK moments (the epilogue of spconv_osm_kernel): block b of
// `rows_per_block` rows (the last one shorter) wrote partial[b][0..C) = sum_r x and partial[b][C..2C) =
// sum_r (x - mean_b)^2.  Chan's pairwise update, all blocks against the grand mean, in double and in a
// fixed order:  M2 = sum_b [ M2_b + n_b (mean_b - mean)^2 ].  No shift needed: every block is centred
// on its own mean.  Publishes what col_combine_kernel<0> publishes.
__global__ __launch_bounds__(kCombineThreads) void col_combine_blocks_kernel(
    const float* __restrict__ partial, int nb, int c, int64_t n, int64_t rows_per_block, float eps,
    float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
    float* __restrict__ out) {
  __shared__ double r0[kCombineThreads];
  __shared__ double s_mean[32];
  const int tid = threadIdx.x, q = tid >> 5, cc = tid & 31;
  const int ch = blockIdx.x * 32 + cc;
  double a0 = 0.0;
  if (ch < c) {
#pragma unroll 8
    for (int b = q; b < nb; b += kCombineSubs) a0 += (double)partial[(int64_t)b * 2 * c + ch];
  }
  r0[tid] = a0;
  __syncthreads();
  if (q == 0) {
    double t0 = 0.0;
    for (int k = 0; k < kCombineSubs; ++k) t0 += r0[k * 32 + cc];
    s_mean[cc] = t0 / (double)n;
  }
  __syncthreads();
  const double mean = s_mean[cc];
  double a1 = 0.0;
  if (ch < c) {
#pragma unroll 4
    for (int b = q; b < nb; b += kCombineSubs) {
      const int64_t left = n - (int64_t)b * rows_per_block;
      const double nbk = (double)(left < rows_per_block ? left : rows_per_block);
      const double d = (double)partial[(int64_t)b * 2 * c + ch] / nbk - mean;
      a1 += (double)partial[(int64_t)b * 2 * c + c + ch] + nbk * d * d;
    }
  }
  r0[tid] = a1;
  __syncthreads();
  if (q != 0 || ch >= c) return;
  double t1 = 0.0;
  for (int k = 0; k < kCombineSubs; ++k) t1 += r0[k * 32 + cc];
  double var = t1 / (double)n;
  if (var < 0.0) var = 0.0;
  out[ch] = (float)mean;
  out[c + ch] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean != nullptr) {
    const double unbiased = n > 1 ? var * (double)n / (double)(n - 1) : var;
    running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * mean);
    running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * unbiased);
  }
}

// y = [relu]( (x - mean) * invstd * w + b [+ residual] )
template <typename EX, typename EY>
__global__ __launch_bounds__(kThreads) void bn_apply_kernel(
    const typename EX::type* __restrict__ x, int64_t total, int c,
    const float* __restrict__ mean_invstd, const float* __restrict__ w, const float* __restrict__ b,
    const typename EY::type* __restrict__ residual, int relu, typename EY::type* __restrict__ y) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int ch = (int)(i % c);
    float v = (ld<EX>(x, i) - mean_invstd[ch]) * mean_invstd[c + ch];
    v = v * (w ? w[ch] : 1.f) + (b ? b[ch] : 0.f);
    if (residual) v += ld<EY>(residual, i);
    if (relu && !(v > 0.f)) v = 0.f;
    st<EY>(y, i, v);
  }
}

// g = dy * mask;  dx = w * invstd * (g - mean(g) - xhat * mean(g * xhat));  dres = g
template <typename EX, typename EY>
__global__ __launch_bounds__(kThreads) void bn_backward_apply_kernel(
    const typename EY::type* __restrict__ dy, const typename EX::type* __restrict__ x,
    const typename EY::type* __restrict__ y, const float* __restrict__ mean_invstd,
    const float* __restrict__ w, const float* __restrict__ gsum, int64_t n, int c,
    typename EX::type* __restrict__ dx, typename EY::type* __restrict__ dres,
    int64_t n_norm = 0) {
  const int64_t total = n * c;
  const float inv_n = 1.0f / (float)(n_norm > 0 ? n_norm : n);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int ch = (int)(i % c);
    float g = ld<EY>(dy, i);
    if (y != nullptr && !(ld<EY>(y, i) > 0.f)) g = 0.f;
    const float is = mean_invstd[c + ch];
    const float xh = (ld<EX>(x, i) - mean_invstd[ch]) * is;
    const float mg = gsum[ch] * inv_n, mgx = gsum[c + ch] * inv_n;
    st<EX>(dx, i, (w ? w[ch] : 1.f) * is * (g - mg - xh * mgx));
    if (dres) st<EY>(dres, i, g);
  }
}

// 8 elements per thread (c % 8 == 0)
template <typename EX, typename EY>
__global__ __launch_bounds__(kThreads) void bn_apply_vec_kernel(
    const typename EX::type* __restrict__ x, int64_t total, int c,
    const float* __restrict__ mean_invstd, const float* __restrict__ w, const float* __restrict__ b,
    const typename EY::type* __restrict__ residual, int relu, typename EY::type* __restrict__ y) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < total; i += stride) {
    const int ch = (int)(i % c);
    float v[8], rs[8], mu[8], is[8], wv[8], bv[8];
    ld8<EX>(x, i, v);
    if (residual) ld8<EY>(residual, i, rs);
    ld8<F32>(mean_invstd, ch, mu);
    ld8<F32>(mean_invstd, c + ch, is);
    if (w) ld8<F32>(w, ch, wv);
    if (b) ld8<F32>(b, ch, bv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = (v[j] - mu[j]) * is[j];
      t = t * (w ? wv[j] : 1.f) + (b ? bv[j] : 0.f);
      if (residual) t += rs[j];
      if (relu && !(t > 0.f)) t = 0.f;
      v[j] = t;
    }
    st8<EY>(y, i, v);
  }
}

template <typename EX, typename EY>
__global__ __launch_bounds__(kThreads) void bn_backward_apply_vec_kernel(
    const typename EY::type* __restrict__ dy, const typename EX::type* __restrict__ x,
    const typename EY::type* __restrict__ y, const float* __restrict__ mean_invstd,
    const float* __restrict__ w, const float* __restrict__ gsum, int64_t n, int c,
    typename EX::type* __restrict__ dx, typename EY::type* __restrict__ dres,
    int64_t n_norm = 0) {
  const int64_t total = n * c;
  const float inv_n = 1.0f / (float)(n_norm > 0 ? n_norm : n);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < total; i += stride) {
    const int ch = (int)(i % c);
    float g[8], vx[8], vy[8], o[8], mu[8], is[8], wv[8], s0[8], s1[8];
    ld8<EY>(dy, i, g);
    ld8<EX>(x, i, vx);
    if (y != nullptr) ld8<EY>(y, i, vy);
    ld8<F32>(mean_invstd, ch, mu);
    ld8<F32>(mean_invstd, c + ch, is);
    ld8<F32>(gsum, ch, s0);
    ld8<F32>(gsum, c + ch, s1);
    if (w) ld8<F32>(w, ch, wv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (y != nullptr && !(vy[j] > 0.f)) g[j] = 0.f;
      const float xh = (vx[j] - mu[j]) * is[j];
      o[j] = (w ? wv[j] : 1.f) * is[j] * (g[j] - s0[j] * inv_n - xh * s1[j] * inv_n);
    }
    st8<EX>(dx, i, o);
    if (dres) st8<EY>(dres, i, g);
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
  if (want > 1024) want = 1024;   // (measured best for the stand-alone partials kernel)
  int64_t rpb = (n + want - 1) / want;
  rpb = (rpb + rpi - 1) / rpi * rpi;  // whole iterations
  *rows_per_block = rpb;
  *blocks = (int)((n + rpb - 1) / rpb);
}

}  // namespace

extern "C" {

int64_t pv2_bn_workspace_floats(int c) { return (int64_t)kMaxPartialBlocks * 2 * c; }

}  // extern "C"

namespace pv2 {

void bn_partial_geometry(int64_t n, int c, int* blocks, int64_t* rows_per_block) {
  partial_geometry(n, c, blocks, rows_per_block);
}

int bn_apply(const float* x, int64_t n, int c, const float* mean_invstd, const float* weight,
             const float* bias, const float* residual, int relu, float* y, hipStream_t s) {
  if ((c % 8) == 0)
    hipLaunchKernelGGL((bn_apply_vec_kernel<F32, F32>), dim3(pv2::grid_for(n * c / 8, kThreads)),
                       dim3(kThreads), 0, s, x, n * c, c, mean_invstd, weight, bias, residual, relu,
                       y);
  else
    hipLaunchKernelGGL((bn_apply_kernel<F32, F32>), dim3(pv2::grid_for(n * c, kThreads)),
                       dim3(kThreads), 0, s, x, n * c, c, mean_invstd, weight, bias, residual, relu,
                       y);
  return pv2::check_launch("bn_apply");
}

// The two remaining steps of the BatchNorm backward when the per-block partial sums of {g, g * xhat} were
// written by the epilogue of the grad-input row reduce that completed dy (sparse_conv_pr.hip STATS == 2):
// the ordered combine into gsum, and - when the unit's turn comes - the elementwise pass.
int bn_backward_combine(const float* partial, int blocks, int c, float* gsum, hipStream_t s) {
  launch_combine<1, F32>(s, c, partial, blocks, c, (const float*)nullptr, (int64_t)0, 0.f, 0.f, nullptr, nullptr,
                     gsum);
  return pv2::check_launch("bn_backward_combine");
}

int bn_backward_apply(const float* dy, const float* x, const float* y_or_null, const float* mean_invstd,
                      const float* weight, const float* gsum, int64_t n, int c, float* dx, float* dres,
                      hipStream_t s) {
  if ((c % 8) == 0)
    hipLaunchKernelGGL((bn_backward_apply_vec_kernel<F32, F32>),
                       dim3(pv2::grid_for(n * c / 8, kThreads)), dim3(kThreads), 0, s, dy, x, y_or_null,
                       mean_invstd, weight, gsum, n, c, dx, dres, (int64_t)0);
  else
    hipLaunchKernelGGL((bn_backward_apply_kernel<F32, F32>), dim3(pv2::grid_for(n * c, kThreads)),
                       dim3(kThreads), 0, s, dy, x, y_or_null, mean_invstd, weight, gsum, n, c, dx, dres,
                       (int64_t)0);
  return pv2::check_launch("bn_backward_apply");
}

int bn_forward_from_partials(const float* x, int64_t n, int c, const float* partial, int blocks,
                             const float* weight, const float* bias, const float* residual,
                             int relu, float eps, float momentum, float* running_mean,
                             float* running_var, float* mean_invstd, float* y, hipStream_t s) {
  launch_combine<0, F32>(s, c, partial, blocks, c, x, n, eps, momentum, running_mean, running_var,
                     mean_invstd);
  if ((c % 8) == 0)
    hipLaunchKernelGGL((bn_apply_vec_kernel<F32, F32>), dim3(pv2::grid_for(n * c / 8, kThreads)),
                       dim3(kThreads), 0, s, x, n * c, c, mean_invstd, weight, bias, residual, relu,
                       y);
  else
    hipLaunchKernelGGL((bn_apply_kernel<F32, F32>), dim3(pv2::grid_for(n * c, kThreads)),
                       dim3(kThreads), 0, s, x, n * c, c, mean_invstd, weight, bias, residual, relu,
                       y);
  return pv2::check_launch("bn_forward_from_partials");
}

// the same from per-block moments (spconv_osm_kernel's epilogue): combine + apply
int bn_forward_from_block_stats(const float* x, int64_t n, int c, const float* partial, int blocks,
                                int64_t rows_per_block, const float* weight, const float* bias,
                                const float* residual, int relu, float eps, float momentum,
                                float* running_mean, float* running_var, float* mean_invstd, float* y,
                                hipStream_t s) {
  hipLaunchKernelGGL(col_combine_blocks_kernel, dim3((c + 31) / 32), dim3(kCombineThreads), 0, s, partial,
                     blocks, c, n, rows_per_block, eps, momentum, running_mean, running_var, mean_invstd);
  if ((c % 8) == 0)
    hipLaunchKernelGGL((bn_apply_vec_kernel<F32, F32>), dim3(pv2::grid_for(n * c / 8, kThreads)),
                       dim3(kThreads), 0, s, x, n * c, c, mean_invstd, weight, bias, residual, relu,
                       y);
  else
    hipLaunchKernelGGL((bn_apply_kernel<F32, F32>), dim3(pv2::grid_for(n * c, kThreads)),
                       dim3(kThreads), 0, s, x, n * c, c, mean_invstd, weight, bias, residual, relu,
                       y);
  return pv2::check_launch("bn_forward_from_block_stats");
}

}  // namespace pv2

namespace {

template <typename EX, typename EY>
int bn_forward_t(const void* x, int64_t n, int c, const float* weight, const float* bias,
                 const void* residual, int relu, float eps, float momentum, float* running_mean,
                 float* running_var, float* workspace, float* mean_invstd, void* y, hipStream_t s) {
  using TX = typename EX::type;
  using TY = typename EY::type;
  int blocks;
  int64_t rpb;
  partial_geometry(n, c, &blocks, &rpb);
  const bool vec = (c % 8) == 0;  // 16-byte pieces of rows
  if (vec)
    hipLaunchKernelGGL((col_partials_vec_kernel<0, EX, EX, EY>), dim3(blocks), dim3(kThreads), 0, s,
                       (const TX*)x, (const TX*)nullptr, (const TY*)nullptr, nullptr, n, c, rpb,
                       workspace);
  else
    hipLaunchKernelGGL((col_partials_kernel<0, EX, EX, EY>), dim3(blocks), dim3(kThreads), 0, s,
                       (const TX*)x, (const TX*)nullptr, (const TY*)nullptr, nullptr, n, c, rpb,
                       workspace);
  launch_combine<0, EX>(s, c, workspace, blocks, c, (const TX*)x, n, eps, momentum, running_mean, running_var,
                     mean_invstd);
  if (vec)
    hipLaunchKernelGGL((bn_apply_vec_kernel<EX, EY>), dim3(pv2::grid_for(n * c / 8, kThreads)),
                       dim3(kThreads), 0, s, (const TX*)x, n * c, c, mean_invstd, weight, bias,
                       (const TY*)residual, relu, (TY*)y);
  else
    hipLaunchKernelGGL((bn_apply_kernel<EX, EY>), dim3(pv2::grid_for(n * c, kThreads)),
                       dim3(kThreads), 0, s, (const TX*)x, n * c, c, mean_invstd, weight, bias,
                       (const TY*)residual, relu, (TY*)y);
  return pv2::check_launch("bn_forward");
}

template <typename EX, typename EY>
int bn_backward_t(const void* dy, const void* x, const void* y_or_null, const float* mean_invstd,
                  const float* weight, int64_t n, int c, float* workspace, float* gsum, void* dx,
                  void* dres, hipStream_t s) {
  using TX = typename EX::type;
  using TY = typename EY::type;
  int blocks;
  int64_t rpb;
  partial_geometry(n, c, &blocks, &rpb);
  const bool vec = (c % 8) == 0;
  if (vec)
    hipLaunchKernelGGL((col_partials_vec_kernel<1, EY, EX, EY>), dim3(blocks), dim3(kThreads), 0, s,
                       (const TY*)dy, (const TX*)x, (const TY*)y_or_null, mean_invstd, n, c, rpb,
                       workspace);
  else
    hipLaunchKernelGGL((col_partials_kernel<1, EY, EX, EY>), dim3(blocks), dim3(kThreads), 0, s,
                       (const TY*)dy, (const TX*)x, (const TY*)y_or_null, mean_invstd, n, c, rpb,
                       workspace);
  launch_combine<1, F32>(s, c, workspace, blocks, c, (const float*)nullptr, n, 0.f, 0.f, nullptr, nullptr, gsum);
  if (vec)
    hipLaunchKernelGGL((bn_backward_apply_vec_kernel<EX, EY>),
                       dim3(pv2::grid_for(n * c / 8, kThreads)), dim3(kThreads), 0, s, (const TY*)dy,
                       (const TX*)x, (const TY*)y_or_null, mean_invstd, weight, gsum, n, c, (TX*)dx,
                       (TY*)dres);
  else
    hipLaunchKernelGGL((bn_backward_apply_kernel<EX, EY>), dim3(pv2::grid_for(n * c, kThreads)),
                       dim3(kThreads), 0, s, (const TY*)dy, (const TX*)x, (const TY*)y_or_null,
                       mean_invstd, weight, gsum, n, c, (TX*)dx, (TY*)dres);
  return pv2::check_launch("bn_backward");
}

}  // namespace

extern "C" {

// (x dtype, y dtype) pairs the sparse backbone produces: all-fp32; 16-bit throughout; and the
// fp32 -> 16-bit boundary after the stem conv (whose 6-channel input stays fp32).
#define PV2_BN_DISPATCH(FN, ...)                                                        \
  do {                                                                                  \
    if (x_dtype == PV2_F32 && y_dtype == PV2_F32) return FN<F32, F32>(__VA_ARGS__);     \
    if (x_dtype == PV2_F32 && y_dtype == PV2_BF16) return FN<F32, B16>(__VA_ARGS__);    \
    if (x_dtype == PV2_F32 && y_dtype == PV2_F16) return FN<F32, H16>(__VA_ARGS__);     \
    if (x_dtype == PV2_BF16 && y_dtype == PV2_BF16) return FN<B16, B16>(__VA_ARGS__);   \
    if (x_dtype == PV2_F16 && y_dtype == PV2_F16) return FN<H16, H16>(__VA_ARGS__);     \
    pv2::set_error("pv2_bn: unsupported (x, y) dtype pair");                            \
    return PV2_E_UNSUPPORTED;                                                           \
  } while (0)

int pv2_bn_forward_mixed(const void* x, int x_dtype, int64_t n, int c, const float* weight,
                         const float* bias, const void* residual, int relu, float eps,
                         float momentum, float* running_mean, float* running_var,
                         float* workspace, float* mean_invstd, void* y, int y_dtype,
                         pv2_stream_t stream) {
  PV2_REQUIRE(n >= 1 && c >= 1, "pv2_bn_forward: empty input");
  PV2_REQUIRE(c <= kMaxChannels, "pv2_bn_forward: at most 1024 channels");
  hipStream_t s = (hipStream_t)stream;
  PV2_BN_DISPATCH(bn_forward_t, x, n, c, weight, bias, residual, relu, eps, momentum, running_mean,
                  running_var, workspace, mean_invstd, y, s);
}

int pv2_bn_backward_mixed(const void* dy, const void* x, int x_dtype, const void* y_or_null,
                          int y_dtype, const float* mean_invstd, const float* weight, int64_t n,
                          int c, float* workspace, float* gsum, void* dx,
                          void* dresidual_or_null, pv2_stream_t stream) {
  PV2_REQUIRE(n >= 1 && c >= 1, "pv2_bn_backward: empty input");
  PV2_REQUIRE(c <= kMaxChannels, "pv2_bn_backward: at most 1024 channels");
  hipStream_t s = (hipStream_t)stream;
  PV2_BN_DISPATCH(bn_backward_t, dy, x, y_or_null, mean_invstd, weight, n, c, workspace, gsum, dx,
                  dresidual_or_null, s);
}

int pv2_bn_forward(const float* x, int64_t n, int c, const float* weight, const float* bias,
                   const float* residual, int relu, float eps, float momentum,
                   float* running_mean, float* running_var, float* workspace,
                   float* mean_invstd, float* y, pv2_stream_t stream) {
  return pv2_bn_forward_mixed(x, PV2_F32, n, c, weight, bias, residual, relu, eps, momentum,
                              running_mean, running_var, workspace, mean_invstd, y, PV2_F32, stream);
}

int pv2_bn_backward(const float* dy, const float* x, const float* y_or_null,
                    const float* mean_invstd, const float* weight, int64_t n, int c,
                    float* workspace, float* gsum, float* dx, float* dresidual_or_null,
                    pv2_stream_t stream) {
  return pv2_bn_backward_mixed(dy, x, PV2_F32, y_or_null, PV2_F32, mean_invstd, weight, n, c,
                               workspace, gsum, dx, dresidual_or_null, stream);
}

// Training-mode BatchNorm statistics of the rows of x WITHOUT applying them: mean / invstd, the
// running statistics, and the affine pair (scale = weight * invstd, shift = bias - mean * scale) a
// consumer folds into its own load path (the dense convolutions of csrc/dense_conv.hip).
int pv2_bn_statistics(const float* x, int64_t n, int c, const float* weight, const float* bias,
                      float eps, float momentum, float* running_mean, float* running_var,
                      float* workspace, float* mean_invstd, float* affine, pv2_stream_t stream) {
  PV2_REQUIRE(x != nullptr && workspace != nullptr && mean_invstd != nullptr && affine != nullptr,
              "bn_statistics: null pointer");
  PV2_REQUIRE(n > 0 && c > 0 && c <= kMaxChannels, "bn_statistics: bad shape");
  hipStream_t s = (hipStream_t)stream;
  int blocks;
  int64_t rpb;
  partial_geometry(n, c, &blocks, &rpb);
  if ((c % 8) == 0)
    hipLaunchKernelGGL((col_partials_vec_kernel<0, F32, F32, F32>), dim3(blocks), dim3(kThreads), 0, s,
                       x, (const float*)nullptr, (const float*)nullptr, nullptr, n, c, rpb, workspace);
  else
    hipLaunchKernelGGL((col_partials_kernel<0, F32, F32, F32>), dim3(blocks), dim3(kThreads), 0, s,
                       x, (const float*)nullptr, (const float*)nullptr, nullptr, n, c, rpb, workspace);
  launch_combine<0, F32>(s, c, workspace, blocks, c, x, n, eps, momentum, running_mean, running_var, mean_invstd,
                     weight, bias, affine);
  return pv2::check_launch("bn_statistics");
}

// The same statistics for a matrix whose trailing `extra_zero_rows` rows are all zero and were never
// stored: x holds the occupied cells of a mostly empty dense grid (n_rows of them), the BatchNorm3d in
// front of the projection network's first convolution normalises over every cell
// (ponder/models/ponder/sparse_input.py; reference ponder_indoor_base.py:177-342 builds the grid).
int pv2_bn_statistics_padded(const float* x, int64_t n_rows, int64_t extra_zero_rows, int c,
                             const float* weight, const float* bias, float eps, float momentum,
                             float* running_mean, float* running_var, float* workspace,
                             float* mean_invstd, float* affine, pv2_stream_t stream) {
  PV2_REQUIRE(x != nullptr && workspace != nullptr && mean_invstd != nullptr && affine != nullptr,
              "bn_statistics_padded: null pointer");
  PV2_REQUIRE(n_rows > 0 && n_rows + extra_zero_rows > 0 && c > 0 && c <= kMaxChannels,
              "bn_statistics_padded: bad shape");
  hipStream_t s = (hipStream_t)stream;
  int blocks;
  int64_t rpb;
  partial_geometry(n_rows, c, &blocks, &rpb);
  if ((c % 8) == 0)
    hipLaunchKernelGGL((col_partials_vec_kernel<0, F32, F32, F32>), dim3(blocks), dim3(kThreads), 0, s,
                       x, (const float*)nullptr, (const float*)nullptr, nullptr, n_rows, c, rpb,
                       workspace);
  else
    hipLaunchKernelGGL((col_partials_kernel<0, F32, F32, F32>), dim3(blocks), dim3(kThreads), 0, s,
                       x, (const float*)nullptr, (const float*)nullptr, nullptr, n_rows, c, rpb,
                       workspace);
  launch_combine<0, F32>(s, c, workspace, blocks, c, x, n_rows + extra_zero_rows, eps, momentum, running_mean,
                     running_var, mean_invstd, weight, bias, affine, extra_zero_rows);
  return pv2::check_launch("bn_statistics_padded");
}

// Backward of that normalisation.  dy / x: the stored rows; the gradient rows of the rows that were
// never stored are known only by their column sums TOTAL over all rows (stored + not stored), handed
// over as `n_total_parts` ordered pieces total_parts[k][c] (added in order of k).
// gsum[0..c) = d bias, gsum[c..2c) = d weight; dx for the stored rows.
int pv2_bn_backward_padded(const float* dy, const float* x, int64_t n_rows, int64_t extra_zero_rows,
                           int c, const float* mean_invstd, const float* weight,
                           const float* total_parts, int n_total_parts, float* workspace,
                           float* gsum, float* dx, pv2_stream_t stream) {
  PV2_REQUIRE(dy != nullptr && x != nullptr && mean_invstd != nullptr && total_parts != nullptr &&
                  workspace != nullptr && gsum != nullptr && dx != nullptr,
              "bn_backward_padded: null pointer");
  PV2_REQUIRE(n_rows > 0 && n_rows + extra_zero_rows > 0 && c > 0 && c <= kMaxChannels && n_total_parts > 0,
              "bn_backward_padded: bad shape");
  hipStream_t s = (hipStream_t)stream;
  int blocks;
  int64_t rpb;
  partial_geometry(n_rows, c, &blocks, &rpb);
  const bool vec = (c % 8) == 0;
  if (vec)
    hipLaunchKernelGGL((col_partials_vec_kernel<1, F32, F32, F32>), dim3(blocks), dim3(kThreads), 0, s,
                       dy, x, (const float*)nullptr, mean_invstd, n_rows, c, rpb, workspace);
  else
    hipLaunchKernelGGL((col_partials_kernel<1, F32, F32, F32>), dim3(blocks), dim3(kThreads), 0, s,
                       dy, x, (const float*)nullptr, mean_invstd, n_rows, c, rpb, workspace);
  launch_combine<1, F32>(s, c, workspace, blocks, c, (const float*)nullptr, n_rows, 0.f, 0.f, nullptr, nullptr,
                     gsum, nullptr, nullptr, nullptr, (int64_t)0, total_parts, n_total_parts,
                     mean_invstd);
  const int64_t n_norm = n_rows + extra_zero_rows;
  if (vec)
    hipLaunchKernelGGL((bn_backward_apply_vec_kernel<F32, F32>),
                       dim3(pv2::grid_for(n_rows * c / 8, kThreads)), dim3(kThreads), 0, s, dy, x,
                       (const float*)nullptr, mean_invstd, weight, gsum, n_rows, c, dx,
                       (float*)nullptr, n_norm);
  else
    hipLaunchKernelGGL((bn_backward_apply_kernel<F32, F32>), dim3(pv2::grid_for(n_rows * c, kThreads)),
                       dim3(kThreads), 0, s, dy, x, (const float*)nullptr, mean_invstd, weight, gsum,
                       n_rows, c, dx, (float*)nullptr, n_norm);
  return pv2::check_launch("bn_backward_padded");
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
