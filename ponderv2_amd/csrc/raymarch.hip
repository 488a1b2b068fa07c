// Alpha compositing along rays for gfx950: NeuS weights from alphas and weighted sums of per-sample
// quantities (colour, depth, normal, semantic features), forward and backward.
//
// Stands in for the elementwise / cumprod / reduction chain behind
// ponder/models/ponder/render_utils/rays.py:83-105 (get_weights_and_transmittance_from_alphas) and
// renderers.py:5-75 (RGB / Depth / Normal / Semantic renderers: sum_s w_s * value_s), which costs
// ~40 small launches forward and ~80 backward per step in the stock path.
//
//   T_s = prod_{j<s} (1 - alpha_j + 1e-7)        w_s = alpha_s * T_s          (transmittance, weight)
//   out_f = sum_s w_s * x_{s,f}
//   d alpha_s = gw_s * T_s - (sum_{k>s} gw_k * w_k) / (1 - alpha_s + 1e-7)
//   d x_{s,f} = w_s * gout_f                      d w_s = sum_f x_{s,f} * gout_f
//
// A ray's S samples (132 / 96 here, <= 256) are a few hundred bytes: one wave owns a ray for the
// weight kernels (each lane a run of consecutive samples, products / suffix sums combined with
// wave scans), one workgroup owns a ray for the weighted sums.  All HBM-bound: every operand is
// read once and every result written once.
#include "common.h"

namespace {

constexpr int kMaxPerLane = 4;  // S <= 256

__device__ __forceinline__ float wave_excl_prod(float v, int lane, float* total) {
  float incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float up = __shfl_up(incl, off);
    if (lane >= off) incl *= up;
  }
  *total = __shfl(incl, 63);
  const float prev = __shfl_up(incl, 1);
  return lane == 0 ? 1.f : prev;
}

// exclusive SUFFIX sum: result for lane l = sum of v over lanes > l
__device__ __forceinline__ float wave_excl_suffix_sum(float v, int lane) {
  float incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float dn = __shfl_down(incl, off);
    if (lane + off < 64) incl += dn;
  }
  const float next = __shfl_down(incl, 1);
  return lane == 63 ? 0.f : next;
}

// one wave per ray; lane l owns samples [l*per, (l+1)*per)
__global__ __launch_bounds__(256) void weights_fwd_kernel(const float* __restrict__ alpha,
                                                          int64_t n_rays, int S,
                                                          float* __restrict__ weights,
                                                          float* __restrict__ trans) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const int per = (S + 63) / 64;
  const float* a_row = alpha + ray * S;
  float a[kMaxPerLane], x[kMaxPerLane];
  float prod = 1.f;
#pragma unroll
  for (int j = 0; j < kMaxPerLane; ++j) {
    const int s = lane * per + j;
    const bool on = j < per && s < S;
    a[j] = on ? a_row[s] : 0.f;
    x[j] = on ? 1.f - a[j] + 1e-7f : 1.f;
    prod *= x[j];
  }
  float total;
  float T = wave_excl_prod(prod, lane, &total);
#pragma unroll
  for (int j = 0; j < kMaxPerLane; ++j) {
    const int s = lane * per + j;
    if (j < per && s < S) {
      weights[ray * S + s] = a[j] * T;
      if (trans) trans[ray * (S + 1) + s] = T;
    }
    T *= x[j];
  }
  if (trans && lane == 63) trans[ray * (S + 1) + S] = total;
}

__global__ __launch_bounds__(256) void weights_bwd_kernel(const float* __restrict__ alpha,
                                                          const float* __restrict__ gw,
                                                          int64_t n_rays, int S,
                                                          float* __restrict__ galpha) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const int per = (S + 63) / 64;
  const float* a_row = alpha + ray * S;
  const float* g_row = gw + ray * S;
  float a[kMaxPerLane], x[kMaxPerLane], g[kMaxPerLane], T_s[kMaxPerLane], q[kMaxPerLane];
  float prod = 1.f;
#pragma unroll
  for (int j = 0; j < kMaxPerLane; ++j) {
    const int s = lane * per + j;
    const bool on = j < per && s < S;
    a[j] = on ? a_row[s] : 0.f;
    g[j] = on ? g_row[s] : 0.f;
    x[j] = on ? 1.f - a[j] + 1e-7f : 1.f;
    prod *= x[j];
  }
  float total;
  float T = wave_excl_prod(prod, lane, &total);
  float lane_q = 0.f;  // sum over this lane's samples of gw * w
#pragma unroll
  for (int j = 0; j < kMaxPerLane; ++j) {
    T_s[j] = T;
    q[j] = g[j] * a[j] * T;
    lane_q += q[j];
    T *= x[j];
  }
  float after = wave_excl_suffix_sum(lane_q, lane);  // everything owned by later lanes
#pragma unroll
  for (int j = kMaxPerLane - 1; j >= 0; --j) {
    const int s = lane * per + j;
    if (j < per && s < S) galpha[ray * S + s] = g[j] * T_s[j] - after / x[j];
    after += q[j];
  }
}

// out[r, f] = sum_s w[r, s] * x[r, s, f].  One workgroup per ray.
// WIDE (F >= 32): a wave walks samples s = wave, wave+4, ..., lanes stride the channels; the four
// waves' partial rows meet in LDS.  NARROW (F < 32): threads stride the samples, every channel is
// block-reduced.
constexpr int kMaxWideF = 512;

__global__ __launch_bounds__(256) void accumulate_fwd_kernel(const float* __restrict__ w,
                                                             const float* __restrict__ x,
                                                             int S, int F,
                                                             float* __restrict__ out) {
  __shared__ float part[4][kMaxWideF];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t ray = blockIdx.x;
  const float* w_row = w + ray * S;
  const float* x_ray = x + ray * (int64_t)S * F;
  if (F >= 32) {
    float acc[kMaxWideF / 64];
#pragma unroll
    for (int u = 0; u < kMaxWideF / 64; ++u) acc[u] = 0.f;
    for (int s = wave; s < S; s += 4) {
      const float ws = w_row[s];
      const float* xs = x_ray + (int64_t)s * F;
#pragma unroll
      for (int u = 0; u < kMaxWideF / 64; ++u) {
        const int f = lane + 64 * u;
        if (f < F) acc[u] += ws * xs[f];
      }
    }
#pragma unroll
    for (int u = 0; u < kMaxWideF / 64; ++u) {
      const int f = lane + 64 * u;
      if (f < F) part[wave][f] = acc[u];
    }
    __syncthreads();
    for (int f = tid; f < F; f += 256)
      out[ray * F + f] = part[0][f] + part[1][f] + part[2][f] + part[3][f];
  } else {
    for (int f = 0; f < F; ++f) {
      float p = 0.f;
      for (int s = tid; s < S; s += 256) p += w_row[s] * x_ray[(int64_t)s * F + f];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o);
      if (lane == 0) part[wave][f] = p;
    }
    __syncthreads();
    if (tid < F) out[ray * F + tid] = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
  }
}

// gx[r,s,f] = w[r,s] * gout[r,f];  gw[r,s] = sum_f x[r,s,f] * gout[r,f]   (either may be null)
__global__ __launch_bounds__(256) void accumulate_bwd_kernel(const float* __restrict__ w,
                                                             const float* __restrict__ x,
                                                             const float* __restrict__ gout,
                                                             int S, int F,
                                                             float* __restrict__ gw,
                                                             float* __restrict__ gx) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t ray = blockIdx.x;
  const float* w_row = w + ray * S;
  const float* x_ray = x + ray * (int64_t)S * F;
  const float* go = gout + ray * F;
  float* gx_ray = gx ? gx + ray * (int64_t)S * F : nullptr;
  if (F >= 32) {
    float g[kMaxWideF / 64];
#pragma unroll
    for (int u = 0; u < kMaxWideF / 64; ++u) {
      const int f = lane + 64 * u;
      g[u] = f < F ? go[f] : 0.f;
    }
    for (int s = wave; s < S; s += 4) {
      const float ws = w_row[s];
      const float* xs = x_ray + (int64_t)s * F;
      float p = 0.f;
#pragma unroll
      for (int u = 0; u < kMaxWideF / 64; ++u) {
        const int f = lane + 64 * u;
        if (f < F) {
          if (gw) p += xs[f] * g[u];
          if (gx_ray) gx_ray[(int64_t)s * F + f] = ws * g[u];
        }
      }
      if (gw) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o);
        if (lane == 0) gw[ray * S + s] = p;
      }
    }
  } else {
    for (int s = tid; s < S; s += 256) {
      const float ws = w_row[s];
      float p = 0.f;
      for (int f = 0; f < F; ++f) {
        const float gf = go[f];
        if (gw) p += x_ray[(int64_t)s * F + f] * gf;
        if (gx_ray) gx_ray[(int64_t)s * F + f] = ws * gf;
      }
      if (gw) gw[ray * S + s] = p;
    }
  }
}

}  // namespace

extern "C" {

int pv2_raymarch_weights_forward(const float* alpha, int64_t n_rays, int n_samples, float* weights,
                                 float* transmittance_or_null, pv2_stream_t stream) {
  PV2_REQUIRE(n_samples >= 1 && n_samples <= 64 * kMaxPerLane,
              "pv2_raymarch_weights_forward: 1 <= n_samples <= 256");
  PV2_REQUIRE(n_rays >= 0 && (n_rays + 3) / 4 < 0x7fffffffLL, "pv2_raymarch_weights_forward: bad ray count");
  if (n_rays == 0) return PV2_OK;
  hipLaunchKernelGGL(weights_fwd_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, alpha, n_rays, n_samples, weights, transmittance_or_null);
  return pv2::check_launch("raymarch_weights_forward");
}

int pv2_raymarch_weights_backward(const float* alpha, const float* grad_weights, int64_t n_rays,
                                  int n_samples, float* grad_alpha, pv2_stream_t stream) {
  PV2_REQUIRE(n_samples >= 1 && n_samples <= 64 * kMaxPerLane,
              "pv2_raymarch_weights_backward: 1 <= n_samples <= 256");
  PV2_REQUIRE(n_rays >= 0 && (n_rays + 3) / 4 < 0x7fffffffLL, "pv2_raymarch_weights_backward: bad ray count");
  if (n_rays == 0) return PV2_OK;
  hipLaunchKernelGGL(weights_bwd_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, alpha, grad_weights, n_rays, n_samples, grad_alpha);
  return pv2::check_launch("raymarch_weights_backward");
}

int pv2_raymarch_accumulate_forward(const float* weights, const float* values, int64_t n_rays,
                                    int n_samples, int n_features, float* out, pv2_stream_t stream) {
  PV2_REQUIRE(n_samples >= 1 && n_features >= 1 && n_features <= kMaxWideF,
              "pv2_raymarch_accumulate_forward: 1 <= n_features <= 512");
  PV2_REQUIRE(n_rays >= 0 && n_rays < 0x7fffffffLL, "pv2_raymarch_accumulate_forward: bad ray count");
  if (n_rays == 0) return PV2_OK;
  hipLaunchKernelGGL(accumulate_fwd_kernel, dim3((unsigned)n_rays), dim3(256), 0, (hipStream_t)stream,
                     weights, values, n_samples, n_features, out);
  return pv2::check_launch("raymarch_accumulate_forward");
}

int pv2_raymarch_accumulate_backward(const float* weights, const float* values, const float* grad_out,
                                     int64_t n_rays, int n_samples, int n_features,
                                     float* grad_weights_or_null, float* grad_values_or_null,
                                     pv2_stream_t stream) {
  PV2_REQUIRE(n_samples >= 1 && n_features >= 1 && n_features <= kMaxWideF,
              "pv2_raymarch_accumulate_backward: 1 <= n_features <= 512");
  PV2_REQUIRE(n_rays >= 0 && n_rays < 0x7fffffffLL, "pv2_raymarch_accumulate_backward: bad ray count");
  if (n_rays == 0 || (grad_weights_or_null == nullptr && grad_values_or_null == nullptr)) return PV2_OK;
  hipLaunchKernelGGL(accumulate_bwd_kernel, dim3((unsigned)n_rays), dim3(256), 0, (hipStream_t)stream,
                     weights, values, grad_out, n_samples, n_features, grad_weights_or_null,
                     grad_values_or_null);
  return pv2::check_launch("raymarch_accumulate_backward");
}

}  // extern "C"
