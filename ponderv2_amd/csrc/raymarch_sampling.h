// Shared device code of the fused ray-march kernels (raymarch_fused.hip: the ScanNet head shape,
// raymarch_narrow.hip: narrow SDF decoders): the trilinear corner model on a channels-last
// volume, Softplus(beta=100) with its derivatives, and the head-independent part of the coarse
// pass - stratified bins, fixed-inv_s section alphas, weights, inverse-CDF importance samples and
// the sorted merge (ponder/models/ponder/render_utils/ray_samplers.py:55-107, 227-322, 355-463,
// rays.py:83-105, 118-153 of the reference).
#pragma once
#include "common.h"

namespace pv2rm {

struct Vol {
  const float* p;
  int B, Z, Y, X;
  int64_t rays_per_scene;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// torch.nn.Softplus(beta=100, threshold=20) and its first two derivatives
__device__ __forceinline__ void softplus100(float h, float* sp, float* d1, float* d2) {
  const float bx = 100.f * h;
  if (bx > 20.f) {
    *sp = h;
    *d1 = 1.f;
    *d2 = 0.f;
  } else {
    const float e = expf(bx);
    *sp = log1pf(e) * 0.01f;
    const float s = e / (1.f + e);
    *d1 = s;
    *d2 = 100.f * s * (1.f - s);
  }
}

// Softplus(beta = 100, threshold = 20) with its first two derivatives on the hardware exp / log / rcp
// units (1 ulp each): ~8 instructions per element where the libm forms take ~50 - the activation, not
// the matrix products, was the longest part of a layer.  log(1 + e) instead of log1p(e) costs at most
// 6e-8 / 100 ABSOLUTE on a value that is >= 0 and enters sums of O(1) terms.
__device__ __forceinline__ void softplus_fast(float h, float* sp, float* d1, float* d2) {
  const float bx = 100.f * h;
  const float e = __expf(fminf(bx, 20.f));
  const float r = __frcp_rn(1.f + e);
  const bool lin = bx > 20.f;
  const float s = e * r;
  *sp = lin ? h : __logf(1.f + e) * 0.01f;
  *d1 = lin ? 1.f : s;
  *d2 = lin ? 0.f : 100.f * s * r;    // s (1 - s) = e / (1 + e)^2
}

// Trilinear corner model, zeros padding, align_corners, no smoothstep.  p in [0,1]^3 (x, y, z).
struct Axes {
  int ix, iy, iz;
  float tx, ty, tz;
};

__device__ __forceinline__ Axes make_axes(float px, float py, float pz, const Vol& v) {
  Axes a;
  const float x = px * (float)(v.X - 1), y = py * (float)(v.Y - 1), z = pz * (float)(v.Z - 1);
  const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
  a.tx = x - fx;
  a.ty = y - fy;
  a.tz = z - fz;
  // clamp before the int conversion: far-away points (the coarse pass samples un-normalised
  // coordinates) must not overflow; any index outside [-1, size] is out of bounds either way
  a.ix = (int)fminf(fmaxf(fx, -2.f), (float)v.X + 1.f);
  a.iy = (int)fminf(fmaxf(fy, -2.f), (float)v.Y + 1.f);
  a.iz = (int)fminf(fmaxf(fz, -2.f), (float)v.Z + 1.f);
  return a;
}

// corner c (bit0 = x, bit1 = y, bit2 = z) of a channels-last volume with nch channels: in-bounds
// flag, element offset of its channel 0, weight and d weight / d p (0 when out of bounds)
__device__ __forceinline__ bool corner(const Axes& a, const Vol& v, int scene, int c, int nch,
                                       int64_t* off, float* w, float* dx, float* dy, float* dz) {
  const int bx = c & 1, by = (c >> 1) & 1, bz = (c >> 2) & 1;
  const int x = a.ix + bx, y = a.iy + by, z = a.iz + bz;
  const bool ok = x >= 0 && x < v.X && y >= 0 && y < v.Y && z >= 0 && z < v.Z;
  const float wx = bx ? a.tx : 1.f - a.tx, wy = by ? a.ty : 1.f - a.ty, wz = bz ? a.tz : 1.f - a.tz;
  const float sx = (bx ? 1.f : -1.f) * (float)(v.X - 1);
  const float sy = (by ? 1.f : -1.f) * (float)(v.Y - 1);
  const float sz = (bz ? 1.f : -1.f) * (float)(v.Z - 1);
  *off = ok ? ((((int64_t)scene * v.Z + z) * v.Y + y) * v.X + x) * nch : 0;
  *w = ok ? wx * wy * wz : 0.f;
  *dx = ok ? sx * wy * wz : 0.f;
  *dy = ok ? wx * sy * wz : 0.f;
  *dz = ok ? wx * wy * sz : 0.f;
  return ok;
}

__device__ __forceinline__ float4 ldg4(const float* p) {
  return *reinterpret_cast<const float4*>(p);
}

// ------------------------------------------------------------------------------------------
// Coarse pass, one workgroup per ray: everything but the SDF evaluation itself.
// ------------------------------------------------------------------------------------------
constexpr int kMaxS0 = 128;
constexpr int kMaxImp = 63;

struct SampleLds {
  float bins[kMaxS0 + 1], e[kMaxS0 + 1], sdf[kMaxS0], cos[kMaxS0], alpha[kMaxS0], w[kMaxS0],
      cdf[kMaxS0 + 1], nw[kMaxImp + 1], out[kMaxS0 + kMaxImp + 2];
};

__device__ __forceinline__ float lerp_rn(float lo, float hi, float t) {
  return __fadd_rn(lo, __fmul_rn(__fsub_rn(hi, lo), t));
}
__device__ __forceinline__ float to_euclid(float x, float nearv, float farv) {
  return __fadd_rn(__fmul_rn(x, farv), __fmul_rn(__fsub_rn(1.f, x), nearv));
}

// spacing bin edges and their ray distances into L.bins / L.e (ends with a workgroup barrier)
__device__ __forceinline__ void coarse_bins(SampleLds& L, int64_t ray, float nearv, float farv, int S0,
                                            const float* __restrict__ lin_bins,
                                            const float* __restrict__ t_rand, int t_rand_cols,
                                            int tid, int nthreads) {
  // spacing bin edges (ray_samplers.py:70-88) and their ray distances
  for (int j = tid; j <= S0; j += nthreads) {
    float b = lin_bins[j];
    if (t_rand) {
      const float lo = j == 0 ? lin_bins[0] : __fmul_rn(__fadd_rn(lin_bins[j], lin_bins[j - 1]), 0.5f);
      const float hi = j == S0 ? lin_bins[S0] : __fmul_rn(__fadd_rn(lin_bins[j + 1], lin_bins[j]), 0.5f);
      const float t = t_rand[ray * t_rand_cols + (t_rand_cols == 1 ? 0 : j)];
      b = lerp_rn(lo, hi, t);
    }
    L.bins[j] = b;
    L.e[j] = to_euclid(b, nearv, farv);
  }
  __syncthreads();

}

// From the coarse SDF values in L.sdf (written by the caller, barrier included) to the merged
// sample list of the ray.  Needs wave 0 complete (64 lanes) for the scans.
__device__ __forceinline__ void importance_merge(
    SampleLds& L, int64_t ray, float nearv, float farv, int S0, int n_imp,
    const float* __restrict__ lin_u, const float* __restrict__ u_rand, int u_rand_cols,
    float base_inv_s, float* __restrict__ bins_out, float* __restrict__ starts_out,
    float* __restrict__ deltas_out, int32_t* __restrict__ dbg_idx, float* __restrict__ dbg_sdf,
    float* __restrict__ dbg_w, int tid, int nthreads, int wave, int lane) {
  // fixed-inv_s section alphas (ray_samplers.py:426-463)
  const int n1 = S0 - 1;
  for (int j = tid; j < n1; j += nthreads) {
    const float dist = __fsub_rn(L.e[j + 1], L.e[j]);
    L.cos[j] = (L.sdf[j + 1] - L.sdf[j]) / (dist + 1e-5f);
  }
  __syncthreads();
  for (int j = tid; j < n1; j += nthreads) {
    const float dist = __fsub_rn(L.e[j + 1], L.e[j]);
    const float prev = j > 0 ? L.cos[j - 1] : 0.f;
    const float cv = fminf(fmaxf(fminf(prev, L.cos[j]), -1e3f), 0.f);
    const float mid = (L.sdf[j] + L.sdf[j + 1]) * 0.5f;
    const float pc = sigmoidf_((mid - cv * dist * 0.5f) * base_inv_s);
    const float nc = sigmoidf_((mid + cv * dist * 0.5f) * base_inv_s);
    L.alpha[j] = (pc - nc + 1e-5f) / (pc + 1e-5f);
  }
  __syncthreads();

  // weights (rays.py:83-105) and the padded pdf / cdf of PDFSampler (ray_samplers.py:243-262)
  if (wave == 0) {
    float a[2], x[2];
    float prod = 1.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = lane * 2 + u;
      a[u] = j < n1 ? L.alpha[j] : 0.f;
      x[u] = j < n1 ? 1.f - a[u] + 1e-7f : 1.f;
      prod *= x[u];
    }
    float incl = prod;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float up = __shfl_up(incl, o);
      if (lane >= o) incl *= up;
    }
    float T = __shfl_up(incl, 1);
    if (lane == 0) T = 1.f;
    float w[2], wsum = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      w[u] = a[u] * T;   // 0 for j >= n1 (a = 0): the appended zero weight of sample S0-1
      T *= x[u];
      wsum += w[u];
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) wsum += __shfl_xor(wsum, o);
    const float pad = fmaxf(1e-5f - wsum, 0.f);
    const float den = wsum + pad;
    float pdf[2], run = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = lane * 2 + u;
      if (j < S0) L.w[j] = w[u];
      pdf[u] = j < S0 ? (w[u] + pad / (float)S0) / den : 0.f;
      run += pdf[u];
    }
    float incs = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float up = __shfl_up(incs, o);
      if (lane >= o) incs += up;
    }
    float before = incs - run;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = lane * 2 + u;
      before += pdf[u];
      if (j < S0) L.cdf[j + 1] = fminf(1.f, before);
    }
    if (lane == 0) L.cdf[0] = 0.f;
  }
  __syncthreads();

  // inverse-CDF samples (ray_samplers.py:263-313)
  const int nb = n_imp + 1;
  for (int m = tid; m < nb; m += nthreads) {
    float u = lin_u[m];
    if (u_rand) u = __fadd_rn(u, u_rand[ray * u_rand_cols + (u_rand_cols == 1 ? 0 : m)] / (float)nb);
    else u = __fadd_rn(u, 1.f / (float)(2 * nb));
    int idx = 0;  // searchsorted(cdf, u, right=True): number of entries <= u
    for (int j = 0; j <= S0; ++j) idx += L.cdf[j] <= u ? 1 : 0;
    const int below = min(max(idx - 1, 0), S0), above = min(max(idx, 0), S0);
    const float c0 = L.cdf[below], c1 = L.cdf[above], b0 = L.bins[below], b1 = L.bins[above];
    float den = __fsub_rn(c1, c0);
    if (den < 1e-5f) den = 1.f;
    const float t = fminf(fmaxf(__fsub_rn(u, c0) / den, 0.f), 1.f);
    L.nw[m] = lerp_rn(b0, b1, t);
    if (dbg_idx) dbg_idx[ray * nb + m] = idx;
  }
  __syncthreads();

  // sorted merge of the S0 coarse and n_imp new spacing starts (rays.py:118-153); both lists are
  // non-decreasing, so every element's output slot is its own index plus a count in the other list
  for (int j = tid; j < S0; j += nthreads) {
    const float v = L.bins[j];
    int pos = j;
    for (int m = 0; m < n_imp; ++m) pos += L.nw[m] < v ? 1 : 0;
    L.out[pos] = v;
  }
  for (int m = tid; m < n_imp; m += nthreads) {
    const float v = L.nw[m];
    int pos = m;
    for (int j = 0; j < S0; ++j) pos += L.bins[j] <= v ? 1 : 0;
    L.out[pos] = v;
  }
  const int S = S0 + n_imp;
  if (tid == 0) L.out[S] = fmaxf(L.bins[S0], L.nw[n_imp]);
  __syncthreads();
  for (int j = tid; j <= S; j += nthreads) {
    bins_out[ray * (S + 1) + j] = L.out[j];
    if (j < S) {
      const float e0 = to_euclid(L.out[j], nearv, farv), e1 = to_euclid(L.out[j + 1], nearv, farv);
      starts_out[ray * S + j] = e0;
      deltas_out[ray * S + j] = __fsub_rn(e1, e0);
    }
  }
  if (dbg_sdf)
    for (int j = tid; j < S0; j += nthreads) {
      dbg_sdf[ray * S0 + j] = L.sdf[j];
      dbg_w[ray * S0 + j] = L.w[j];
    }
}

}  // namespace pv2rm
