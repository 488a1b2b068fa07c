// Per-ray epilogue of the rendered composite rows and the semantic (contrastive) loss term on gfx950.
//
// Stands in for the tail of SurfaceModel.get_outputs - RGBRenderer / DepthRenderer / SemanticRenderer
// (ponder/models/ponder/render_utils/renderers.py:5-75: background blend, depth = sum w t / (sum w +
// 1e-10) clamped to the scene's sample range, the semantic head on the composited features) - and for
// the semantic term of SurfaceModel.get_loss (base_surface_model.py:102-211: F.normalize, logits =
// pred . gt^T / temperature, cross entropy against the ray's own index over the rays whose target is
// set).  With torch ops these were ~80 launches forward and ~90 backward of <= 10 us each on the
// training stream; here the matrix products stay with the library (three forward, four backward) and
// everything around them is four launches forward (bounds, rows, cross entropy, finalize) and two
// backward.  Every sum runs in a fixed order: bitwise reproducible.
//
// Composite row layout (fused_head.py): comp[r] = f'(n_f2) geo(n_geo) grad(3) normal(3) rgb(3) t 1 pad;
// the semantic head's input row is xbar = [grad | f' | geo | sum w]  (the last column carries the
// biases: the head is linear, so it was composited BEFORE its last layer - SURVEY Q5).
#include "common.h"

namespace {

constexpr int kThreads = 256;

// fixed-order block reductions (tree over LDS): every thread gets the result
template <bool MAX>
__device__ __forceinline__ float block_reduce(float v, float* s_red) {
  const int tid = threadIdx.x;
  __syncthreads();
  s_red[tid] = v;
  __syncthreads();
#pragma unroll
  for (int o = kThreads / 2; o > 0; o >>= 1) {
    if (tid < o) s_red[tid] = MAX ? fmaxf(s_red[tid], s_red[tid + o]) : s_red[tid] + s_red[tid + o];
    __syncthreads();
  }
  return s_red[0];
}

// lohi[b] = {min, max} of starts over the scene's rays (scene-major, rays_per_scene * S values)
__global__ __launch_bounds__(1024) void scene_bounds_of_samples_kernel(const float* __restrict__ starts,
                                                                      int64_t per_scene,
                                                                      float* __restrict__ lohi) {
  __shared__ float s_lo[1024], s_hi[1024];
  const int tid = threadIdx.x;
  const float* p = starts + (int64_t)blockIdx.x * per_scene;
  float lo = INFINITY, hi = -INFINITY;
  for (int64_t i = tid; i < per_scene; i += 1024) {
    const float v = p[i];
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
  s_lo[tid] = lo;
  s_hi[tid] = hi;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) {
      s_lo[tid] = fminf(s_lo[tid], s_lo[tid + o]);
      s_hi[tid] = fmaxf(s_hi[tid], s_hi[tid + o]);
    }
    __syncthreads();
  }
  if (tid == 0) {
    lohi[2 * blockIdx.x] = s_lo[0];
    lohi[2 * blockIdx.x + 1] = s_hi[0];
  }
}

struct Cols {
  int nv, f2, n_f2, geo, n_geo, g, rgb, t, w;
};

// one thread per (ray, column of xbar): xbar[r] = [grad(3) | f'(n_f2) | geo(n_geo) | sum w];
// the first four threads of a ray also form rgb (3) and depth
__global__ __launch_bounds__(kThreads) void ray_rows_forward_kernel(
    const float* __restrict__ comp, Cols c, int64_t R, int64_t rays_per_scene,
    const float* __restrict__ lohi, float bg0, float bg1, float bg2, float* __restrict__ xbar,
    float* __restrict__ rgb, float* __restrict__ depth) {
  const int nx = 3 + c.n_f2 + c.n_geo + 1;
  const int64_t total = R * nx;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += stride) {
    const int64_t r = e / nx;
    const int j = (int)(e - r * nx);
    const float* row = comp + r * c.nv;
    const int src = j < 3 ? c.g + j : j < 3 + c.n_f2 ? c.f2 + (j - 3) : j < nx - 1 ? c.geo + (j - 3 - c.n_f2) : c.w;
    xbar[e] = row[src];
    if (j < 3) {
      const float bg = j == 0 ? bg0 : j == 1 ? bg1 : bg2;
      // torch.addcmul(rgbc + bg, wsum, bg, value=-1):  (rgbc + bg) + (-1 * wsum) * bg
      rgb[r * 3 + j] = (row[c.rgb + j] + bg) + (-1.f * row[c.w]) * bg;
    } else if (j == 3) {
      const int64_t b = r / rays_per_scene;
      const float d = row[c.t] / (row[c.w] + 1e-10f);
      depth[r] = fminf(fmaxf(d, lohi[2 * b]), lohi[2 * b + 1]);
    }
  }
}

// Row i of the contrastive cross entropy.  raw[i, j] = sem_i . gt_j; n_i = |sem_i|;
// l_ij = raw_ij / max(n_i, 1e-12) / T;  ok_i = depth_gt_i > 0 and gt_i has a non-zero entry;
// loss_i = ok_i ? logsumexp_j l_ij - l_ii : 0.   info[i] = {n_i, scale_i, lse_i, ok_i, loss_i, 0, 0, 0}
constexpr int kInfo = 8;
__global__ __launch_bounds__(kThreads) void semantic_ce_forward_kernel(
    const float* __restrict__ raw, const float* __restrict__ sem, const float* __restrict__ gt,
    const float* __restrict__ depth_gt, int64_t R, int C, float temperature, float* __restrict__ info) {
  __shared__ float s_red[kThreads];
  const int64_t i = blockIdx.x;
  const int tid = threadIdx.x;
  float ss = 0.f, any = 0.f;
  for (int k = tid; k < C; k += kThreads) {
    const float v = sem[i * C + k];
    ss += v * v;
    if (gt[i * C + k] != 0.f) any = 1.f;
  }
  const float n = sqrtf(block_reduce<false>(ss, s_red));
  const bool ok = block_reduce<true>(any, s_red) > 0.f && depth_gt[i] > 0.f;
  const float scale = 1.f / fmaxf(n, 1e-12f) / temperature;
  const float* row = raw + i * R;
  float m = -INFINITY;
  for (int64_t j = tid; j < R; j += kThreads) m = fmaxf(m, row[j] * scale);
  m = block_reduce<true>(m, s_red);
  float se = 0.f;
  for (int64_t j = tid; j < R; j += kThreads) se += expf(row[j] * scale - m);
  se = block_reduce<false>(se, s_red);
  if (tid == 0) {
    const float lse = m + logf(se);
    float* o = info + i * kInfo;
    o[0] = n;
    o[1] = scale;
    o[2] = lse;
    o[3] = ok ? 1.f : 0.f;
    o[4] = ok ? lse - row[i] * scale : 0.f;
    o[5] = o[6] = o[7] = 0.f;
  }
}

// out[8] = depth, rgb, psnr, semantic, free_space, sdf, eikonal, TOTAL = the terms added in the order the
// model adds them (depth, rgb, semantic, free_space, sdf, eikonal; psnr is no loss); out[8] = count of ok rays.
// surf[6] = pv2_surface_loss_forward's out (terms with a zero weight are 0 there); has_sem: 0 / 1.
__global__ __launch_bounds__(kThreads) void ray_loss_finalize_kernel(const float* __restrict__ info,
                                                                     int64_t R, float w_sem, int has_sem,
                                                                     const float* __restrict__ surf,
                                                                     float* __restrict__ out) {
  __shared__ float s_red[kThreads];
  const int tid = threadIdx.x;
  float sl = 0.f, sc = 0.f;
  if (has_sem)
    for (int64_t i = tid; i < R; i += kThreads) {
      sl += info[i * kInfo + 4];
      sc += info[i * kInfo + 3];
    }
  sl = block_reduce<false>(sl, s_red);
  sc = block_reduce<false>(sc, s_red);
  if (tid == 0) {
    const float sem = has_sem ? sl / fmaxf(sc, 1.f) * w_sem : 0.f;
    out[0] = surf[0];
    out[1] = surf[1];
    out[2] = surf[2];
    out[3] = sem;
    out[4] = surf[3];
    out[5] = surf[4];
    out[6] = surf[5];
    float total = 0.f;
    total += surf[0];
    total += surf[1];
    if (has_sem) total += sem;
    total += surf[3];
    total += surf[4];
    total += surf[5];
    out[7] = total;
    out[8] = sc;
  }
}

// Backward of row i: dl_ij = a (softmax_ij - delta_ij), a = g w_sem / max(count, 1) for ok rows, else 0;
// d_raw[i, j] = dl_ij scale_i;  the path through the norm: d sem_i = sem_i * (-(sum_j dl_ij raw_ij)
// scale_i / n_i^2) when n_i > 1e-12 (F.normalize clamps the norm: no gradient through it below that).
__global__ __launch_bounds__(kThreads) void semantic_ce_backward_kernel(
    const float* __restrict__ raw, const float* __restrict__ sem, const float* __restrict__ info,
    const float* __restrict__ g_total, const float* __restrict__ out9, int64_t R, int C, float w_sem,
    float* __restrict__ d_raw, float* __restrict__ d_sem_norm) {
  __shared__ float s_red[kThreads];
  const int64_t i = blockIdx.x;
  const int tid = threadIdx.x;
  const float* o = info + i * kInfo;
  const float n = o[0], scale = o[1], lse = o[2];
  const float a = o[3] > 0.f ? g_total[0] * w_sem / fmaxf(out9[8], 1.f) : 0.f;
  const float* row = raw + i * R;
  float dot = 0.f;
  for (int64_t j = tid; j < R; j += kThreads) {
    const float rv = row[j];
    float dl = a * expf(rv * scale - lse);
    if (j == i) dl -= a;
    d_raw[i * R + j] = dl * scale;
    dot += dl * rv;
  }
  dot = block_reduce<false>(dot, s_red);
  const float coef = n > 1e-12f ? -dot * scale / (n * n) : 0.f;
  for (int k = tid; k < C; k += kThreads) d_sem_norm[i * C + k] = coef * sem[i * C + k];
}

// d comp from d xbar (the semantic head), d rgb, d depth (the loss terms): every element written
__global__ __launch_bounds__(kThreads) void ray_rows_backward_kernel(
    const float* __restrict__ comp, Cols c, int64_t R, int64_t rays_per_scene,
    const float* __restrict__ lohi, float bg0, float bg1, float bg2, const float* __restrict__ d_xbar,
    const float* __restrict__ g_rgb, const float* __restrict__ g_depth, float* __restrict__ d_comp) {
  const int nx = 3 + c.n_f2 + c.n_geo + 1;
  const int64_t total = R * c.nv;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += stride) {
    const int64_t r = e / c.nv;
    const int col = (int)(e - r * c.nv);
    const float* dx = d_xbar ? d_xbar + r * nx : nullptr;
    float v = 0.f;
    if (col >= c.f2 && col < c.f2 + c.n_f2) {
      v = dx ? dx[3 + col - c.f2] : 0.f;
    } else if (col >= c.geo && col < c.geo + c.n_geo) {
      v = dx ? dx[3 + c.n_f2 + col - c.geo] : 0.f;
    } else if (col >= c.g && col < c.g + 3) {
      v = dx ? dx[col - c.g] : 0.f;
    } else if (col >= c.rgb && col < c.rgb + 3) {
      v = g_rgb ? g_rgb[r * 3 + col - c.rgb] : 0.f;
    } else if (col == c.t || col == c.w) {
      const float* row = comp + r * c.nv;
      const int64_t b = r / rays_per_scene;
      const float den = row[c.w] + 1e-10f;
      const float d = row[c.t] / den;
      // clamp(d, lo, hi) passes the gradient where lo <= d <= hi
      const float gd = (g_depth && d >= lohi[2 * b] && d <= lohi[2 * b + 1]) ? g_depth[r] : 0.f;
      if (col == c.t) {
        v = gd / den;
      } else {
        v = -gd * row[c.t] / (den * den);
        if (dx) v += dx[nx - 1];
        if (g_rgb) v += -(g_rgb[r * 3] * bg0 + g_rgb[r * 3 + 1] * bg1 + g_rgb[r * 3 + 2] * bg2);
      }
    }
    d_comp[e] = v;
  }
}

Cols make_cols(int nv, int n_f2, int n_geo) {
  Cols c;
  c.nv = nv;
  c.f2 = 0;
  c.n_f2 = n_f2;
  c.geo = n_f2;
  c.n_geo = n_geo;
  c.g = n_f2 + n_geo;
  c.rgb = c.g + 6;
  c.t = c.g + 9;
  c.w = c.g + 10;
  return c;
}

}  // namespace

extern "C" {

int pv2_ray_rows_forward(const float* comp, int nv, int n_f2, int n_geo, const float* starts,
                         int64_t n_rays, int n_samples, int n_scenes, float bg0, float bg1, float bg2,
                         float* lohi, float* xbar, float* rgb, float* depth, pv2_stream_t stream) {
  PV2_REQUIRE(comp && starts && lohi && xbar && rgb && depth, "ray_rows_forward: null pointer");
  PV2_REQUIRE(n_rays > 0 && n_samples > 0 && n_scenes > 0 && n_rays % n_scenes == 0 &&
                  nv >= n_f2 + n_geo + 11,
              "ray_rows_forward: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const int64_t rps = n_rays / n_scenes;
  hipLaunchKernelGGL(scene_bounds_of_samples_kernel, dim3(n_scenes), dim3(1024), 0, s, starts,
                     rps * n_samples, lohi);
  const Cols c = make_cols(nv, n_f2, n_geo);
  const int nx = 3 + n_f2 + n_geo + 1;
  hipLaunchKernelGGL(ray_rows_forward_kernel, dim3(pv2::grid_for(n_rays * nx, kThreads)), dim3(kThreads), 0,
                     s, comp, c, n_rays, rps, lohi, bg0, bg1, bg2, xbar, rgb, depth);
  return pv2::check_launch("ray_rows_forward");
}

int pv2_ray_rows_backward(const float* comp, int nv, int n_f2, int n_geo, int64_t n_rays, int n_scenes,
                          float bg0, float bg1, float bg2, const float* lohi, const float* d_xbar,
                          const float* g_rgb, const float* g_depth, float* d_comp, pv2_stream_t stream) {
  PV2_REQUIRE(comp && lohi && d_comp, "ray_rows_backward: null pointer");
  PV2_REQUIRE(n_rays > 0 && n_scenes > 0 && n_rays % n_scenes == 0 && nv >= n_f2 + n_geo + 11,
              "ray_rows_backward: bad shape");
  const Cols c = make_cols(nv, n_f2, n_geo);
  hipLaunchKernelGGL(ray_rows_backward_kernel, dim3(pv2::grid_for(n_rays * nv, kThreads)), dim3(kThreads),
                     0, (hipStream_t)stream, comp, c, n_rays, n_rays / n_scenes, lohi, bg0, bg1, bg2, d_xbar,
                     g_rgb, g_depth, d_comp);
  return pv2::check_launch("ray_rows_backward");
}

int pv2_semantic_ce_forward(const float* raw, const float* sem, const float* gt, const float* depth_gt,
                            int64_t n_rays, int c_sem, float temperature, float* info,
                            pv2_stream_t stream) {
  PV2_REQUIRE(raw && sem && gt && depth_gt && info, "semantic_ce_forward: null pointer");
  PV2_REQUIRE(n_rays > 0 && c_sem > 0 && temperature > 0.f, "semantic_ce_forward: bad shape");
  hipLaunchKernelGGL(semantic_ce_forward_kernel, dim3((unsigned)n_rays), dim3(kThreads), 0,
                     (hipStream_t)stream, raw, sem, gt, depth_gt, n_rays, c_sem, temperature, info);
  return pv2::check_launch("semantic_ce_forward");
}

int pv2_ray_loss_finalize(const float* info, int64_t n_rays, float w_sem, const float* surface_terms,
                          float* out, pv2_stream_t stream) {
  PV2_REQUIRE(surface_terms && out && n_rays > 0, "ray_loss_finalize: bad arguments");
  hipLaunchKernelGGL(ray_loss_finalize_kernel, dim3(1), dim3(kThreads), 0, (hipStream_t)stream, info,
                     n_rays, w_sem, info != nullptr ? 1 : 0, surface_terms, out);
  return pv2::check_launch("ray_loss_finalize");
}

int pv2_semantic_ce_backward(const float* raw, const float* sem, const float* info, const float* g_total,
                             const float* out, int64_t n_rays, int c_sem, float w_sem, float* d_raw,
                             float* d_sem_norm, pv2_stream_t stream) {
  PV2_REQUIRE(raw && sem && info && g_total && out && d_raw && d_sem_norm,
              "semantic_ce_backward: null pointer");
  PV2_REQUIRE(n_rays > 0 && c_sem > 0, "semantic_ce_backward: bad shape");
  hipLaunchKernelGGL(semantic_ce_backward_kernel, dim3((unsigned)n_rays), dim3(kThreads), 0,
                     (hipStream_t)stream, raw, sem, info, g_total, out, n_rays, c_sem, w_sem, d_raw,
                     d_sem_norm);
  return pv2::check_launch("semantic_ce_backward");
}

int pv2_ray_loss_info_floats(void) { return kInfo; }

}  // extern "C"
