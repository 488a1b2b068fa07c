// The per-ray / per-sample loss terms of the surface-rendering models in three launches (gfx950).
//
// ponder/models/ponder/render_utils/models/base_surface_model.py:102-211 of the reference evaluates
// depth, colour (+ psnr), free-space, SDF and eikonal terms as ~60 small elementwise / reduction ops on
// (R,), (R, 3), (R, S) and (R, S, 3) tensors, and autograd runs ~100 more on the way back: 2 ms of HOST
// time on a step the host bounds.  Here:
//   partials  one pass over rays and samples -> per-workgroup sums of the nine quantities below
//   finalize  one workgroup adds the partial rows in order (reproducible) and forms the terms
//   backward  one elementwise pass: d/d depth, d/d rgb, d/d sdf, d/d grad sdf
// with, for ray r (gt depth D, valid = D > 0) and sample k at distance z:
//   front = valid & z < D - trunc     back = valid & z > D + trunc     near = valid & !front & !back
//   s0 = sum valid |D - depth|        s1 = sum valid
//   s2 = sum |rgb - rgb_gt|           s3 = sum (rgb - rgb_gt)^2                  (over R x 3)
//   s4 = sum front relu(trunc - sdf)  s5 = sum front
//   s6 = sum near |z + sdf - D|       s7 = sum near
//   s8 = sum (|grad| - 1)^2                                                      (over R x S)
//   depth_loss = w s0 / max(s1, 1)    rgb_loss = w s2 / 3R    psnr = 20 log10(1 / sqrt(s3 / 3R))
//   free_space = w s4 / max(s5, 1)    sdf_loss = w s6 / max(s7, 1)    eikonal = w s8 / RS
// The semantic term (a matrix product and a cross entropy) stays with the library.
#include "common.h"

namespace {

constexpr int kNS = 9;
constexpr int kThreads = 256;
constexpr int kMaxBlocks = 256;

__device__ __forceinline__ float block_sum(float v, float* s_red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0)
    for (int w = 0; w < kThreads / 64; ++w) t += s_red[w];
  return t;   // (thread 0)
}

__global__ __launch_bounds__(kThreads) void surface_loss_partials_kernel(
    const float* __restrict__ depth, const float* __restrict__ depth_gt, const float* __restrict__ rgb,
    const float* __restrict__ rgb_gt, const float* __restrict__ sdf, const float* __restrict__ z,
    const float* __restrict__ grad, int64_t R, int S, float trunc, float* __restrict__ partials) {
  __shared__ float s_red[kThreads / 64];
  float acc[kNS];
#pragma unroll
  for (int i = 0; i < kNS; ++i) acc[i] = 0.f;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  const int64_t t0 = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  for (int64_t r = t0; r < R; r += stride) {
    const float D = depth_gt[r];
    const float valid = D > 0.f ? 1.f : 0.f;
    acc[0] += valid * fabsf(D - depth[r]);
    acc[1] += valid;
    if (rgb) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float d = rgb[r * 3 + c] - rgb_gt[r * 3 + c];
        acc[2] += fabsf(d);
        acc[3] += d * d;
      }
    }
  }
  const int64_t N = R * S;
  for (int64_t n = t0; n < N; n += stride) {
    const int64_t r = n / S;
    const float D = depth_gt[r];
    const bool valid = D > 0.f;
    const float zz = z[n], sd = sdf[n];
    const bool front = valid && zz < D - trunc;
    const bool back = valid && zz > D + trunc;
    const bool near = valid && !front && !back;
    if (front) {
      acc[4] += fmaxf(trunc - sd, 0.f);
      acc[5] += 1.f;
    }
    if (near) {
      acc[6] += fabsf(zz + sd - D);
      acc[7] += 1.f;
    }
    if (grad) {
      const float gx = grad[n * 3], gy = grad[n * 3 + 1], gz = grad[n * 3 + 2];
      const float e = sqrtf(gx * gx + gy * gy + gz * gz) - 1.f;
      acc[8] += e * e;
    }
  }
#pragma unroll
  for (int i = 0; i < kNS; ++i) {
    const float t = block_sum(acc[i], s_red);
    if (threadIdx.x == 0) partials[(int64_t)blockIdx.x * kNS + i] = t;
  }
}

// weights: [depth, rgb, free_space, sdf, eikonal];  out: [depth_loss, rgb_loss, psnr, free_space_loss,
// sdf_loss, eikonal_loss];  sums: the nine totals (kept for the backward)
__global__ void surface_loss_finalize_kernel(const float* __restrict__ partials, int n_blocks, int64_t R,
                                             int S, const float* __restrict__ weights,
                                             float* __restrict__ out, float* __restrict__ sums) {
  __shared__ float s[kNS];
  if (threadIdx.x < kNS) {
    float t = 0.f;
    for (int b = 0; b < n_blocks; ++b) t += partials[(int64_t)b * kNS + threadIdx.x];
    s[threadIdx.x] = t;
    sums[threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float n3 = 3.f * (float)R, ns = (float)R * (float)S;
    out[0] = s[0] / fmaxf(s[1], 1.f) * weights[0];
    out[1] = s[2] / n3 * weights[1];
    out[2] = 20.f * log10f(1.f / sqrtf(s[3] / n3));
    out[3] = s[4] / fmaxf(s[5], 1.f) * weights[2];
    out[4] = s[6] / fmaxf(s[7], 1.f) * weights[3];
    out[5] = s[8] / ns * weights[4];
  }
}

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// up: the upstream gradients of the six outputs (a missing one arrives as NULL)
__global__ __launch_bounds__(kThreads) void surface_loss_backward_kernel(
    const float* __restrict__ depth, const float* __restrict__ depth_gt, const float* __restrict__ rgb,
    const float* __restrict__ rgb_gt, const float* __restrict__ sdf, const float* __restrict__ z,
    const float* __restrict__ grad, int64_t R, int S, float trunc, const float* __restrict__ weights,
    const float* __restrict__ sums, const float* up0, const float* up1, const float* up2,
    const float* up3, const float* up4, const float* up5, float* __restrict__ g_depth,
    float* __restrict__ g_rgb, float* __restrict__ g_sdf, float* __restrict__ g_grad) {
  const float u_depth = up0 ? up0[0] : 0.f, u_rgb = up1 ? up1[0] : 0.f, u_psnr = up2 ? up2[0] : 0.f;
  const float u_fs = up3 ? up3[0] : 0.f, u_sdf = up4 ? up4[0] : 0.f, u_eik = up5 ? up5[0] : 0.f;
  const float n3 = 3.f * (float)R, ns = (float)R * (float)S;
  const float k_depth = u_depth * weights[0] / fmaxf(sums[1], 1.f);
  const float k_rgb = u_rgb * weights[1] / n3;
  // psnr = -10 log10(mse): d/d rgb = -(10 / ln 10) / mse * 2 (rgb - gt) / 3R
  const float k_psnr = sums[3] > 0.f ? u_psnr * (-10.f / 2.302585093f) / sums[3] * 2.f : 0.f;
  const float k_fs = u_fs * weights[2] / fmaxf(sums[5], 1.f);
  const float k_sdf = u_sdf * weights[3] / fmaxf(sums[7], 1.f);
  const float k_eik = u_eik * weights[4] / ns * 2.f;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  const int64_t t0 = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  for (int64_t r = t0; r < R; r += stride) {
    const float D = depth_gt[r];
    g_depth[r] = D > 0.f ? k_depth * sgn(depth[r] - D) : 0.f;
    if (g_rgb) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float d = rgb[r * 3 + c] - rgb_gt[r * 3 + c];
        g_rgb[r * 3 + c] = k_rgb * sgn(d) + k_psnr * d;
      }
    }
  }
  const int64_t N = R * S;
  for (int64_t n = t0; n < N; n += stride) {
    const int64_t r = n / S;
    const float D = depth_gt[r];
    const bool valid = D > 0.f;
    const float zz = z[n], sd = sdf[n];
    const bool front = valid && zz < D - trunc;
    const bool back = valid && zz > D + trunc;
    const bool near = valid && !front && !back;
    float g = 0.f;
    if (front && trunc - sd > 0.f) g -= k_fs;
    if (near) g += k_sdf * sgn(zz + sd - D);
    g_sdf[n] = g;
    if (g_grad) {
      const float gx = grad[n * 3], gy = grad[n * 3 + 1], gz = grad[n * 3 + 2];
      const float nrm = sqrtf(gx * gx + gy * gy + gz * gz);
      const float f = nrm > 0.f ? k_eik * (nrm - 1.f) / nrm : 0.f;
      g_grad[n * 3] = f * gx;
      g_grad[n * 3 + 1] = f * gy;
      g_grad[n * 3 + 2] = f * gz;
    }
  }
}

int blocks_for(int64_t R, int S) {
  const int64_t b = (R * S + kThreads - 1) / kThreads;
  return (int)(b < 1 ? 1 : (b > kMaxBlocks ? kMaxBlocks : b));
}

}  // namespace

extern "C" {

int pv2_surface_loss_workspace_floats(void) { return kMaxBlocks * kNS; }

int pv2_surface_loss_forward(const float* depth, const float* depth_gt, const float* rgb,
                             const float* rgb_gt, const float* sdf, const float* z, const float* grad,
                             int64_t n_rays, int n_samples, float trunc, const float* weights,
                             float* workspace, float* out, float* sums, pv2_stream_t stream) {
  PV2_REQUIRE(n_rays >= 1 && n_samples >= 1, "pv2_surface_loss_forward: empty input");
  PV2_REQUIRE((rgb == nullptr) == (rgb_gt == nullptr), "pv2_surface_loss_forward: rgb and its target");
  const int nb = blocks_for(n_rays, n_samples);
  hipLaunchKernelGGL(surface_loss_partials_kernel, dim3(nb), dim3(kThreads), 0, (hipStream_t)stream, depth,
                     depth_gt, rgb, rgb_gt, sdf, z, grad, n_rays, n_samples, trunc, workspace);
  hipLaunchKernelGGL(surface_loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, workspace, nb,
                     n_rays, n_samples, weights, out, sums);
  return pv2::check_launch("surface_loss_forward");
}

int pv2_surface_loss_backward(const float* depth, const float* depth_gt, const float* rgb,
                              const float* rgb_gt, const float* sdf, const float* z, const float* grad,
                              int64_t n_rays, int n_samples, float trunc, const float* weights,
                              const float* sums, const float* const* upstream, float* g_depth,
                              float* g_rgb, float* g_sdf, float* g_grad, pv2_stream_t stream) {
  PV2_REQUIRE(n_rays >= 1 && n_samples >= 1 && upstream != nullptr, "pv2_surface_loss_backward: bad input");
  const int nb = blocks_for(n_rays, n_samples);
  hipLaunchKernelGGL(surface_loss_backward_kernel, dim3(nb), dim3(kThreads), 0, (hipStream_t)stream, depth,
                     depth_gt, rgb, rgb_gt, sdf, z, grad, n_rays, n_samples, trunc, weights, sums, upstream[0],
                     upstream[1], upstream[2], upstream[3], upstream[4], upstream[5], g_depth, g_rgb, g_sdf,
                     g_grad);
  return pv2::check_launch("surface_loss_backward");
}

}  // extern "C"
