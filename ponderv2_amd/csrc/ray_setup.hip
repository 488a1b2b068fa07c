// Ray set-up of the indoor pre-training model in four launches (gfx950).
//
// ponder/models/ponder/ponder_indoor_base.py:344-470 of the reference (to_unit_cube, ray_sample,
// get_mask_at_box) normalises every scene into the unit cube, composes the camera matrices with that
// transform, and turns the chosen pixels of every view into rays with their colour / depth / semantic
// targets - numpy per view there, ~140 small batched torch launches in ponderv2_amd's first device
// version (2 ms of HOST time on a step the host bounds).  Here, same arithmetic (fp32, no contraction,
// the slab test in double as there):
//   scene_bounds      per-scene min / max of the point coordinates (one workgroup per scene)
//   unit_cube_setup   per scene: centre, extent, scale, floor level -> S = diag(scale) | t, its inverse,
//                     depth / point-cloud scales, bounding box; per view: world -> camera composed with
//                     S^-1, its inverse (the camera pose), K^-1, the image-plane normal
//   unit_cube_points  every point: clip(p * scale + t), then back to metres of the scaled cloud
//   ray_gen           every chosen pixel: direction through K^-1 and the pose, colour / depth / semantic
//                     lookup, plane-to-ray depth, slab test against the padded cube
// The uniform random choice of n valid pixels per view stays with the library (rand + topk).
#include "common.h"

#pragma clang fp contract(off)

namespace {

// inverse of an n x n matrix (n <= 4) by Gauss-Jordan elimination with partial pivoting in double
// precision, result rounded to fp32 - the arithmetic of pv2_small_inverse
__device__ __forceinline__ void inverse_small(const float* a, int n, float* out) {
  double w[4][8];
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < n; ++c) {
      w[r][c] = (double)a[r * n + c];
      w[r][n + c] = r == c ? 1.0 : 0.0;
    }
  for (int col = 0; col < n; ++col) {
    int piv = col;
    double best = fabs(w[col][col]);
    for (int r = col + 1; r < n; ++r)
      if (fabs(w[r][col]) > best) {
        best = fabs(w[r][col]);
        piv = r;
      }
    if (piv != col)
      for (int c = 0; c < 2 * n; ++c) {
        const double t = w[col][c];
        w[col][c] = w[piv][c];
        w[piv][c] = t;
      }
    const double inv = 1.0 / w[col][col];
    for (int c = 0; c < 2 * n; ++c) w[col][c] *= inv;
    for (int r = 0; r < n; ++r) {
      if (r == col) continue;
      const double f = w[r][col];
      for (int c = 0; c < 2 * n; ++c) w[r][c] -= f * w[col][c];
    }
  }
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < n; ++c) out[r * n + c] = (float)w[r][n + c];
}

constexpr float kClipLo = -0.5f + 1e-5f, kClipHi = 0.5f - 1e-5f;

__global__ __launch_bounds__(256) void scene_bounds_kernel(const float* __restrict__ coords,
                                                           const int64_t* __restrict__ offsets,
                                                           float* __restrict__ mm) {
  __shared__ float s_lo[4][3], s_hi[4][3];
  const int b = blockIdx.x;
  const int64_t start = b == 0 ? 0 : offsets[b - 1], end = offsets[b];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int64_t i = start + threadIdx.x; i < end; i += blockDim.x)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = coords[i * 3 + a];
      lo[a] = fminf(lo[a], v);
      hi[a] = fmaxf(hi[a], v);
    }
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
    }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      s_lo[wave][a] = lo[a];
      s_hi[wave][a] = hi[a];
    }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int a = threadIdx.x;
    float l = s_lo[0][a], h = s_hi[0][a];
    for (int w = 1; w < 4; ++w) {
      l = fminf(l, s_lo[w][a]);
      h = fmaxf(h, s_hi[w][a]);
    }
    mm[b * 6 + a] = l;
    mm[b * 6 + 3 + a] = h;
  }
}

// scene record (floats): 0 scale | 1..3 t | 4 extent (pc_scale) | 5 depth scale | 6..11 bbox lo, hi
constexpr int kSceneRec = 12;
// view record (floats): 0..11 pose rows 0..2 (camera -> world, [R | o]) | 12..20 K^-1 | 21..23 plane normal
constexpr int kViewRec = 24;

__global__ void unit_cube_setup_kernel(const float* __restrict__ mm, int B, int V, float z_level,
                                       const float* __restrict__ depth_scale,
                                       const float* __restrict__ extrinsic,   // [B, V, 4, 4] world -> camera
                                       const float* __restrict__ kmat,        // [B, V, 3, 3]
                                       float* __restrict__ scene, float* __restrict__ extr_out,
                                       float* __restrict__ view) {
  extern __shared__ float s_sinv[];   // [B][16]
  const int t = threadIdx.x;
  if (t < B) {
    float lo[3], hi[3], loc[3];
    float extent = -INFINITY;
    for (int a = 0; a < 3; ++a) {
      lo[a] = mm[t * 6 + a] - 1e-5f;
      hi[a] = mm[t * 6 + 3 + a] + 1e-5f;
      loc[a] = (lo[a] + hi[a]) / 2.f;
      extent = fmaxf(extent, hi[a] - lo[a]);
    }
    const float scale = 1.0f / extent;
    const float z_min = (mm[t * 6 + 2] - loc[2]) * scale;
    const float m23 = -z_min + z_level;
    float S[16] = {0};
    S[0] = S[5] = S[10] = scale;
    S[15] = 1.f;
    S[3] = scale * -loc[0];
    S[7] = scale * -loc[1];
    S[11] = scale * -loc[2] + m23;
    inverse_small(S, 4, s_sinv + t * 16);
    float* rec = scene + t * kSceneRec;
    rec[0] = scale;
    rec[1] = S[3];
    rec[2] = S[7];
    rec[3] = S[11];
    rec[4] = extent;
    rec[5] = scale * depth_scale[t];
    for (int a = 0; a < 3; ++a) {   // min / max of the clipped, transformed cloud: the transform is monotone
      const float nl = fminf(fmaxf(mm[t * 6 + a] * scale + S[3 + 4 * a], kClipLo), kClipHi);
      const float nh = fminf(fmaxf(mm[t * 6 + 3 + a] * scale + S[3 + 4 * a], kClipLo), kClipHi);
      rec[6 + a] = ((nl - 1e-5f) + 0.5f) * extent;
      rec[9 + a] = ((nh + 1e-5f) + 0.5f) * extent;
    }
  }
  __syncthreads();
  if (t < B * V) {
    const int b = t / V;
    float pose_in[16], ext[16];
    for (int i = 0; i < 16; ++i) pose_in[i] = extrinsic[t * 16 + i];
    pose_in[15] = 1.f;
    const float* si = s_sinv + b * 16;
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) {
        float acc = pose_in[r * 4] * si[c];
        for (int k = 1; k < 4; ++k) acc = fmaf(pose_in[r * 4 + k], si[k * 4 + c], acc);
        ext[r * 4 + c] = acc;
        extr_out[t * 16 + r * 4 + c] = acc;
      }
    float RT[16], pose[16], c2w[16], kin[9];
    for (int i = 0; i < 12; ++i) RT[i] = ext[i];
    RT[12] = RT[13] = RT[14] = 0.f;
    RT[15] = 1.f;
    inverse_small(RT, 4, pose);
    inverse_small(ext, 4, c2w);
    inverse_small(kmat + t * 9, 3, kin);
    float* rec = view + t * kViewRec;
    for (int i = 0; i < 12; ++i) rec[i] = pose[i];
    for (int i = 0; i < 9; ++i) rec[12 + i] = kin[i];
    float pl[3];
    for (int a = 0; a < 3; ++a) pl[a] = (c2w[a * 4 + 2] + c2w[a * 4 + 3]) - pose[a * 4 + 3];
    const float nrm = sqrtf(pl[0] * pl[0] + pl[1] * pl[1] + pl[2] * pl[2]);
    for (int a = 0; a < 3; ++a) rec[21 + a] = pl[a] / nrm;
  }
}

__global__ __launch_bounds__(256) void unit_cube_points_kernel(const float* __restrict__ coords,
                                                               const int64_t* __restrict__ offsets, int B,
                                                               int64_t n, const float* __restrict__ scene,
                                                               float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int b = 0;
  while (b < B - 1 && i >= offsets[b]) ++b;
  const float* rec = scene + b * kSceneRec;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float v = fminf(fmaxf(coords[i * 3 + a] * rec[0] + rec[1 + a], kClipLo), kClipHi);
    out[i * 3 + a] = (v + 0.5f) * rec[4];
  }
}

__global__ __launch_bounds__(256) void ray_gen_kernel(
    const int64_t* __restrict__ flat, int B, int V, int n, int H, int W, const float* __restrict__ view,
    const float* __restrict__ scene, const float* __restrict__ colors, const float* __restrict__ depths,
    const int64_t* __restrict__ semantic, double lo0, double lo1, double lo2, double hi0, double hi1,
    double hi2, float* __restrict__ ray_o, float* __restrict__ ray_d, float* __restrict__ rgb,
    float* __restrict__ depth, int64_t* __restrict__ sem_row) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * V * n;
  if (i >= total) return;
  const int64_t bv = i / n;
  const int b = (int)(bv / V);
  const float* rec = view + bv * kViewRec;
  const int64_t pix = flat[i];
  const float px = (float)(pix % W), py = (float)(pix / W);
  float p[3], v[3], d[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float acc = rec[12 + a * 3] * px;
    acc = fmaf(rec[12 + a * 3 + 1], py, acc);
    p[a] = fmaf(rec[12 + a * 3 + 2], 1.f, acc);
  }
  const float pn = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
#pragma unroll
  for (int a = 0; a < 3; ++a) p[a] = p[a] / pn;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float acc = rec[a * 4] * p[0];
    acc = fmaf(rec[a * 4 + 1], p[1], acc);
    v[a] = fmaf(rec[a * 4 + 2], p[2], acc);
  }
  const float vn = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
  float o[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    d[a] = v[a] / vn;
    o[a] = rec[a * 4 + 3];
  }
  // plane-to-plane depth -> distance along the ray
  const int64_t src = bv * H * W + pix;
  const float dep_raw = depths[src];
  const float dep = dep_raw * (dep_raw > 0.f ? 1.f : 0.f) * scene[b * kSceneRec + 5];
  const float cosang = d[0] * rec[21] + d[1] * rec[22] + d[2] * rec[23];
  float dist = dep / cosang;
  // slab test against the padded cube (double, as the reference's numpy)
  const float dn = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  const double lo[3] = {lo0, lo1, lo2}, hi[3] = {hi0, hi1, hi2};
  double nearv = -INFINITY, farv = INFINITY;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float u = d[a] / dn;
    if (u < 1e-5f && u > -1e-10f) u = 1e-5f;
    if (u > -1e-5f && u < 1e-10f) u = -1e-5f;
    const double inv = (double)(1.0f / u);
    const double ta = (lo[a] - (double)o[a]) * inv, tb = (hi[a] - (double)o[a]) * inv;
    nearv = fmax(nearv, fmin(ta, tb));
    farv = fmin(farv, fmax(ta, tb));
  }
  nearv = fmax(nearv, 0.1);
  const bool inside = nearv < farv;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    ray_o[i * 3 + a] = o[a];
    ray_d[i * 3 + a] = d[a];
    rgb[i * 3 + a] = inside ? colors[src * 3 + a] : 0.f;
  }
  depth[i] = inside ? dist : -0.001f;
  if (sem_row) {
    const int64_t s = inside ? semantic[src] : -1;
    sem_row[i] = s > 0 ? s + 1 : 0;   // class 0 and ignore (-1) -> the zero row of the table
  }
}

}  // namespace

extern "C" {

int pv2_ray_setup_record_sizes(int* scene_floats, int* view_floats) {
  *scene_floats = kSceneRec;
  *view_floats = kViewRec;
  return PV2_OK;
}

int pv2_unit_cube(const float* coords, const int64_t* offsets, int n_scenes, int64_t n_points,
                  int n_views, float z_level, const float* depth_scale, const float* extrinsic,
                  const float* kmat, float* bounds_ws, float* scene, float* extrinsic_out, float* view,
                  float* coords_out, pv2_stream_t stream) {
  PV2_REQUIRE(n_scenes >= 1 && n_scenes <= 64 && n_views >= 1 && n_scenes * n_views <= 256,
              "pv2_unit_cube: 1 <= scenes <= 64, scenes x views <= 256");
  PV2_REQUIRE(n_points >= 1, "pv2_unit_cube: empty cloud");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(scene_bounds_kernel, dim3(n_scenes), dim3(256), 0, s, coords, offsets, bounds_ws);
  hipLaunchKernelGGL(unit_cube_setup_kernel, dim3(1), dim3(256), n_scenes * 16 * sizeof(float), s,
                     bounds_ws, n_scenes, n_views, z_level, depth_scale, extrinsic, kmat, scene,
                     extrinsic_out, view);
  hipLaunchKernelGGL(unit_cube_points_kernel, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, s,
                     coords, offsets, n_scenes, n_points, scene, coords_out);
  return pv2::check_launch("unit_cube");
}

int pv2_ray_gen(const int64_t* pixels, int n_scenes, int n_views, int n_rays, int height, int width,
                const float* view, const float* scene, const float* colors, const float* depths,
                const int64_t* semantic, const double* bounds_lo, const double* bounds_hi, float* ray_o,
                float* ray_d, float* rgb, float* depth, int64_t* semantic_row, pv2_stream_t stream) {
  PV2_REQUIRE(n_scenes >= 1 && n_views >= 1 && n_rays >= 1 && height >= 1 && width >= 1,
              "pv2_ray_gen: bad sizes");
  const int64_t total = (int64_t)n_scenes * n_views * n_rays;
  hipLaunchKernelGGL(ray_gen_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, pixels, n_scenes, n_views, n_rays, height, width, view, scene,
                     colors, depths, semantic, bounds_lo[0], bounds_lo[1], bounds_lo[2], bounds_hi[0],
                     bounds_hi[1], bounds_hi[2], ray_o, ray_d, rgb, depth, semantic_row);
  return pv2::check_launch("ray_gen");
}

}  // extern "C"
