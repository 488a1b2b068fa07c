// 2 x 2 x 2 max-pooling of a channels-last dense grid on gfx950.
//
// Stands in for nn.MaxPool3d(kernel_size=2) of the projection network's encoders
// (ponder/models/ponder/unet3d.py:326-330 of the reference).  ATen has no channels-last 3-D pooling
// kernel: it transposes the (B, Z, Y, X, C) grid to NCDHW, pools, and transposes the gradient back
// (0.5 ms for the full-resolution level plus a strided ReLU backward behind it).
// Here the layout never changes: a thread owns 4 consecutive channels of one OUTPUT cell, reads its
// eight input cells as 16-byte pieces, keeps the maximum and a 3-bit window position per channel
// (first maximum in (z, y, x) window order wins, as max_pool3d_with_indices does); the backward
// writes EVERY input element once - the gradient where the window position matches, zero elsewhere
// - so there is nothing to clear and no atomics.  HBM-bound: 4*C bytes read per input cell forward,
// written per input cell backward.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void maxpool_cl_fwd_kernel(
    const float4* __restrict__ x, int B, int Z, int Y, int X, int c4, float4* __restrict__ y,
    uint32_t* __restrict__ idx) {
  const int Zo = Z >> 1, Yo = Y >> 1, Xo = X >> 1;
  const int64_t total = (int64_t)B * Zo * Yo * Xo * c4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int c = (int)(e % c4);
    int64_t cell = e / c4;
    const int xo = (int)(cell % Xo);
    cell /= Xo;
    const int yo = (int)(cell % Yo);
    cell /= Yo;
    const int zo = (int)(cell % Zo);
    const int b = (int)(cell / Zo);
    float4 best = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t pos = 0;  // 4 x 8 bits: window position of the maximum per channel
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const int z = 2 * zo + (w >> 2), yy = 2 * yo + ((w >> 1) & 1), xx = 2 * xo + (w & 1);
      const float4 v = x[((((int64_t)b * Z + z) * Y + yy) * X + xx) * c4 + c];
      if (w == 0) {
        best = v;
      } else {  // strictly greater (or NaN) replaces: the first maximum wins
        if (v.x > best.x || v.x != v.x) { best.x = v.x; pos = (pos & 0xffffff00u) | (uint32_t)w; }
        if (v.y > best.y || v.y != v.y) { best.y = v.y; pos = (pos & 0xffff00ffu) | ((uint32_t)w << 8); }
        if (v.z > best.z || v.z != v.z) { best.z = v.z; pos = (pos & 0xff00ffffu) | ((uint32_t)w << 16); }
        if (v.w > best.w || v.w != v.w) { best.w = v.w; pos = (pos & 0x00ffffffu) | ((uint32_t)w << 24); }
      }
    }
    y[e] = best;
    idx[e] = pos;
  }
}

// one thread per 4 channels of one INPUT cell
__global__ __launch_bounds__(256) void maxpool_cl_bwd_kernel(
    const float4* __restrict__ gy, const uint32_t* __restrict__ idx, int B, int Z, int Y, int X,
    int c4, const float4* __restrict__ addend, const float4* __restrict__ mask_src,
    float4* __restrict__ gx) {
  const int Zo = Z >> 1, Yo = Y >> 1, Xo = X >> 1;
  const int64_t total = (int64_t)B * Z * Y * X * c4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int c = (int)(e % c4);
    int64_t cell = e / c4;
    const int xx = (int)(cell % X);
    cell /= X;
    const int yy = (int)(cell % Y);
    cell /= Y;
    const int z = (int)(cell % Z);
    const int b = (int)(cell / Z);
    // (addend: the other gradient of the pooled tensor - an encoder level's output also feeds a
    // decoder level as its skip connection - summed here instead of by a separate pass)
    float4 g = addend != nullptr ? addend[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int zo = z >> 1, yo = yy >> 1, xo = xx >> 1;
    if (zo < Zo && yo < Yo && xo < Xo) {  // (odd trailing planes are outside every window)
      const int64_t o = ((((int64_t)b * Zo + zo) * Yo + yo) * Xo + xo) * c4 + c;
      const uint32_t w = (uint32_t)(((z & 1) << 2) | ((yy & 1) << 1) | (xx & 1));
      const uint32_t pos = idx[o];
      const float4 v = gy[o];
      if ((pos & 0xffu) == w) g.x += v.x;
      if (((pos >> 8) & 0xffu) == w) g.y += v.y;
      if (((pos >> 16) & 0xffu) == w) g.z += v.z;
      if ((pos >> 24) == w) g.w += v.w;
    }
    if (mask_src != nullptr) {   // the pooled tensor was a ReLU's output: its gradient passes where it was > 0
      const float4 m = mask_src[e];
      if (!(m.x > 0.f)) g.x = 0.f;
      if (!(m.y > 0.f)) g.y = 0.f;
      if (!(m.z > 0.f)) g.z = 0.f;
      if (!(m.w > 0.f)) g.w = 0.f;
    }
    gx[e] = g;
  }
}

}  // namespace

extern "C" {

int pv2_maxpool3d_cl_forward(const float* x, int B, int Z, int Y, int X, int C, float* y,
                             uint32_t* idx, pv2_stream_t stream) {
  PV2_REQUIRE(B >= 1 && Z >= 2 && Y >= 2 && X >= 2 && C >= 4 && (C % 4) == 0,
              "pv2_maxpool3d_cl_forward: needs C % 4 == 0 and at least one 2x2x2 window");
  const int64_t total = (int64_t)B * (Z / 2) * (Y / 2) * (X / 2) * (C / 4);
  hipLaunchKernelGGL(maxpool_cl_fwd_kernel, dim3(pv2::grid_for(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const float4*)x, B, Z, Y, X, C / 4, (float4*)y, idx);
  return pv2::check_launch("maxpool3d_cl_forward");
}

int pv2_maxpool3d_cl_backward(const float* grad_y, const uint32_t* idx, int B, int Z, int Y, int X,
                              int C, float* grad_x, pv2_stream_t stream) {
  PV2_REQUIRE(B >= 1 && Z >= 2 && Y >= 2 && X >= 2 && C >= 4 && (C % 4) == 0,
              "pv2_maxpool3d_cl_backward: needs C % 4 == 0 and at least one 2x2x2 window");
  const int64_t total = (int64_t)B * Z * Y * X * (C / 4);
  hipLaunchKernelGGL(maxpool_cl_bwd_kernel, dim3(pv2::grid_for(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const float4*)grad_y, idx, B, Z, Y, X, C / 4,
                     (const float4*)nullptr, (const float4*)nullptr, (float4*)grad_x);
  return pv2::check_launch("maxpool3d_cl_backward");
}

int pv2_maxpool3d_cl_backward_add(const float* grad_y, const uint32_t* idx, const float* addend,
                                  const float* relu_mask_src, int B, int Z, int Y, int X, int C,
                                  float* grad_x, pv2_stream_t stream) {
  PV2_REQUIRE(B >= 1 && Z >= 2 && Y >= 2 && X >= 2 && C >= 4 && (C % 4) == 0,
              "pv2_maxpool3d_cl_backward_add: needs C % 4 == 0 and at least one 2x2x2 window");
  const int64_t total = (int64_t)B * Z * Y * X * (C / 4);
  hipLaunchKernelGGL(maxpool_cl_bwd_kernel, dim3(pv2::grid_for(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const float4*)grad_y, idx, B, Z, Y, X, C / 4,
                     (const float4*)addend, (const float4*)relu_mask_src, (float4*)grad_x);
  return pv2::check_launch("maxpool3d_cl_backward_add");
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// Batched inverse of tiny matrices (n <= 4): the camera / scene transforms of the ray set-up
// (ponder/models/ponder/ponder_indoor_base.py:380-470 of the reference: torch.linalg.inv on (V, 4, 4)
// poses, (3, 3) intrinsics and the per-scene unit-cube transform).  The library route
// (rocSOLVER getrf + getri behind torch.linalg.inv_ex) is 11 launches per call for a dozen 4x4
// matrices; here one thread inverts one matrix by Gauss-Jordan elimination with partial pivoting in
// double precision and rounds the result to fp32 (at least as accurate as an fp32 LU).
namespace {

__global__ void small_inverse_kernel(const float* __restrict__ a, int64_t batch, int n,
                                     float* __restrict__ out) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= batch) return;
  double w[4][8];
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < n; ++c) {
      w[r][c] = (double)a[(m * n + r) * n + c];
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
    const double inv = 1.0 / w[col][col];   // (a singular input gives inf / nan, as the library does)
    for (int c = 0; c < 2 * n; ++c) w[col][c] *= inv;
    for (int r = 0; r < n; ++r) {
      if (r == col) continue;
      const double f = w[r][col];
      for (int c = 0; c < 2 * n; ++c) w[r][c] -= f * w[col][c];
    }
  }
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < n; ++c) out[(m * n + r) * n + c] = (float)w[r][n + c];
}

}  // namespace

extern "C" {

int pv2_small_inverse(const float* a, int64_t batch, int n, float* out, pv2_stream_t stream) {
  PV2_REQUIRE(n >= 1 && n <= 4 && batch >= 0, "pv2_small_inverse: 1 <= n <= 4");
  if (batch == 0) return PV2_OK;
  hipLaunchKernelGGL(small_inverse_kernel, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0,
                     (hipStream_t)stream, a, batch, n, out);
  return pv2::check_launch("small_inverse");
}

}  // extern "C"
