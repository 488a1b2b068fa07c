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
    int c4, float4* __restrict__ gx) {
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
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    const int zo = z >> 1, yo = yy >> 1, xo = xx >> 1;
    if (zo < Zo && yo < Yo && xo < Xo) {  // (odd trailing planes are outside every window)
      const int64_t o = ((((int64_t)b * Zo + zo) * Yo + yo) * Xo + xo) * c4 + c;
      const uint32_t w = (uint32_t)(((z & 1) << 2) | ((yy & 1) << 1) | (xx & 1));
      const uint32_t pos = idx[o];
      const float4 v = gy[o];
      if ((pos & 0xffu) == w) g.x = v.x;
      if (((pos >> 8) & 0xffu) == w) g.y = v.y;
      if (((pos >> 16) & 0xffu) == w) g.z = v.z;
      if ((pos >> 24) == w) g.w = v.w;
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
                     (float4*)grad_x);
  return pv2::check_launch("maxpool3d_cl_backward");
}

}  // extern "C"
