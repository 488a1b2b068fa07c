// First level of the dense projection network, evaluated from the occupied cells: the kernels around
// the sparse convolution.
//
// The reference scatters the backbone features into a dense (B, C, X, Y, Z) grid that is > 90 % empty
// and runs UNet3D's first "bcr" level (BatchNorm3d -> Conv3d 3x3x3 -> ReLU) on it
// (ponder/models/ponder/ponder_indoor_base.py:177-342 to_dense, unet3d.py:292-318 SingleConv/Encoder).
// BatchNorm is affine per channel, so the normalised grid is  y0 + [cell occupied] * x * scale  with
// y0 = beta - mean * scale, and the convolution splits into
//   * a CONSTANT part, the response to the field y0: it depends only on which of the 27 taps fall
//     inside the grid, i.e. on whether the position is first / last along each axis
//     (cells_expand_kernel writes it: u[tap][o] = sum_c W[o, c, tap] * y0[c] summed over the taps
//     inside), and
//   * an OCCUPIED part, a sparse convolution of the cell rows with W * scale (sparse_conv.hip), added
//     on top.
// Backward: the gradient of the constant part needs, per tap, the sum of the output gradient over
// the positions where that tap is inside - sums over all / first / last positions per axis
// (cells_class_sums_kernel, one pass over the gradient), folded per tap (cells_bwd_fold_kernel); the
// rest is the sparse convolution's grad-input / grad-weight and a BatchNorm backward that knows
// about the rows never stored (rownorm.hip pv2_bn_backward_padded).
// Python side: ponderv2_amd/cells_level.py; the composite of torch ops it replaces stays in
// ponder/models/ponder/sparse_input.py (CPU path and odd channel counts).
#include "common.h"

namespace {

constexpr int kTaps = 27;

// tbl[k * cap + i]: output row that cell i feeds through tap k (cross-correlation: out[p] += W_k .
// in[p + k - 1], so cell q reaches p = q - (k - 1)), or -1 (outside the grid / padding cell).
__global__ __launch_bounds__(256) void cells_tap_table_kernel(const int64_t* __restrict__ lin,
                                                              int64_t cap, int Z, int Y, int X,
                                                              int32_t* __restrict__ tbl) {
  const int64_t total = cap * kTaps;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int k = (int)(e / cap);
    const int64_t i = e - (int64_t)k * cap;
    const int64_t l = lin[i];
    int32_t row = -1;
    if (l >= 0) {
      const int x = (int)(l % X);
      const int64_t t = l / X;
      const int y = (int)(t % Y);
      const int z = (int)((t / Y) % Z);
      const int pz = z - (k / 9 - 1), py = y - ((k / 3) % 3 - 1), px = x - (k % 3 - 1);
      if (pz >= 0 && pz < Z && py >= 0 && py < Y && px >= 0 && px < X)
        row = (int32_t)(l + ((int64_t)(pz - z) * Y + (py - y)) * X + (px - x));
    }
    tbl[e] = row;
  }
}

// One workgroup per tap k: the [c_out, 27, c_in] weight layouts of the sparse kernels - plain (for
// the grad-input pass) and with the BatchNorm scale folded in (forward) - and the constant response
// u[k][o] = sum_c W[o, c, k] * y0[c].  affine = [scale(c_in) | y0(c_in)].
__global__ __launch_bounds__(256) void cells_fold_kernel(
    const float* __restrict__ W, int64_t so, int64_t sc, int64_t sz, int64_t sy, int64_t sx,
    int c_out, int c_in, const float* __restrict__ affine, float* __restrict__ w_okc,
    float* __restrict__ ws_okc, float* __restrict__ u) {
  __shared__ float s_part[256];
  const int k = blockIdx.x, tid = threadIdx.x;
  const int64_t woff = (k / 9) * sz + ((k / 3) % 3) * sy + (k % 3) * sx;
  const float* scale = affine;
  const float* y0 = affine + c_in;
  for (int e = tid; e < c_out * c_in; e += 256) {
    const int o = e / c_in, c = e - o * c_in;
    const float w = W[o * so + c * sc + woff];
    const int64_t dst = ((int64_t)o * kTaps + k) * c_in + c;
    w_okc[dst] = w;
    ws_okc[dst] = w * scale[c];
  }
  // u: `parts` threads share an output channel (c_out <= 256), partial sums added in a fixed order
  const int parts = 256 / c_out;
  const int o = tid % c_out, part = tid / c_out;
  float acc = 0.f;
  if (part < parts)
    for (int c = part; c < c_in; c += parts) acc += W[o * so + c * sc + woff] * y0[c];
  s_part[tid] = acc;
  __syncthreads();
  if (tid < c_out) {
    float t = 0.f;
    for (int q = 0; q < parts; ++q) t += s_part[q * c_out + tid];
    u[k * c_out + tid] = t;
  }
}

// tap i of a size-3 kernel reads inside the grid at a position that is first (bit 0) / last (bit 1)
__device__ __forceinline__ bool tap_inside(int i, int cls) {
  return i == 1 || (i == 0 ? !(cls & 1) : !(cls & 2));
}

// out[row, :] = bias + sum over the taps inside at `row` of u[tap, :]: the constant part, a pure
// write of the (B, Z, Y, X, c_out) channels-last grid.  A workgroup takes whole x-lines: (z, y) fix
// the class along two axes, the four x classes (interior / first / last / both) are built per line.
__global__ __launch_bounds__(256) void cells_expand_kernel(const float* __restrict__ u,
                                                           const float* __restrict__ bias, int B,
                                                           int Z, int Y, int X, int c_out,
                                                           float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float s_tab[4 * 256];   // [x class][c_out <= 256]
  const int tid = threadIdx.x;
  const int c4 = c_out >> 2;
  const int64_t lines = (int64_t)B * Z * Y;
  for (int64_t line = blockIdx.x; line < lines; line += gridDim.x) {
    const int y = (int)(line % Y);
    const int z = (int)((line / Y) % Z);
    const int cz = (z == 0 ? 1 : 0) | (z == Z - 1 ? 2 : 0);
    const int cy = (y == 0 ? 1 : 0) | (y == Y - 1 ? 2 : 0);
    __syncthreads();   // (the previous line's readers are done with the table)
    for (int e = tid; e < 4 * c_out; e += 256) {
      const int cx = e / c_out, o = e - cx * c_out;
      float t = bias ? bias[o] : 0.f;
      for (int i = 0; i < 3; ++i) {
        if (!tap_inside(i, cz)) continue;
        for (int j = 0; j < 3; ++j) {
          if (!tap_inside(j, cy)) continue;
          for (int k = 0; k < 3; ++k)
            if (tap_inside(k, cx)) t += u[((i * 3 + j) * 3 + k) * c_out + o];
        }
      }
      s_tab[cx * c_out + o] = t;
    }
    __syncthreads();
    float4* dst = reinterpret_cast<float4*>(out + line * X * c_out);
    for (int e = tid; e < X * c4; e += 256) {
      const int x = e / c4, q = e - x * c4;
      const int cx = (x == 0 ? 1 : 0) | (x == X - 1 ? 2 : 0);
      dst[e] = *reinterpret_cast<const float4*>(&s_tab[cx * c_out + 4 * q]);
    }
  }
}

// Sums of the gradient rows of one (b, z, y-chunk) slab over [all | first | last] positions along y
// times [all | first | last] along x: partial[wg][9][c_out], no atomics, fixed order.
__global__ __launch_bounds__(256) void cells_class_sums_kernel(const float* __restrict__ g, int Z,
                                                               int Y, int X, int c_out, int ychunk,
                                                               int nyc, float* __restrict__ partial) {
  __shared__ float s_red[9 * 1024];
  const int tid = threadIdx.x;
  const int c4 = c_out >> 2;
  const int lanes = 256 / c4;                 // rows in flight per pass (c_out <= 256... c4 <= 64)
  const int q = tid % c4, lr = tid / c4;
  const int wg = blockIdx.x;
  const int yc = wg % nyc;
  const int64_t bz = wg / nyc;                // b * Z + z
  const int y_lo = yc * ychunk, y_hi = min(Y, y_lo + ychunk);
  const int rows = (y_hi - y_lo) * X;
  float4 acc[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) acc[a][b] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* src = reinterpret_cast<const float4*>(g + (bz * Y + y_lo) * (int64_t)X * c_out);
  if (lr < lanes) {
    auto one = [&](int r, const float4& v) __attribute__((always_inline)) {
      const int yy = y_lo + r / X, x = r % X;
      const float fy[3] = {1.f, yy == 0 ? 1.f : 0.f, yy == Y - 1 ? 1.f : 0.f};
      const float fx[3] = {1.f, x == 0 ? 1.f : 0.f, x == X - 1 ? 1.f : 0.f};
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const float f = fy[a] * fx[b];
          acc[a][b].x += v.x * f;
          acc[a][b].y += v.y * f;
          acc[a][b].z += v.z * f;
          acc[a][b].w += v.w * f;
        }
    };
    int r = lr;
    for (; r + 3 * lanes < rows; r += 4 * lanes) {   // four rows in flight per thread
      const float4 v0 = src[(int64_t)r * c4 + q], v1 = src[(int64_t)(r + lanes) * c4 + q];
      const float4 v2 = src[(int64_t)(r + 2 * lanes) * c4 + q], v3 = src[(int64_t)(r + 3 * lanes) * c4 + q];
      one(r, v0);
      one(r + lanes, v1);
      one(r + 2 * lanes, v2);
      one(r + 3 * lanes, v3);
    }
    for (; r < rows; r += lanes) one(r, src[(int64_t)r * c4 + q]);
  }
  // lanes of a column group are added in lane order
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      float* d = &s_red[(a * 3 + b) * 1024 + tid * 4];
      d[0] = acc[a][b].x, d[1] = acc[a][b].y, d[2] = acc[a][b].z, d[3] = acc[a][b].w;
    }
  __syncthreads();
  for (int e = tid; e < 9 * c_out; e += 256) {
    const int ab = e / c_out, o = e - ab * c_out;
    float t = 0.f;
    for (int l = 0; l < lanes; ++l) t += s_red[ab * 1024 + (l * c4 + (o >> 2)) * 4 + (o & 3)];
    partial[((int64_t)wg * 9 + ab) * c_out + o] = t;
  }
}

// One workgroup per tap (i, j, k): g_u[tap][o] = sum over the positions where the tap reads inside
// of the output gradient - from the slab sums: along z the slab's own position decides, along y / x
// inside(0, .) = all - first, inside(1, .) = all, inside(2, .) = all - last - and the tap's share of
// d y0:  gy0_part[tap][c] = sum_o g_u[tap][o] * W[o, c, tap].
__global__ __launch_bounds__(256) void cells_bwd_fold_kernel(
    const float* __restrict__ partial, int n_wg, int Z, int nyc, int c_out,
    const float* __restrict__ W, int64_t so, int64_t sc, int64_t sz, int64_t sy, int64_t sx, int c_in,
    float* __restrict__ gu, float* __restrict__ gy0_part) {
  __shared__ double s_part[256];
  __shared__ float s_gu[256];
  const int tap = blockIdx.x, tid = threadIdx.x;
  const int i = tap / 9, j = (tap / 3) % 3, k = tap % 3;
  const int subs = 256 / c_out;
  const int o = tid % c_out, sub = tid / c_out;
  double acc = 0.0;
  if (sub < subs) {
    for (int wg = sub; wg < n_wg; wg += subs) {
      const int z = (wg / nyc) % Z;
      if ((i == 0 && z == 0) || (i == 2 && z == Z - 1)) continue;   // tap outside along z
      const float* p = partial + (int64_t)wg * 9 * c_out + o;
      // rows a = y class (0 all, 1 first, 2 last), columns b = x class
      auto row = [&](int a) __attribute__((always_inline)) {
        double t = (double)p[(a * 3 + 0) * c_out];
        if (k == 0) t -= (double)p[(a * 3 + 1) * c_out];
        if (k == 2) t -= (double)p[(a * 3 + 2) * c_out];
        return t;
      };
      double t = row(0);
      if (j == 0) t -= row(1);
      if (j == 2) t -= row(2);
      acc += t;
    }
  }
  s_part[tid] = acc;
  __syncthreads();
  if (tid < c_out) {
    double t = 0.0;
    for (int q = 0; q < subs; ++q) t += s_part[q * c_out + tid];
    s_gu[tid] = (float)t;
    gu[tap * c_out + tid] = (float)t;
  }
  __syncthreads();
  const int64_t woff = i * sz + j * sy + k * sx;
  for (int c = tid; c < c_in; c += 256) {
    float t = 0.f;
    for (int oo = 0; oo < c_out; ++oo) t += s_gu[oo] * W[oo * so + c * sc + woff];
    gy0_part[tap * c_in + c] = t;
  }
}

// dW[o, c, tap] = dWs[o, tap, c] * scale[c] + g_u[tap][o] * y0[c], written with the parameter's strides
__global__ __launch_bounds__(256) void cells_dw_finish_kernel(
    const float* __restrict__ dws_okc, const float* __restrict__ gu, const float* __restrict__ affine,
    int c_out, int c_in, float* __restrict__ dW, int64_t so, int64_t sc, int64_t sz, int64_t sy,
    int64_t sx) {
  const int total = c_out * kTaps * c_in;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int c = e % c_in;
    const int t = e / c_in;
    const int tap = t % kTaps, o = t / kTaps;
    const float v = dws_okc[e] * affine[c] + gu[tap * c_out + o] * affine[c_in + c];
    dW[o * so + c * sc + (tap / 9) * sz + ((tap / 3) % 3) * sy + (tap % 3) * sx] = v;
  }
}

}  // namespace

extern "C" {

int pv2_cells_tap_table(const int64_t* lin, int64_t cap, int z, int y, int x, int32_t* table,
                        pv2_stream_t stream) {
  PV2_REQUIRE(lin != nullptr && table != nullptr, "cells_tap_table: null pointer");
  PV2_REQUIRE(cap >= 0 && z > 0 && y > 0 && x > 0, "cells_tap_table: bad shape");
  if (cap == 0) return PV2_OK;
  hipLaunchKernelGGL(cells_tap_table_kernel, dim3(pv2::grid_for(cap * kTaps, 256)), dim3(256), 0,
                     (hipStream_t)stream, lin, cap, z, y, x, table);
  return pv2::check_launch("cells_tap_table");
}

int pv2_cells_fold_weights(const float* weight, int64_t s_out, int64_t s_in, int64_t s_z, int64_t s_y,
                           int64_t s_x, int c_out, int c_in, const float* affine, float* w_okc,
                           float* ws_okc, float* u, pv2_stream_t stream) {
  PV2_REQUIRE(weight != nullptr && affine != nullptr && w_okc != nullptr && ws_okc != nullptr &&
                  u != nullptr, "cells_fold_weights: null pointer");
  PV2_REQUIRE(c_out >= 1 && c_out <= 256 && c_in >= 1, "cells_fold_weights: 1..256 output channels");
  hipLaunchKernelGGL(cells_fold_kernel, dim3(kTaps), dim3(256), 0, (hipStream_t)stream, weight, s_out,
                     s_in, s_z, s_y, s_x, c_out, c_in, affine, w_okc, ws_okc, u);
  return pv2::check_launch("cells_fold_weights");
}

int pv2_cells_expand(const float* u, const float* bias_or_null, int b, int z, int y, int x, int c_out,
                     float* out, pv2_stream_t stream) {
  PV2_REQUIRE(u != nullptr && out != nullptr, "cells_expand: null pointer");
  PV2_REQUIRE(b > 0 && z > 0 && y > 0 && x > 0 && c_out >= 4 && c_out <= 256 && (c_out % 4) == 0,
              "cells_expand: c_out must be a multiple of 4, at most 256");
  const int64_t lines = (int64_t)b * z * y;
  const int grid = (int)(lines < 2048 ? lines : 2048);
  hipLaunchKernelGGL(cells_expand_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, u,
                     bias_or_null, b, z, y, x, c_out, out);
  return pv2::check_launch("cells_expand");
}

static int class_sum_chunks(int y, int* ychunk) {
  // ~16 x-lines per slab: 512 workgroups at the ScanNet grid (2 x 32 x 128 lines of 128 cells)
  *ychunk = y < 16 ? y : 16;
  return (y + *ychunk - 1) / *ychunk;
}

int64_t pv2_cells_backward_workspace_floats(int b, int z, int y, int c_out) {
  int ychunk;
  const int nyc = class_sum_chunks(y, &ychunk);
  return (int64_t)b * z * nyc * 9 * c_out;
}

// g: (B, Z, Y, X, c_out) gradient rows.  Out: gu[27][c_out] (gradient of the constant responses) and
// gy0_parts[27][c_in] (their shares of d y0, to be added in tap order).
int pv2_cells_backward_table(const float* g, int b, int z, int y, int x, int c_out,
                             const float* weight, int64_t s_out, int64_t s_in, int64_t s_z,
                             int64_t s_y, int64_t s_x, int c_in, float* workspace, float* gu,
                             float* gy0_parts, pv2_stream_t stream) {
  PV2_REQUIRE(g != nullptr && weight != nullptr && workspace != nullptr && gu != nullptr &&
                  gy0_parts != nullptr, "cells_backward_table: null pointer");
  PV2_REQUIRE(b > 0 && z > 0 && y > 0 && x > 0 && c_in >= 1, "cells_backward_table: bad shape");
  PV2_REQUIRE(c_out >= 4 && c_out <= 256 && (c_out % 4) == 0 && (256 % (c_out / 4)) == 0,
              "cells_backward_table: c_out / 4 must divide 256");
  int ychunk;
  const int nyc = class_sum_chunks(y, &ychunk);
  const int64_t n_wg = (int64_t)b * z * nyc;
  PV2_REQUIRE(n_wg < 0x7fffffffLL, "cells_backward_table: grid too large");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(cells_class_sums_kernel, dim3((unsigned)n_wg), dim3(256), 0, s, g, z, y, x, c_out,
                     ychunk, nyc, workspace);
  hipLaunchKernelGGL(cells_bwd_fold_kernel, dim3(kTaps), dim3(256), 0, s, workspace, (int)n_wg, z, nyc,
                     c_out, weight, s_out, s_in, s_z, s_y, s_x, c_in, gu, gy0_parts);
  return pv2::check_launch("cells_backward_table");
}

int pv2_cells_dw_finish(const float* dws_okc, const float* gu, const float* affine, int c_out, int c_in,
                        float* dweight, int64_t s_out, int64_t s_in, int64_t s_z, int64_t s_y,
                        int64_t s_x, pv2_stream_t stream) {
  PV2_REQUIRE(dws_okc != nullptr && gu != nullptr && affine != nullptr && dweight != nullptr,
              "cells_dw_finish: null pointer");
  PV2_REQUIRE(c_out >= 1 && c_in >= 1, "cells_dw_finish: bad shape");
  hipLaunchKernelGGL(cells_dw_finish_kernel, dim3(pv2::grid_for((int64_t)c_out * kTaps * c_in, 256)),
                     dim3(256), 0, (hipStream_t)stream, dws_okc, gu, affine, c_out, c_in, dweight, s_out,
                     s_in, s_z, s_y, s_x);
  return pv2::check_launch("cells_dw_finish");
}

}  // extern "C"
