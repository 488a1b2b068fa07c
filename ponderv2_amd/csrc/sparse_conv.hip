// Sparse 3-D convolution arithmetic on gfx950: fused gather -> f32 MFMA -> scatter-add.
//
// Stands in for spconv 2.x's indice_conv / indice_conv_backward behind SubMConv3d / SparseConv3d /
// SparseInverseConv3d (reference call sites: ponder/models/sparse_unet/spconv_unet_v1m1_base.py
// :41,47,58,112,135,171).  One launch covers every kernel offset of a layer.
//
// Tiling (wave64, v_mfma_f32_32x32x2_f32 - exact fp32, 64 FLOP/clk/SIMD):
//   * a work item = one 32-pair tile of ONE kernel offset k  x  one group of NB 32-wide output
//     channel blocks; one wave per work item, 4 independent waves per workgroup;
//   * A operand = gathered input rows: lane (i, h) streams 16-byte pieces of row pair_in[p0+i]
//     (each active-voxel row is read as contiguous 32-byte runs by the lane pair h=0/1);
//   * B operand = W[n, k, :] rows in the spconv [Cout, K, Cin] layout, again 16 bytes per lane
//     along the reduction axis, so neither operand needs a transpose or an LDS round trip;
//   * the 32x32 result block has its 32 columns (output channels) on lanes 0..31, so the
//     scatter-add is 128-byte contiguous per output row.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int find_offset(const int32_t* __restrict__ tile_start, int K,
                                           int tile) {
  int lo = 0, hi = K;  // invariant: tile_start[lo] <= tile < tile_start[hi]
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (tile_start[mid] <= tile) lo = mid; else hi = mid;
  }
  return lo;
}

template <bool VEC>
__device__ __forceinline__ float4 load4(const float* __restrict__ row, int kk, int c, bool ok) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (VEC) {
    if (ok) v = *reinterpret_cast<const float4*>(row + kk);
  } else {
    if (ok) {
      if (kk + 0 < c) v.x = row[kk + 0];
      if (kk + 1 < c) v.y = row[kk + 1];
      if (kk + 2 < c) v.z = row[kk + 2];
      if (kk + 3 < c) v.w = row[kk + 3];
    }
  }
  return v;
}

// out[pair_out[p], n] += sum_c in[pair_in[p], c] * W[n, k, c]
template <int NB, bool VEC>
__global__ __launch_bounds__(256) void spconv_fwd_kernel(
    const float* __restrict__ X, int c_in, const float* __restrict__ W, int K, int c_out,
    const int32_t* __restrict__ pair_in, const int32_t* __restrict__ pair_out,
    const int32_t* __restrict__ kstart, const int32_t* __restrict__ tile_start, int64_t n_items,
    int n_groups, float* __restrict__ Y) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + wave;
  if (item >= n_items) return;
  const int tile = (int)(item / n_groups), grp = (int)(item % n_groups);
  const int k = find_offset(tile_start, K, tile);
  const int p0 = kstart[k] + (tile - tile_start[k]) * PV2_PAIR_TILE;
  const int pend = kstart[k + 1];
  const int i = lane & 31, h = lane >> 5;
  const int p = p0 + i;
  const bool pv = p < pend;
  const int row_in = pv ? pair_in[p] : 0;
  const int row_out = pv ? pair_out[p] : -1;
  const int n0 = grp * NB * 32;

  const float* xrow = X + (int64_t)row_in * c_in + 4 * h;
  const float* wrow[NB];
  bool wok[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    int n = n0 + nb * 32 + i;
    wok[nb] = n < c_out;
    wrow[nb] = W + ((int64_t)(wok[nb] ? n : 0) * K + k) * c_in + 4 * h;
  }

  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  for (int kk0 = 0; kk0 < c_in; kk0 += 8) {
    const float4 a = load4<VEC>(xrow, kk0, c_in - 4 * h, pv);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const float4 b = load4<VEC>(wrow[nb], kk0, c_in - 4 * h, wok[nb]);
      acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[nb], 0, 0, 0);
    }
  }

  // D[i'][j]: j = lane & 31 (output channel), i' = (r & 3) + 8 * (r >> 2) + 4 * h (pair in tile)
  int orow[16];
#pragma unroll
  for (int r = 0; r < 16; ++r)  // all shuffles before any divergent code
    orow[r] = __shfl(row_out, (r & 3) + 8 * (r >> 2) + 4 * h);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (orow[r] >= 0) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int n = n0 + nb * 32 + i;
        if (n < c_out) unsafeAtomicAdd(Y + (int64_t)orow[r] * c_out + n, acc[nb][r]);
      }
    }
  }
}

// dW[n, k, c] += sum_{p in tile} dY[pair_out[p], n] * X[pair_in[p], c]
// work item = (wgrad tile of PV2_WGRAD_TILE pairs of one k) x (32-wide n block) x (CB c blocks)
template <int CB>
__global__ __launch_bounds__(256) void spconv_wgrad_kernel(
    const float* __restrict__ X, int c_in, const float* __restrict__ dY, int c_out, int K,
    const int32_t* __restrict__ pair_in, const int32_t* __restrict__ pair_out,
    const int32_t* __restrict__ kstart, const int32_t* __restrict__ tile_start, int64_t n_items,
    int n_nblk, int n_cgrp, float* __restrict__ dW) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + wave;
  if (item >= n_items) return;
  const int per_tile = n_nblk * n_cgrp;
  const int tile = (int)(item / per_tile);
  const int sub = (int)(item % per_tile);
  const int nblk = sub / n_cgrp, cgrp = sub % n_cgrp;
  const int k = find_offset(tile_start, K, tile);
  const int p0 = kstart[k] + (tile - tile_start[k]) * PV2_WGRAD_TILE;
  const int pend = min(kstart[k + 1], p0 + PV2_WGRAD_TILE);
  const int i = lane & 31, h = lane >> 5;
  const int n = nblk * 32 + i;
  const bool nok = n < c_out;
  const int c0 = cgrp * CB * 32;

  f32x16 acc[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;

  for (int pb = p0; pb < pend; pb += 64) {
    // each lane fetches the indices of one pair of this 64-pair batch
    const int pl = pb + lane;
    const int my_in = pl < pend ? pair_in[pl] : -1;
    const int my_out = pl < pend ? pair_out[pl] : -1;
    const int steps = min(32, (pend - pb + 1) >> 1);
    for (int s = 0; s < steps; ++s) {
      const int rin = __shfl(my_in, 2 * s + h);
      const int rout = __shfl(my_out, 2 * s + h);
      const bool ok = rin >= 0;
      const float a = (ok && nok) ? dY[(int64_t)rout * c_out + n] : 0.f;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const int c = c0 + cb * 32 + i;
        const float b = (ok && c < c_in) ? X[(int64_t)rin * c_in + c] : 0.f;
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[cb], 0, 0, 0);
      }
    }
  }

  // D[i'][j]: i' = output channel n within block, j = input channel within block
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int np = nblk * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (np >= c_out) continue;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const int c = c0 + cb * 32 + i;
      if (c < c_in) unsafeAtomicAdd(dW + ((int64_t)np * K + k) * c_in + c, acc[cb][r]);
    }
  }
}

template <int NB>
int launch_fwd(const float* X, int c_in, const float* W, int K, int c_out, const int32_t* pi,
               const int32_t* po, const int32_t* ks, const int32_t* ts, int64_t n_tiles,
               float* Y, hipStream_t s) {
  const int n_groups = (c_out + NB * 32 - 1) / (NB * 32);
  const int64_t n_items = n_tiles * n_groups;
  const int64_t blocks = (n_items + 3) / 4;
  if (blocks > 0x7fffffffLL) {
    pv2::set_error("pv2_spconv_forward: grid too large");
    return PV2_E_BADARG;
  }
  const bool vec = (c_in % 8) == 0;
  if (vec)
    hipLaunchKernelGGL((spconv_fwd_kernel<NB, true>), dim3((unsigned)blocks), dim3(256), 0, s, X,
                       c_in, W, K, c_out, pi, po, ks, ts, n_items, n_groups, Y);
  else
    hipLaunchKernelGGL((spconv_fwd_kernel<NB, false>), dim3((unsigned)blocks), dim3(256), 0, s, X,
                       c_in, W, K, c_out, pi, po, ks, ts, n_items, n_groups, Y);
  return pv2::check_launch("spconv_fwd");
}

}  // namespace

extern "C" {

int pv2_spconv_forward(const float* in_feat, int64_t n_in, int c_in, const float* weight, int K,
                       int c_out, const int32_t* pair_in, const int32_t* pair_out,
                       const int32_t* kstart, const int32_t* tile_start, int64_t n_tiles,
                       float* out_feat, int64_t n_out, pv2_stream_t stream) {
  PV2_REQUIRE(c_in >= 1 && c_out >= 1 && K >= 1, "pv2_spconv_forward: bad channel/offset count");
  (void)n_in;
  (void)n_out;
  if (n_tiles == 0) return PV2_OK;
  hipStream_t s = (hipStream_t)stream;
  const int nblk = (c_out + 31) / 32;
  // Few tiles: split the output channels over more waves so the 1024 SIMDs have work.
  int nb = nblk >= 4 ? 4 : nblk;
  if (nb == 4 && n_tiles * ((nblk + 3) / 4) < 2048) nb = 2;
  switch (nb) {
    case 1: return launch_fwd<1>(in_feat, c_in, weight, K, c_out, pair_in, pair_out, kstart,
                                 tile_start, n_tiles, out_feat, s);
    case 2: return launch_fwd<2>(in_feat, c_in, weight, K, c_out, pair_in, pair_out, kstart,
                                 tile_start, n_tiles, out_feat, s);
    case 3: return launch_fwd<3>(in_feat, c_in, weight, K, c_out, pair_in, pair_out, kstart,
                                 tile_start, n_tiles, out_feat, s);
    default: return launch_fwd<4>(in_feat, c_in, weight, K, c_out, pair_in, pair_out, kstart,
                                  tile_start, n_tiles, out_feat, s);
  }
}

int pv2_spconv_backward_weight(const float* in_feat, int64_t n_in, int c_in, const float* dout,
                               int64_t n_out, int c_out, int K, const int32_t* pair_in,
                               const int32_t* pair_out, const int32_t* kstart,
                               const int32_t* tile_start, int64_t n_tiles, float* dweight,
                               pv2_stream_t stream) {
  PV2_REQUIRE(c_in >= 1 && c_out >= 1 && K >= 1, "pv2_spconv_backward_weight: bad sizes");
  (void)n_in;
  (void)n_out;
  if (n_tiles == 0) return PV2_OK;
  hipStream_t s = (hipStream_t)stream;
  const int n_nblk = (c_out + 31) / 32;
  const int cblk = (c_in + 31) / 32;
  const int cb = cblk >= 4 ? 4 : (cblk >= 2 ? 2 : 1);
  const int n_cgrp = (cblk + cb - 1) / cb;
  const int64_t n_items = n_tiles * n_nblk * n_cgrp;
  const int64_t blocks = (n_items + 3) / 4;
  if (blocks > 0x7fffffffLL) {
    pv2::set_error("pv2_spconv_backward_weight: grid too large");
    return PV2_E_BADARG;
  }
#define PV2_LAUNCH_WGRAD(CB)                                                                     \
  hipLaunchKernelGGL((spconv_wgrad_kernel<CB>), dim3((unsigned)blocks), dim3(256), 0, s, in_feat, \
                     c_in, dout, c_out, K, pair_in, pair_out, kstart, tile_start, n_items,       \
                     n_nblk, n_cgrp, dweight)
  if (cb == 4) PV2_LAUNCH_WGRAD(4);
  else if (cb == 2) PV2_LAUNCH_WGRAD(2);
  else PV2_LAUNCH_WGRAD(1);
#undef PV2_LAUNCH_WGRAD
  return pv2::check_launch("spconv_wgrad");
}

}  // extern "C"
