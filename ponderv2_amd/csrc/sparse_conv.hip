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
#include <stdlib.h>

#include "common.h"
#include "mfma_split.h"

namespace {

using pv2::f32x16;

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

__device__ __forceinline__ float4 ld4(const float* __restrict__ p, bool ok) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) v = *reinterpret_cast<const float4*>(p);
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
  if (tile >= tile_start[K]) return;  // grid sized from an upper bound of the pair counts
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
  if (tile >= tile_start[K]) return;  // grid sized from an upper bound of the pair counts
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


// ------------------------------------------------------------------------------------------
// v2 kernels: workgroup-cooperative, LDS-staged and software-pipelined.
// ------------------------------------------------------------------------------------------

// Forward / grad-input, c_in % 32 == 0.  A workgroup (4 waves) owns 128 pairs of ONE offset k and
// NT = 32*NB output channels.  The weight slab W[n0:n0+NT, k, kk0:kk0+32] is shared by the four
// waves through a double-buffered LDS tile (rows padded by 16 B: conflict-free ds_read_b128);
// every wave streams its own 32 gathered feature rows straight into registers (they are private
// to the wave's MFMA A operand, so an LDS round trip would buy nothing).  The loads of K-slab
// t+1 (global -> registers) are issued before the 16*NB MFMAs of slab t; the LDS write of the
// prefetched weights follows the MFMAs; one barrier per slab.
constexpr int kFwdTile = 128;
constexpr int kKC = 32;
constexpr int kWPad = kKC + 4;

// TRANS: W is the FORWARD weight [c_in, K, c_out] of the conv whose grad-input this launch computes
// (reduction axis outermost): slabs are staged reduction-major in LDS (coalesced 16-byte reads
// along the output-channel axis, no transposed copy of the weights in HBM) and the B fragment is
// four 4-byte LDS reads instead of one 16-byte read.
// SPLIT: the products run on the bf16 matrix cores (mfma_split.h): every operand is cut into three
// bf16 pieces - the gathered rows in registers, the weight slab on its way into LDS, where it lives as
// three planes of [NT rows][32 bf16 + 16 bytes of padding] (conflict-free ds_read_b128 of the eight
// reduction steps a lane owns) - and six v_mfma_f32_32x32x16_bf16 per 16 reduction steps and column
// block replace sixteen fp32 MFMAs of half the length: the same sums to within fp32 rounding.
constexpr int kRowDw = 20;   // dwords per row of a piece plane
template <int NB, bool TRANS, bool SPLIT>
__global__ __launch_bounds__(256) void spconv_fwd_lds_kernel(
    const float* __restrict__ X, int c_in, const float* __restrict__ W, int K, int c_out,
    const int32_t* __restrict__ pair_in, const int32_t* __restrict__ pair_out,
    const int32_t* __restrict__ kstart, const int32_t* __restrict__ tile_start, int n_groups,
    float* __restrict__ Y, int tile_base, int skip_lo, int skip_len, int store) {
  constexpr int NT = 32 * NB;
  constexpr int LDT = NT + 4;  // row stride of the reduction-major (TRANS) slab
  // two weight slabs; the product-row epilogue re-uses the space as four 32 x 32 staging tiles
  constexpr int kSlabFloats = SPLIT ? 3 * NT * kRowDw : NT * kWPad;
  constexpr int kStageFloats = 32 * kWPad;
  constexpr int kLdsFloats = 2 * kSlabFloats > 4 * kStageFloats ? 2 * kSlabFloats : 4 * kStageFloats;
  __shared__ __attribute__((aligned(16))) float sBuf[kLdsFloats];
  float(*sW)[kSlabFloats] = reinterpret_cast<float(*)[kSlabFloats]>(sBuf);
  static_assert(SPLIT || kKC * LDT <= NT * kWPad, "TRANS slab fits the same buffer");
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int tile = blockIdx.x / n_groups + tile_base;
  if (tile >= skip_lo) tile += skip_len;  // the tiles of the centre offset ran in the store pass
  const int grp = blockIdx.x % n_groups;
  int k, p0, pend;
  if (K < 64) {
    // offset of this tile: ONE round of loads - lane l holds the l-th entries of both prefixes, the
    // offset is the last lane whose tile prefix is <= tile (a ballot), its pair range comes from
    // lane reads - instead of a five-step binary search of dependent loads: every workgroup of the
    // launch sits in this prologue at the same time, nothing overlaps it
    const int ts_l = lane <= K ? tile_start[lane] : 0x7fffffff;
    const int ks_l = lane <= K ? kstart[lane] : 0;
    if (tile >= __builtin_amdgcn_readlane(ts_l, K)) return;  // grid sized from an upper bound
    const unsigned long long m = __ballot(ts_l <= tile);
    k = 63 - __builtin_clzll(m);
    p0 = __builtin_amdgcn_readlane(ks_l, k) + (tile - __builtin_amdgcn_readlane(ts_l, k)) * kFwdTile;
    pend = __builtin_amdgcn_readlane(ks_l, k + 1);
  } else {
    if (tile >= tile_start[K]) return;
    k = find_offset(tile_start, K, tile);
    p0 = kstart[k] + (tile - tile_start[k]) * kFwdTile;
    pend = kstart[k + 1];
  }
  const int i = lane & 31, h = lane >> 5;
  const int p = p0 + wave * 32 + i;
  const bool pv = p < pend;
  const int row_in = pv ? pair_in[p] : 0;
  // store == 2: PRODUCT-ROW mode - the result of pair p is row p of Y (one row per pair, plain
  // coalesced stores; the rows of one output voxel are summed in a fixed order by
  // row_reduce_kernel, sparse_conv_pr.hip).  Otherwise rows go to pair_out[p].
  const int row_out = pv ? (store == 2 ? p : pair_out[p]) : -1;
  const int n0 = grp * NT;
  const float* xrow = X + (int64_t)row_in * c_in + 4 * h;

  // weight staging: NT rows x 8 float4 per slab; thread t owns float4 q = t + 256*u, i.e. row
  // (t >> 3) + 32*u and 16-byte column t & 7
  // plain layout: thread t stages float4 (row r0 + 32u, 16-byte column c4) of the [NT x 32] slab;
  // TRANS: float4 q = t + 256u of the [32 x NT] slab: reduction row q / (NT/4), columns 4 (q % (NT/4))
  const int r0 = tid >> 3, c4 = tid & 7;
  const float* wbase = W + ((int64_t)(n0 + r0) * K + k) * c_in + 4 * c4;
  const int64_t wstride = (int64_t)32 * K * c_in;
  const int wdst0 = r0 * kWPad + 4 * c4;
  auto load_w = [&](int u, int kk) -> float4 {
    if (!TRANS) return ld4(wbase + u * wstride + kk, n0 + r0 + 32 * u < c_out);
    const int q = tid + 256 * u;
    const int rr = q / (NT / 4), nn = n0 + 4 * (q % (NT / 4));
    return ld4(W + ((int64_t)(kk + rr) * K + k) * c_out + nn, nn < c_out);
  };
  auto store_w = [&](int buf, int u, const float4& v) {
    if (!TRANS) {
      *reinterpret_cast<float4*>(&sW[buf][wdst0 + u * 32 * kWPad]) = v;
    } else {
      const int q = tid + 256 * u;
      *reinterpret_cast<float4*>(&sW[buf][(q / (NT / 4)) * LDT + 4 * (q % (NT / 4))]) = v;
    }
  };

  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  const int nslab = c_in / kKC;
  if constexpr (SPLIT) {
    unsigned* sP = reinterpret_cast<unsigned*>(sBuf);
    // A: lane (i, h) owns reduction steps 16 st + 8 h .. + 7 of its row for st = 0, 1
    const float* xrow2 = X + (int64_t)row_in * c_in + 8 * h;
    // weight items.  Plain: float4 (row r0 + 32 u, steps 4 c4 .. + 3) -> two dwords per piece at
    // [row][2 c4].  TRANS: item q = tid + 256 u < 128 NB is the step pair rr2 = q & 15 of the four output
    // channels 4 (q >> 4) .. + 3: two float4 (steps 2 rr2 and 2 rr2 + 1), one dword per piece and channel
    // at [channel][rr2] - lanes run along the steps, so the transposing writes spread over all banks
    constexpr int NW = TRANS ? (NB + 1) / 2 : NB;
    // Register sets of the slabs in flight: with the six bf16 MFMAs a 32-step slab is ~0.6 us of matrix
    // work per workgroup - less than one trip to L2 / HBM -, so the loads run TWO slabs ahead: while slab
    // t is multiplied, slab t + 1 has arrived (its weights go to LDS at the end of the iteration) and
    // slab t + 2 is being fetched into the set slab t came from.
    struct Slab {
      float4 a[4];
      float4 wa[NW], wb[NW];
    };
    Slab sl[2];
    // (UNCONDITIONAL loads: a predicated load is a branch around it, and the s_waitcnt pass merges the
    // two paths to vmcnt(0) - waiting for the slab just requested.  Pairs past the end read row 0,
    // weight rows / columns past c_out are clamped to the last ones: none of it is stored.)
    auto ldu4 = [](const float* q) __attribute__((always_inline)) { return *reinterpret_cast<const float4*>(q); };
    const float* wrow[NW];
#pragma unroll
    for (int u = 0; u < NW; ++u) {
      if (!TRANS) {
        wrow[u] = W + ((int64_t)min(n0 + r0 + 32 * u, c_out - 1) * K + k) * c_in + 4 * c4;
      } else {
        const int q = min(tid + 256 * u, 128 * NB - 1);
        wrow[u] = W + ((int64_t)(2 * (q & 15)) * K + k) * c_out + min(n0 + 4 * (q >> 4), c_out - 4);
      }
    }
    auto load_slab = [&](Slab& d, int kk) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < 4; ++s) d.a[s] = ldu4(xrow2 + kk + 16 * (s >> 1) + 4 * (s & 1));
#pragma unroll
      for (int u = 0; u < NW; ++u) {
        if (!TRANS) {
          d.wa[u] = ldu4(wrow[u] + kk);
        } else {
          const float* src = wrow[u] + (int64_t)kk * K * c_out;
          d.wa[u] = ldu4(src);
          d.wb[u] = ldu4(src + (int64_t)K * c_out);
        }
      }
    };
    auto store_items = [&](const Slab& d, int buf) __attribute__((always_inline)) {
      unsigned* dst = sP + buf * kSlabFloats;
#pragma unroll
      for (int u = 0; u < NW; ++u) {
        if (!TRANS) {
          const float x[4] = {d.wa[u].x, d.wa[u].y, d.wa[u].z, d.wa[u].w};
          float r1[4], r2[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) r1[j] = pv2::bf16_rest(x[j]), r2[j] = pv2::bf16_rest(r1[j]);
          unsigned* o = dst + (r0 + 32 * u) * kRowDw + 2 * c4;
          *reinterpret_cast<uint2*>(o) = make_uint2(pv2::pack_hi(x[0], x[1]), pv2::pack_hi(x[2], x[3]));
          *reinterpret_cast<uint2*>(o + NT * kRowDw) =
              make_uint2(pv2::pack_hi(r1[0], r1[1]), pv2::pack_hi(r1[2], r1[3]));
          *reinterpret_cast<uint2*>(o + 2 * NT * kRowDw) =
              make_uint2(pv2::pack_hi(r2[0], r2[1]), pv2::pack_hi(r2[2], r2[3]));
        } else {
          const int q = tid + 256 * u;
          if (q < 128 * NB) {
            const int rr2 = q & 15, nl = 4 * (q >> 4);
            const float xa[4] = {d.wa[u].x, d.wa[u].y, d.wa[u].z, d.wa[u].w};
            const float xb[4] = {d.wb[u].x, d.wb[u].y, d.wb[u].z, d.wb[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float a1 = pv2::bf16_rest(xa[j]), b1 = pv2::bf16_rest(xb[j]);
              unsigned* o = dst + (nl + j) * kRowDw + rr2;
              o[0] = pv2::pack_hi(xa[j], xb[j]);
              o[NT * kRowDw] = pv2::pack_hi(a1, b1);
              o[2 * NT * kRowDw] = pv2::pack_hi(pv2::bf16_rest(a1), pv2::bf16_rest(b1));
            }
          }
        }
      }
    };
    float4 a_cur[4];
    // one iteration: multiply slab t (A in a_cur, weights in LDS buffer t & 1); `nxt` holds slab t + 1,
    // `far` (the set slab t came from) receives slab t + 2
    // (no branch between the loads and their use: slabs past the end are clamped to the last one -
    // a redundant load and LDS store at the tail - so that the compiler's own s_waitcnt counts the
    // loads of slab t + 2 as still in flight when slab t + 1 is consumed; a conditional prefetch
    // merges to vmcnt(0) at the join)
    auto iteration = [&](int t, Slab& nxt, Slab& far) __attribute__((always_inline)) {
      const int buf = t & 1;
      load_slab(far, min(t + 2, nslab - 1) * kKC);
      __builtin_amdgcn_sched_barrier(0);   // (those loads stay in front of the MFMAs)
      const unsigned* src = sP + buf * kSlabFloats + i * kRowDw + 4 * h;
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const pv2::Split8 pa = pv2::split8(a_cur[2 * st], a_cur[2 * st + 1]);
        pv2::bf16x8 pb[NB][3];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int pc = 0; pc < 3; ++pc)
            pb[nb][pc] = *reinterpret_cast<const pv2::bf16x8*>(src + (pc * NT + nb * 32) * kRowDw + 8 * st);
        // the six terms, smallest first, each ROUND-ROBIN over the NB accumulators
#define PV2_TERM(ta, tb)                  \
  _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) acc[nb] = pv2::mfma_bf16(pa.p[ta], pb[nb][tb], acc[nb]);
        PV2_SPLIT_TERMS(PV2_TERM)
#undef PV2_TERM
      }
      __builtin_amdgcn_sched_barrier(0);
      store_items(nxt, buf ^ 1);
#pragma unroll
      for (int s = 0; s < 4; ++s) a_cur[s] = nxt.a[s];
      __syncthreads();
    };
    load_slab(sl[0], 0);
    store_items(sl[0], 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) a_cur[s] = sl[0].a[s];
    load_slab(sl[1], min(1, nslab - 1) * kKC);
    __syncthreads();
    int t = 0;
    for (; t + 1 < nslab; t += 2) {
      iteration(t, sl[1], sl[0]);
      iteration(t + 1, sl[0], sl[1]);
    }
    if (t < nslab) iteration(t, sl[1], sl[0]);
  } else {
  float4 a_cur[4], a_nxt[4], w_nxt[NB];
#pragma unroll
  for (int s = 0; s < 4; ++s) a_cur[s] = ld4(xrow + 8 * s, pv);
#pragma unroll
  for (int u = 0; u < NB; ++u) store_w(0, u, load_w(u, 0));
  __syncthreads();

  for (int t = 0; t < nslab; ++t) {
    const int buf = t & 1;
    const bool more = (t + 1) < nslab;
    if (more) {
      const int kk = (t + 1) * kKC;
#pragma unroll
      for (int s = 0; s < 4; ++s)
        a_nxt[s] = ld4(xrow + kk + 8 * s, pv);
#pragma unroll
      for (int u = 0; u < NB; ++u) w_nxt[u] = load_w(u, kk);
    }
    // the NB weight fragments of reduction step s + 1 are read from LDS while the 4 * NB MFMAs of step
    // s run (ping-pong sets, reads interleaved with the MFMAs); the MFMAs go ROUND-ROBIN over the NB
    // accumulators: consecutive MFMAs of a wave never wait for each other's result
    auto read_b = [&](int s, float4 (&b)[NB]) __attribute__((always_inline)) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        if (!TRANS) {
          b[nb] = *reinterpret_cast<const float4*>(&sW[buf][(nb * 32 + i) * kWPad + 8 * s + 4 * h]);
        } else {
          const float* col = &sW[buf][(8 * s + 4 * h) * LDT + nb * 32 + i];
          b[nb] = make_float4(col[0], col[LDT], col[2 * LDT], col[3 * LDT]);
        }
      }
    };
    __builtin_amdgcn_sched_barrier(0);   // (the loads of slab t + 1 stay in front of the MFMAs)
    float4 bb[2][NB];
    read_b(0, bb[0]);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int cur = s & 1;
      if (s + 1 < 4) read_b(s + 1, bb[cur ^ 1]);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s].x, bb[cur][nb].x, acc[nb], 0, 0, 0);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s].y, bb[cur][nb].y, acc[nb], 0, 0, 0);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s].z, bb[cur][nb].z, acc[nb], 0, 0, 0);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s].w, bb[cur][nb].w, acc[nb], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4 * NB; ++q) {   // issue order: one MFMA, one LDS read of the next step
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
#pragma unroll
      for (int u = 0; u < NB; ++u) store_w(buf ^ 1, u, w_nxt[u]);
#pragma unroll
      for (int s = 0; s < 4; ++s) a_cur[s] = a_nxt[s];
    }
    __syncthreads();
  }
  }   // !SPLIT

  if (store == 2 && (c_out & 3) == 0) {
    // product rows: the wave's 32 result rows are CONSECUTIVE rows of Y (row = pair index), so each
    // 32 x 32 block goes through a wave-private LDS tile and leaves as 16-byte stores - 4 store
    // instructions per block instead of 16 (the epilogue is store-issue bound, and every
    // workgroup of the launch reaches it at about the same time).  The slab buffers are free: the
    // loop above ended with a barrier.
    float* stage = sBuf + wave * kStageFloats;
    const int c4 = lane & 7, r8 = lane >> 3;
    const int64_t row_base = (int64_t)p0 + wave * 32;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        stage[((r & 3) + 8 * (r >> 2) + 4 * h) * kWPad + i] = acc[nb][r];
      __builtin_amdgcn_wave_barrier();
      const int n = n0 + nb * 32 + 4 * c4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = r8 + 8 * j;
        const float4 v = *reinterpret_cast<const float4*>(&stage[row * kWPad + 4 * c4]);
        if (row_base + row < pend && n < c_out)
          *reinterpret_cast<float4*>(Y + (row_base + row) * c_out + n) = v;
      }
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  int orow[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) orow[r] = __shfl(row_out, (r & 3) + 8 * (r >> 2) + 4 * h);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (orow[r] >= 0) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int n = n0 + nb * 32 + i;
        if (n < c_out) {
          if (store) Y[(int64_t)orow[r] * c_out + n] = acc[nb][r];  // first writer of this row
          else unsafeAtomicAdd(Y + (int64_t)orow[r] * c_out + n, acc[nb][r]);
        }
      }
    }
  }
}

// Weight gradient, c_in % 4 == 0 and c_out % 4 == 0.  A workgroup owns one chunk of `tile_pairs`
// pairs of ONE offset k and a (64*WN) x (64*WC) block of dW[:, k, :].  Per step 32 pairs are
// staged: their dY rows and X rows (only the block's column ranges) go global -> registers ->
// LDS (double buffered, next step's global loads in flight during the MFMAs).  Waves form a
// WN x WC x WK grid; each owns a 64x64 sub-block (2x2 MFMA accumulators) and, when WK > 1, every
// WK-th pair couple of the step (split-K over waves, merged by the final atomics).
constexpr int kStep = 32;
constexpr int kMaxWgradTile = 2048;

// IDENT: no rulebook - pair p is (row p, row p) of a single offset; this is C = A^T B over `K`
// (= total rows) for dense [M, c_out] / [M, c_in] operands (the MLP heads' weight gradients).
template <int WN, int WC, int WK, bool IDENT>
__global__ __launch_bounds__(256) void spconv_wgrad_lds_kernel(
    const float* __restrict__ X, int c_in, const float* __restrict__ dY, int c_out, int K,
    const int32_t* __restrict__ pair_in, const int32_t* __restrict__ pair_out,
    const int32_t* __restrict__ kstart, const int32_t* __restrict__ tile_start, int tile_pairs,
    int n_ntile, int n_ctile, float* __restrict__ dW, float* __restrict__ part) {
  static_assert(WN * WC * WK == 4, "4 waves");
  constexpr int TN = 64 * WN, TC = 64 * WC;
  constexpr int UA = TN / 32, UB = TC / 32;  // float4 staged per thread per step
  __shared__ __attribute__((aligned(16))) float sA[2][kStep * TN];
  __shared__ __attribute__((aligned(16))) float sB[2][kStep * TC];
  __shared__ int s_in[kMaxWgradTile];
  __shared__ int s_out[kMaxWgradTile];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int per_tile = n_ntile * n_ctile;
  const int tile = blockIdx.x / per_tile, sub = blockIdx.x % per_tile;
  const int n0 = (sub / n_ctile) * TN, c0 = (sub % n_ctile) * TC;
  int k, p0, cnt;
  if (IDENT) {  // here `K` carries the row count M and there is one "offset"
    k = 0;
    p0 = tile * tile_pairs;
    cnt = min(K - p0, tile_pairs);
    for (int t = tid; t < tile_pairs; t += 256) s_in[t] = s_out[t] = t < cnt ? p0 + t : -1;
  } else {
    if (tile >= tile_start[K]) return;  // grid sized from an upper bound of the pair counts
    k = find_offset(tile_start, K, tile);
    p0 = kstart[k] + (tile - tile_start[k]) * tile_pairs;
    cnt = min(kstart[k + 1] - p0, tile_pairs);
    for (int t = tid; t < tile_pairs; t += 256) {
      s_in[t] = t < cnt ? pair_in[p0 + t] : -1;
      s_out[t] = t < cnt ? pair_out[p0 + t] : -1;
    }
  }
  const int Kw = IDENT ? 1 : K;  // offsets in the weight tensor
  __syncthreads();

  const int wk = wave % WK, wc = (wave / WK) % WC, wn = wave / (WK * WC);
  const int i = lane & 31, h = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float4 ra[UA], rb[UB];
  // (all row indices first, then all row loads: read one by one in front of its load, every index
  // cost an LDS round trip of its own - eight dependent ones per step, round 4)
  auto load_step = [&](int s) __attribute__((always_inline)) {
    int oo[UA], ii[UB];
#pragma unroll
    for (int u = 0; u < UA; ++u) oo[u] = s_out[s * kStep + (tid + 256 * u) / (TN / 4)];
#pragma unroll
    for (int u = 0; u < UB; ++u) ii[u] = s_in[s * kStep + (tid + 256 * u) / (TC / 4)];
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      const int col = ((tid + 256 * u) % (TN / 4)) * 4;
      ra[u] = ld4(dY + (int64_t)max(oo[u], 0) * c_out + n0 + col, oo[u] >= 0 && n0 + col < c_out);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int col = ((tid + 256 * u) % (TC / 4)) * 4;
      rb[u] = ld4(X + (int64_t)max(ii[u], 0) * c_in + c0 + col, ii[u] >= 0 && c0 + col < c_in);
    }
  };
  auto store_step = [&](int buf) {
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      const int q = tid + 256 * u;
      *reinterpret_cast<float4*>(&sA[buf][(q / (TN / 4)) * TN + (q % (TN / 4)) * 4]) = ra[u];
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int q = tid + 256 * u;
      *reinterpret_cast<float4*>(&sB[buf][(q / (TC / 4)) * TC + (q % (TC / 4)) * 4]) = rb[u];
    }
  };

  const int nsteps = (cnt + kStep - 1) / kStep;
  load_step(0);
  store_step(0);
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    const bool more = (s + 1) < nsteps;
    if (more) load_step(s + 1);
    __builtin_amdgcn_sched_barrier(0);   // (the row loads stay in front of the MFMAs: see dense_conv.hip)
    const float* A = &sA[buf][wn * 64 + i];
    const float* B = &sB[buf][wc * 64 + i];
    // operands of pair couple kk + WK are read from LDS while the four MFMAs of couple kk run
    // (ping-pong sets; before, every group of four MFMAs waited for its own four LDS reads)
    constexpr int NKK = (kStep / 2 + WK - 1) / WK;
    float fa[2][2], fb[2][2];
    {
      const int pr = 2 * wk + h;
      fa[0][0] = A[pr * TN], fa[0][1] = A[pr * TN + 32];
      fb[0][0] = B[pr * TC], fb[0][1] = B[pr * TC + 32];
    }
#pragma unroll
    for (int j = 0; j < NKK; ++j) {
      const int cur = j & 1, nxt = cur ^ 1;
      if (j + 1 < NKK) {
        const int pr = 2 * (wk + (j + 1) * WK) + h;
        fa[nxt][0] = A[pr * TN], fa[nxt][1] = A[pr * TN + 32];
        fb[nxt][0] = B[pr * TC], fb[nxt][1] = B[pr * TC + 32];
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][0], fb[cur][0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][0], fb[cur][1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][1], fb[cur][0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][1], fb[cur][1], acc[1][1], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {   // issue order: one MFMA, one of the next couple's reads
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) store_step(buf ^ 1);
    __syncthreads();
  }

  // part != nullptr: the DETERMINISTIC two-stage form - every (chunk, split-K wave) writes its
  // block of partial sums with plain stores into slab (tile * WK + wk) of `part` ([c_out, c_in]
  // each); wgrad_reduce_kernel (sparse_conv_pr.hip) adds the slabs of an offset in a fixed order.
  // Nothing to clear, no atomics, bitwise reproducible.
  float* slab = part ? part + ((int64_t)tile * WK + wk) * c_out * c_in : nullptr;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + wn * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (n < c_out) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int c = c0 + wc * 64 + b * 32 + i;
          if (c < c_in) {
            if (slab) slab[(int64_t)n * c_in + c] = acc[a][b][r];
            else unsafeAtomicAdd(dW + ((int64_t)n * Kw + k) * c_in + c, acc[a][b][r]);
          }
        }
      }
    }
}

// The weight gradient on the bf16 matrix cores (mfma_split.h).  The reduction runs over PAIRS, and a
// lane of v_mfma_f32_32x32x16_bf16 owns eight consecutive reduction steps of one channel: both
// operands - rows of dY and X, channel-contiguous in memory - are transposed on their way into LDS.
// A staging item is a couple of consecutive pairs x four channels: two 16-byte loads, and for each
// channel and piece ONE dword (the couple's two bf16) written to [channel][piece][couple] - lanes run
// along the couples, so the transposing writes are conflict-free.  Channel rows are padded to KS/2 * 3
// + 4 dwords (52 for 32-pair steps, 100 for 64): ds_read_b128 of 16 consecutive channels hits 64
// distinct banks.  One LDS buffer; the rows of the next TWO steps travel in registers.
// Waves: WN x WC x WK as in the fp32 kernel; wave wk takes the 16-pair sub-steps wk, wk + WK, ...
template <int WN, int WC, int WK>
__global__ __launch_bounds__(256) void spconv_wgrad_split_kernel(
    const float* __restrict__ X, int c_in, const float* __restrict__ dY, int c_out, int K,
    const int32_t* __restrict__ pair_in, const int32_t* __restrict__ pair_out,
    const int32_t* __restrict__ kstart, const int32_t* __restrict__ tile_start, int tile_pairs,
    int n_ntile, int n_ctile, float* __restrict__ dW, float* __restrict__ part) {
  static_assert(WN * WC * WK == 4, "4 waves");
  constexpr int TN = 64 * WN, TC = 64 * WC;
  constexpr int KS = WK == 4 ? 64 : 32;          // pairs per step
  constexpr int NSUB = KS / 16;                  // 16-pair MFMA sub-steps per step
  constexpr int PW = KS / 2;                     // dwords per piece of a channel row
  constexpr int RW = 3 * PW + 4;                 // dwords per channel row
  constexpr int CPL = KS / 2;                    // pair couples per step
  constexpr int ITEMS_A = CPL * (TN / 4), ITEMS_B = CPL * (TC / 4);
  constexpr int UA = ITEMS_A / 256, UB = ITEMS_B / 256;   // staging items per thread
  static_assert(ITEMS_A % 256 == 0 && ITEMS_B % 256 == 0, "whole items per thread");
  // dynamic LDS: the operand tile, then the tile's pair lists (2 x tile_pairs ints; 57 KB at 512 pairs)
  extern __shared__ __attribute__((aligned(16))) unsigned sT[];
  int* s_in = reinterpret_cast<int*>(sT + (TN + TC) * RW);
  int* s_out = s_in + tile_pairs;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int per_tile = n_ntile * n_ctile;
  const int tile = blockIdx.x / per_tile, sub = blockIdx.x % per_tile;
  const int n0 = (sub / n_ctile) * TN, c0 = (sub % n_ctile) * TC;
  if (tile >= tile_start[K]) return;  // grid sized from an upper bound of the pair counts
  const int k = find_offset(tile_start, K, tile);
  const int p0 = kstart[k] + (tile - tile_start[k]) * tile_pairs;
  const int cnt = min(kstart[k + 1] - p0, tile_pairs);
  const int nsteps = (cnt + KS - 1) / KS;
  // (pairs past the end: row 0 with a zero factor - the loads stay unconditional, see the forward kernel)
  for (int t = tid; t < nsteps * KS; t += 256) {
    s_in[t] = t < cnt ? pair_in[p0 + t] : -1;
    s_out[t] = t < cnt ? pair_out[p0 + t] : -1;
  }
  __syncthreads();

  const int wk = wave % WK, wc = (wave / WK) % WC, wn = wave / (WK * WC);
  const int i = lane & 31, h = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  struct Regs {
    float4 a0[UA], a1[UA], b0[UB], b1[UB];
    float za[UA][2], zb[UB][2];     // 1 / 0: the pair exists
  };
  Regs rg[2];
  // item (u): couple q = item % CPL, channel quad cq = item / CPL
  auto load_step = [&](Regs& d, int s) __attribute__((always_inline)) {
    const int sb = min(s, nsteps - 1) * KS;   // (steps past the end: clamped, loaded and never used)
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      const int item = tid + 256 * u;
      const int q = item % CPL, col = min(n0 + 4 * (item / CPL), c_out - 4);
      const int r0 = s_out[sb + 2 * q], r1 = s_out[sb + 2 * q + 1];
      d.a0[u] = *reinterpret_cast<const float4*>(dY + (int64_t)max(r0, 0) * c_out + col);
      d.a1[u] = *reinterpret_cast<const float4*>(dY + (int64_t)max(r1, 0) * c_out + col);
      d.za[u][0] = r0 >= 0 ? 1.f : 0.f, d.za[u][1] = r1 >= 0 ? 1.f : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int item = tid + 256 * u;
      const int q = item % CPL, col = min(c0 + 4 * (item / CPL), c_in - 4);
      const int r0 = s_in[sb + 2 * q], r1 = s_in[sb + 2 * q + 1];
      d.b0[u] = *reinterpret_cast<const float4*>(X + (int64_t)max(r0, 0) * c_in + col);
      d.b1[u] = *reinterpret_cast<const float4*>(X + (int64_t)max(r1, 0) * c_in + col);
      d.zb[u][0] = r0 >= 0 ? 1.f : 0.f, d.zb[u][1] = r1 >= 0 ? 1.f : 0.f;
    }
  };
  auto put = [&](unsigned* row, const float4& v0, const float4& v1, float z0, float z1, int q)
      __attribute__((always_inline)) {
    // (a SELECT, not a product with 0 / 1: the padding pairs read row 0, and 0 * inf = NaN would carry
    // a non-finite entry of row 0 into every tile's tail - round 5 range tests)
    const bool k0 = z0 != 0.f, k1 = z1 != 0.f;
    const float x0[4] = {k0 ? v0.x : 0.f, k0 ? v0.y : 0.f, k0 ? v0.z : 0.f, k0 ? v0.w : 0.f};
    const float x1[4] = {k1 ? v1.x : 0.f, k1 ? v1.y : 0.f, k1 ? v1.z : 0.f, k1 ? v1.w : 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a1 = pv2::bf16_rest(x0[j]), b1 = pv2::bf16_rest(x1[j]);
      unsigned* d = row + j * RW + q;
      d[0] = pv2::pack_hi(x0[j], x1[j]);
      d[PW] = pv2::pack_hi(a1, b1);
      d[2 * PW] = pv2::pack_hi(pv2::bf16_rest(a1), pv2::bf16_rest(b1));
    }
  };
  auto store_step = [&](const Regs& d) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      const int item = tid + 256 * u;
      put(sT + (4 * (item / CPL)) * RW, d.a0[u], d.a1[u], d.za[u][0], d.za[u][1], item % CPL);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int item = tid + 256 * u;
      put(sT + (TN + 4 * (item / CPL)) * RW, d.b0[u], d.b1[u], d.zb[u][0], d.zb[u][1], item % CPL);
    }
  };
  auto multiply = [&]() __attribute__((always_inline)) {
    const unsigned* arow = sT + (wn * 64 + i) * RW + 4 * h;
    const unsigned* brow = sT + (TN + wc * 64 + i) * RW + 4 * h;
#pragma unroll
    for (int st = wk; st < NSUB; st += WK) {
      pv2::bf16x8 fa[2][3], fb[2][3];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          fa[a][pc] = *reinterpret_cast<const pv2::bf16x8*>(arow + a * 32 * RW + pc * PW + 8 * st);
          fb[a][pc] = *reinterpret_cast<const pv2::bf16x8*>(brow + a * 32 * RW + pc * PW + 8 * st);
        }
#define PV2_TERM(ta, tb)                            \
  _Pragma("unroll") for (int a = 0; a < 2; ++a)     \
  _Pragma("unroll") for (int b = 0; b < 2; ++b)     \
    acc[a][b] = pv2::mfma_bf16(fa[a][ta], fb[b][tb], acc[a][b]);
      PV2_SPLIT_TERMS(PV2_TERM)
#undef PV2_TERM
    }
  };

  load_step(rg[0], 0);
  load_step(rg[1], 1);
  auto iteration = [&](int s, Regs& cur) __attribute__((always_inline)) {
    store_step(cur);            // step s (its loads were issued two iterations ago)
    load_step(cur, s + 2);      // ... and the set is free for step s + 2
    __syncthreads();
    multiply();
    __syncthreads();            // every wave is done reading before the next step overwrites
  };
  int s = 0;
  for (; s + 1 < nsteps; s += 2) {
    iteration(s, rg[0]);
    iteration(s + 1, rg[1]);
  }
  if (s < nsteps) iteration(s, rg[0]);

  // ONE slab per tile: the WK waves that share a 64 x 64 block add their accumulators through LDS in
  // wave order (the operand tile is free: the loop ended with a barrier), then the block leaves as
  // 16-byte pieces of its rows.  (Four slabs per tile made the ordered reduction of the narrow layers
  // a chain of 60 dependent loads per output.)
  if constexpr (WK == 1) {   // nothing to add: straight from the accumulators (128-byte row segments)
    float* slab1 = part + (int64_t)tile * c_out * c_in;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (n < c_out) {
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int c = c0 + wc * 64 + b * 32 + i;
            if (c < c_in) slab1[(int64_t)n * c_in + c] = acc[a][b][r];
          }
        }
      }
    return;
  }
  float* red = reinterpret_cast<float*>(sT) + (wn * WC + wc) * 4096;   // [64][64] per block
#pragma unroll
  for (int w = 0; w < WK; ++w) {
    if (wk == w) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float* d = &red[(a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 64 + b * 32 + i];
            *d = w == 0 ? acc[a][b][r] : *d + acc[a][b][r];
          }
    }
    __syncthreads();
  }
  float* slab = part + (int64_t)tile * c_out * c_in;
  for (int e = tid; e < TN * (TC / 4); e += 256) {
    const int row = e / (TC / 4), c4 = e % (TC / 4);
    const int n = n0 + row, c = c0 + 4 * c4;
    if (n < c_out && c < c_in) {
      const float* src = reinterpret_cast<const float*>(sT) +
                         ((row >> 6) * WC + (c4 >> 4)) * 4096 + (row & 63) * 64 + 4 * (c4 & 15);
      *reinterpret_cast<float4*>(slab + (int64_t)n * c_in + c) = *reinterpret_cast<const float4*>(src);
    }
  }
}

// Dense "tall" GEMM  Y[M, N] = X[M, K] . W[N, K]^T (+ bias): the MLP heads of the render field
// (M = rays x samples ~ 1e5, K and N <= 512).  Same structure as spconv_fwd_lds_kernel without the
// rulebook: 128 rows per workgroup, the weight slab shared through LDS, A fragments streamed from
// global memory one slab ahead, plain (non-atomic) 128-byte row segments out.  K % 8 == 0.
template <int NB>
__global__ __launch_bounds__(256) void tall_gemm_nt_kernel(const float* __restrict__ X, int64_t M,
                                                           int K, const float* __restrict__ W,
                                                           int N, const float* __restrict__ bias,
                                                           int n_groups, float* __restrict__ Y) {
  constexpr int NT = 32 * NB;
  __shared__ __attribute__((aligned(16))) float sW[2][NT * kWPad];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t tile = blockIdx.x / n_groups;
  const int grp = blockIdx.x % n_groups;
  const int i = lane & 31, h = lane >> 5;
  const int64_t row = tile * kFwdTile + wave * 32 + i;
  const bool pv = row < M;
  const int n0 = grp * NT;
  const float* xrow = X + (pv ? row : 0) * K + 4 * h;
  const int r0 = tid >> 3, c4 = tid & 7;
  const float* wbase = W + (int64_t)(n0 + r0) * K + 4 * c4;
  const int64_t wstride = (int64_t)32 * K;
  const int wdst0 = r0 * kWPad + 4 * c4;

  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  float4 a_cur[4], a_nxt[4], w_nxt[NB];
#pragma unroll
  for (int s = 0; s < 4; ++s) a_cur[s] = ld4(xrow + 8 * s, pv && (8 * s + 4 * h) < K);
#pragma unroll
  for (int u = 0; u < NB; ++u)
    *reinterpret_cast<float4*>(&sW[0][wdst0 + u * 32 * kWPad]) =
        ld4(wbase + u * wstride, (n0 + r0 + 32 * u < N) && 4 * c4 < K);
  __syncthreads();

  const int nslab = (K + kKC - 1) / kKC;
  for (int t = 0; t < nslab; ++t) {
    const int buf = t & 1;
    const bool more = (t + 1) < nslab;
    if (more) {
      const int kk = (t + 1) * kKC;
#pragma unroll
      for (int s = 0; s < 4; ++s) a_nxt[s] = ld4(xrow + kk + 8 * s, pv && (kk + 8 * s + 4 * h) < K);
#pragma unroll
      for (int u = 0; u < NB; ++u)
        w_nxt[u] = ld4(wbase + u * wstride + kk, (n0 + r0 + 32 * u < N) && (kk + 4 * c4) < K);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const float4 b = *reinterpret_cast<const float4*>(&sW[buf][(nb * 32 + i) * kWPad + 8 * s + 4 * h]);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s].x, b.x, acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s].y, b.y, acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s].z, b.z, acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s].w, b.w, acc[nb], 0, 0, 0);
      }
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < NB; ++u)
        *reinterpret_cast<float4*>(&sW[buf ^ 1][wdst0 + u * 32 * kWPad]) = w_nxt[u];
#pragma unroll
      for (int s = 0; s < 4; ++s) a_cur[s] = a_nxt[s];
    }
    __syncthreads();
  }

  const int64_t rbase = tile * kFwdTile + wave * 32 + 4 * h;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = n0 + nb * 32 + i;
    if (n < N) {
      const float bv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = rbase + (r & 3) + 8 * (r >> 2);
        if (m < M) Y[m * N + n] = acc[nb][r] + bv;
      }
    }
  }
}

// C[I, J] += A[M, I]^T . B[M, J] for narrow operands (I, J <= 32): the weight gradients of the
// 16..32-wide MLP of the outdoor SDF head, M = rays x samples ~ 1e6.  A pure streaming reduction
// over M (algorithmic traffic M*(I+J)*4 bytes, ~10 flop/byte): no LDS staging, each wave walks a
// contiguous run of rows and lane (i, h) feeds element i of row m+h straight from global memory
// into the 32x32x2 MFMA (k = the row pair).  2*kSkinnyUnroll loads per lane are in flight; two
// accumulators break the MFMA dependency chain.  The four waves of a workgroup merge through
// LDS, then one atomic per valid element.
constexpr int kSkinnyUnroll = 8;

__global__ __launch_bounds__(256) void skinny_gemm_tn_kernel(const float* __restrict__ A, int I,
                                                             const float* __restrict__ B, int J,
                                                             int64_t M, int64_t rows_per_wave,
                                                             float* __restrict__ C) {
  __shared__ float red[4][32 * 32];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 31, h = lane >> 5;
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * rows_per_wave;
  const int64_t r1 = min(M, r0 + rows_per_wave);
  const bool va = i < I, vb = i < J;
  const float* pa = A + i;
  const float* pb = B + i;
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
  for (int64_t m = r0 + h; m < r1; m += 2 * kSkinnyUnroll) {
    float a[kSkinnyUnroll], b[kSkinnyUnroll];
#pragma unroll
    for (int u = 0; u < kSkinnyUnroll; ++u) {
      const int64_t mm = m + 2 * u;
      const bool in = mm < r1;
      a[u] = (va && in) ? pa[mm * I] : 0.f;
      b[u] = (vb && in) ? pb[mm * J] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kSkinnyUnroll; u += 2) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u + 1], b[u + 1], acc1, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = (r & 3) + 8 * (r >> 2) + 4 * h;  // row of C comes from the A operand
    red[wave][n * 32 + i] = acc0[r] + acc1[r];
  }
  __syncthreads();
  for (int e = tid; e < 32 * 32; e += 256) {
    const int n = e >> 5, c = e & 31;
    if (n < I && c < J)
      unsafeAtomicAdd(C + n * J + c, red[0][e] + red[1][e] + red[2][e] + red[3][e]);
  }
}

// ------------------------------------------------------------------------------------------
// Output-stationary sparse conv: no atomics, no zero-fill, bitwise run-to-run reproducible.
//
//   Y[o, n] = bias[n] + sum_k sum_c X[nbr[k][o], c] * W[n, kmap(k), c]        (nbr < 0: no term)
//
// `nbr` is the [K, n_out] gather table of the rulebook (rulebook.hip): for a submanifold conv the
// neighbour table itself serves the forward pass and - read with kmap(k) = K-1-k and the
// transposed weights - the grad-input pass (the pair (j -> i, k) is the pair (i -> j, K-1-k)); a
// strided conv uses its child table forwards and the inverted (parent) table backwards.
//
// A workgroup owns 32 output rows x 32*NB output channels.  Rows are taken in the order `perm`
// (rows sorted by their bit mask of present offsets, so that a tile's rows share offsets and few
// gathered rows are empty).  The four waves split the tile's PRESENT offsets round-robin; each
// accumulates its share in registers (A = gathered rows, global -> registers, 32-byte runs; B =
// weight rows of its offset from L2), then the partial tiles meet in LDS and wave 0 adds them in a
// FIXED order and writes every output element exactly once.
template <int NB, bool VEC>
__global__ __launch_bounds__(256) void spconv_os_kernel(
    const float* __restrict__ X, int c_in, const float* __restrict__ W, int K, int c_out,
    const int32_t* __restrict__ nbr, int64_t nbr_stride, const int32_t* __restrict__ perm,
    int kflip, const float* __restrict__ bias, int64_t n_out, float* __restrict__ Y) {
  __shared__ float red[3][NB * 16 * 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 31, h = lane >> 5;
  const int64_t r = (int64_t)blockIdx.x * 32 + i;
  const bool rv = r < n_out;
  const int orow = rv ? (perm ? perm[r] : (int)r) : 0;
  const int n0 = blockIdx.y * NB * 32;

  const float* wrow[NB];
  bool wok[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = n0 + nb * 32 + i;
    wok[nb] = n < c_out;
    wrow[nb] = W + (int64_t)(wok[nb] ? n : 0) * K * c_in + 4 * h;
  }
  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[nb][q] = 0.f;

  // one offset of the tile: gathered rows x the offset's weight rows into the wave's accumulators
  auto one_offset = [&](int k, int idx) __attribute__((always_inline)) {
    const bool pv = idx >= 0;
    const float* xrow = X + (int64_t)(pv ? idx : 0) * c_in + 4 * h;
    const int kw = kflip ? K - 1 - k : k;
#pragma unroll 2
    for (int kk = 0; kk < c_in; kk += 8) {
      const float4 a = load4<VEC>(xrow, kk, c_in - 4 * h, pv);
      float4 b[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        b[nb] = load4<VEC>(wrow[nb] + (int64_t)kw * c_in, kk, c_in - 4 * h, wok[nb]);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[nb].x, acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[nb].y, acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[nb].z, acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[nb].w, acc[nb], 0, 0, 0);
      }
    }
  };
  constexpr int kMaxStaged = 128;
  if (K <= kMaxStaged) {
    // Round 6: the tile's slice of the gather table goes to LDS in ONE round of loads (all 256 threads,
    // K / 8 independent loads each), the offsets present in the tile are compacted into an ascending
    // list, and the waves walk their shares of it (entry e belongs to wave e & 3, as before: the same
    // sums in the same order).  The loop it replaces asked for one table row per iteration and waited for
    // it: 125 dependent trips to L2 for the 5 x 5 x 5 stem (74 us for 75 MFLOP).
    __shared__ int s_tbl[kMaxStaged * 32];
    __shared__ int s_list[kMaxStaged];
    __shared__ int s_cnt[2];
    {
      const int ir = tid & 31;
      const int64_t rr = (int64_t)blockIdx.x * 32 + ir;
      const bool ok = rr < n_out;
      const int orow_r = ok ? (perm ? perm[rr] : (int)rr) : 0;
      for (int k = tid >> 5; k < K; k += 8)
        s_tbl[k * 32 + ir] = ok ? nbr[(int64_t)k * nbr_stride + orow_r] : -1;
    }
    __syncthreads();
    if (tid < kMaxStaged) {
      bool any = false;
      if (tid < K) {
#pragma unroll 8
        for (int j = 0; j < 32; ++j) any |= s_tbl[tid * 32 + j] >= 0;
      }
      const unsigned long long m = __ballot(any);
      if (lane == 0) s_cnt[wave] = __popcll(m);
      __syncthreads();
      const int base = wave == 0 ? 0 : s_cnt[0];
      if (any) s_list[base + __popcll(m & ((1ull << lane) - 1ull))] = tid;
    } else {
      __syncthreads();
    }
    __syncthreads();
    const int n_present = s_cnt[0] + s_cnt[1];
    // (four offsets per trip: their gathers are requested together, the loop is a chain of loads otherwise)
    int e = wave;
    for (; e + 12 < n_present; e += 16) {
      const int k0 = s_list[e], k1 = s_list[e + 4], k2 = s_list[e + 8], k3 = s_list[e + 12];
      const int i0 = s_tbl[k0 * 32 + i], i1 = s_tbl[k1 * 32 + i], i2 = s_tbl[k2 * 32 + i], i3 = s_tbl[k3 * 32 + i];
      if (c_in == 8 && VEC) {
        const float4 a0 = ld4(X + (int64_t)(i0 >= 0 ? i0 : 0) * 8 + 4 * h, i0 >= 0);
        const float4 a1 = ld4(X + (int64_t)(i1 >= 0 ? i1 : 0) * 8 + 4 * h, i1 >= 0);
        const float4 a2 = ld4(X + (int64_t)(i2 >= 0 ? i2 : 0) * 8 + 4 * h, i2 >= 0);
        const float4 a3 = ld4(X + (int64_t)(i3 >= 0 ? i3 : 0) * 8 + 4 * h, i3 >= 0);
        const float4 av[4] = {a0, a1, a2, a3};
        const int kv[4] = {k0, k1, k2, k3};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kw = kflip ? K - 1 - kv[u] : kv[u];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const float4 b = ld4(wrow[nb] + (int64_t)kw * 8, wok[nb]);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, b.x, acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, b.y, acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].z, b.z, acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].w, b.w, acc[nb], 0, 0, 0);
          }
        }
      } else {
        one_offset(k0, i0);
        one_offset(k1, i1);
        one_offset(k2, i2);
        one_offset(k3, i3);
      }
    }
    for (; e < n_present; e += 4) {
      const int k = s_list[e];
      one_offset(k, s_tbl[k * 32 + i]);
    }
  } else {
  int present = 0;
  int idx_next = rv ? nbr[orow] : -1;
  for (int k = 0; k < K; ++k) {
    const int idx = idx_next;
    if (k + 1 < K) idx_next = rv ? nbr[(int64_t)(k + 1) * nbr_stride + orow] : -1;
    if (__ballot(idx >= 0) == 0ull) continue;     // no row of the tile has this offset
    if ((present++ & 3) != wave) continue;        // another wave's share
    one_offset(k, idx);
  }
  }

  if (wave > 0) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int q = 0; q < 16; ++q) red[wave - 1][(nb * 16 + q) * 64 + lane] = acc[nb][q];
  }
  __syncthreads();
  if (wave != 0) return;
  int out_row[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) out_row[q] = __shfl(rv ? orow : -1, (q & 3) + 8 * (q >> 2) + 4 * h);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = n0 + nb * 32 + i;
    const float bv = (bias && n < c_out) ? bias[n] : 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int e = (nb * 16 + q) * 64 + lane;
      const float v = ((acc[nb][q] + red[0][e]) + red[1][e]) + red[2][e];
      if (out_row[q] >= 0 && n < c_out) Y[(int64_t)out_row[q] * c_out + n] = v + bv;
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

// PV2_FP32_MFMA=1: every product on v_mfma_f32_32x32x2_f32 (the round-3 kernels) instead of the
// bf16-piece form
bool use_split() {
  static const bool v = [] {
    const char* e = getenv("PV2_FP32_MFMA");
    return !(e != nullptr && e[0] == '1');
  }();
  return v;
}

template <int NB, bool TRANS, bool SPLIT>
int launch_fwd_lds_t(const float* X, int c_in, const float* W, int K, int c_out, const int32_t* pi,
                     const int32_t* po, const int32_t* ks, const int32_t* ts, int64_t n_tiles,
                     float* Y, hipStream_t s, int64_t center_lo, int64_t center_hi, bool products) {
  const int n_groups = (c_out + NB * 32 - 1) / (NB * 32);
  if (n_tiles * n_groups > 0x7fffffffLL) {
    pv2::set_error("pv2_spconv_forward: grid too large");
    return PV2_E_BADARG;
  }
  if (products) {  // one result row per pair, plain stores (Y = the product-row buffer)
    hipLaunchKernelGGL((spconv_fwd_lds_kernel<NB, TRANS, SPLIT>), dim3((unsigned)(n_tiles * n_groups)), dim3(256),
                       0, s, X, c_in, W, K, c_out, pi, po, ks, ts, n_groups, Y, 0, 0x7fffffff, 0, 2);
  } else if (center_hi > center_lo) {
    // pass A: the centre offset touches every output row exactly once -> plain stores initialise
    // the output (no zero-fill, no atomics); pass B: every other offset accumulates on top.
    const int64_t nc = center_hi - center_lo;
    hipLaunchKernelGGL((spconv_fwd_lds_kernel<NB, TRANS, SPLIT>), dim3((unsigned)(nc * n_groups)), dim3(256), 0,
                       s, X, c_in, W, K, c_out, pi, po, ks, ts, n_groups, Y,
                       (int)center_lo, 0x7fffffff, 0, 1);
    const int64_t rest = n_tiles - nc;
    if (rest > 0)
      hipLaunchKernelGGL((spconv_fwd_lds_kernel<NB, TRANS, SPLIT>), dim3((unsigned)(rest * n_groups)), dim3(256),
                         0, s, X, c_in, W, K, c_out, pi, po, ks, ts, n_groups, Y, 0,
                         (int)center_lo, (int)nc, 0);
  } else {
    hipLaunchKernelGGL((spconv_fwd_lds_kernel<NB, TRANS, SPLIT>), dim3((unsigned)(n_tiles * n_groups)), dim3(256),
                       0, s, X, c_in, W, K, c_out, pi, po, ks, ts, n_groups, Y, 0,
                       0x7fffffff, 0, 0);
  }
  return pv2::check_launch("spconv_fwd_lds");
}

template <int NB, bool TRANS>
int launch_fwd_lds(const float* X, int c_in, const float* W, int K, int c_out, const int32_t* pi,
                   const int32_t* po, const int32_t* ks, const int32_t* ts, int64_t n_tiles,
                   float* Y, hipStream_t s, int64_t center_lo, int64_t center_hi,
                   bool products = false) {
  // (TRANS stages pairs of reduction rows: c_out % 4 == 0 is required of it anyway)
  if (use_split())
    return launch_fwd_lds_t<NB, TRANS, true>(X, c_in, W, K, c_out, pi, po, ks, ts, n_tiles, Y, s, center_lo,
                                             center_hi, products);
  return launch_fwd_lds_t<NB, TRANS, false>(X, c_in, W, K, c_out, pi, po, ks, ts, n_tiles, Y, s, center_lo,
                                            center_hi, products);
}

}  // namespace

// Debug switch: PV2_SPCONV_GENERIC=1 routes everything to the generic (v1) kernels.
static bool force_generic() {
  static const bool v = [] {
    const char* e = getenv("PV2_SPCONV_GENERIC");
    return e != nullptr && e[0] == '1';
  }();
  return v;
}

extern "C" {

int pv2_spconv_forward_tile(int c_in, int c_out) {
  (void)c_out;
  return ((c_in % kKC) == 0 && !force_generic()) ? kFwdTile : PV2_PAIR_TILE;
}

}  // extern "C"


// split-K factor over the waves of a weight-gradient workgroup (see spconv_wgrad_lds_kernel)
static int wgrad_split_k(int c_in, int c_out) {
  if (use_split()) return 1;   // (the bf16-piece kernel adds its waves' shares itself)
  const bool big_n = c_out > 64, big_c = c_in > 64;
  return (big_n && big_c) ? 1 : (big_n || big_c) ? 2 : 4;
}

namespace {

// dW[n, k, c] = sum over the partial slabs of offset k, in slab order (fixed): the second stage of
// the deterministic weight gradient.  One thread per 16 bytes of dW; writes EVERY element (zeros for
// offsets without pairs), so dW needs no clearing.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(
    const float4* __restrict__ part, const int32_t* __restrict__ tile_start, int K, int wk,
    int c_out, int c_in4, float4* __restrict__ dW) {
  // a workgroup = 64 outputs (16 bytes each) x 4 slab lanes: lane j adds slabs j, j + 4, ... in
  // order, the four sub-sums are added in lane order - a fixed order, four times shorter chains
  __shared__ float4 s_sub[256];
  const int64_t total = (int64_t)c_out * K * c_in4;
  const int lane4 = threadIdx.x >> 6, o = threadIdx.x & 63;
  for (int64_t base = (int64_t)blockIdx.x * 64; base < total; base += (int64_t)gridDim.x * 64) {
    const int64_t e = base + o;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < total) {
      const int c4 = (int)(e % c_in4);
      const int k = (int)((e / c_in4) % K);
      const int n = (int)(e / ((int64_t)c_in4 * K));
      const int64_t t0 = (int64_t)tile_start[k] * wk, t1 = (int64_t)tile_start[k + 1] * wk;
      const int64_t slab4 = (int64_t)c_out * c_in4;
      const float4* src = part + (int64_t)n * c_in4 + c4;
      int64_t t = t0 + lane4;
      for (; t + 4 < t1; t += 8) {  // two loads in flight per lane, added in slab order
        const float4 v0 = src[t * slab4], v1 = src[(t + 4) * slab4];
        acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
        acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
      }
      for (; t < t1; t += 4) {
        const float4 v = src[t * slab4];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    s_sub[threadIdx.x] = acc;
    __syncthreads();
    if (lane4 == 0 && e < total) {
      float4 r = s_sub[o];
#pragma unroll
      for (int j = 1; j < 4; ++j) {
        const float4 v = s_sub[j * 64 + o];
        r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w;
      }
      dW[e] = r;
    }
    __syncthreads();
  }
}

}  // namespace

namespace pv2 {

// Weight gradient.  part == nullptr: workgroups add their blocks into the zero-initialised dweight
// with atomics.  part != nullptr (c_in % 4 == 0, c_out % 4 == 0): two-stage deterministic form.
int spconv_wgrad(const float* in_feat, int64_t n_in, int c_in, const float* dout, int64_t n_out,
                 int c_out, int K, const int32_t* pair_in, const int32_t* pair_out,
                 const int32_t* kstart, const int32_t* tile_start, int tile_pairs, int64_t n_tiles,
                 float* dweight, float* part, hipStream_t s) {
  PV2_REQUIRE(c_in >= 1 && c_out >= 1 && K >= 1, "pv2_spconv_backward_weight: bad sizes");
  PV2_REQUIRE(tile_pairs == PV2_WGRAD_TILE || tile_pairs == kMaxWgradTile,
              "pv2_spconv_backward_weight: tile_pairs must be 512 or 2048");
  (void)n_in;
  (void)n_out;
  const bool lds = (c_in % 4) == 0 && (c_out % 4) == 0 && !force_generic();
  PV2_REQUIRE(part == nullptr || lds,
              "pv2_spconv_backward_weight_det: channel counts must be multiples of 4");
  if (n_tiles == 0 && part == nullptr) return PV2_OK;
  if (lds) {
    const bool big_n = c_out > 64, big_c = c_in > 64;
    const int n_ntile = (c_out + (big_n ? 127 : 63)) / (big_n ? 128 : 64);
    const int n_ctile = (c_in + (big_c ? 127 : 63)) / (big_c ? 128 : 64);
    const int64_t blocks = n_tiles * n_ntile * n_ctile;
    if (blocks > 0x7fffffffLL) {
      pv2::set_error("pv2_spconv_backward_weight: grid too large");
      return PV2_E_BADARG;
    }
#define PV2_LAUNCH_WGRAD_LDS(WN, WC, WK)                                                          \
  do {                                                                                            \
    if (part != nullptr && use_split()) {                                                         \
      constexpr int ks = WK == 4 ? 64 : 32;                                                       \
      const size_t lds = ((size_t)(64 * WN + 64 * WC) * (3 * ks / 2 + 4) + 2 * (size_t)tile_pairs) * 4; \
      if (lds > 65536) {                                                                          \
        static bool allowed = false; /* (per instantiation: the macro expands once per variant) */ \
        if (!allowed) {                                                                           \
          if (int e = pv2::hip_status(hipFuncSetAttribute(                                        \
                  reinterpret_cast<const void*>(&spconv_wgrad_split_kernel<WN, WC, WK>),          \
                  hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)))                        \
            return e;                                                                             \
          allowed = true;                                                                         \
        }                                                                                         \
      }                                                                                           \
      hipLaunchKernelGGL((spconv_wgrad_split_kernel<WN, WC, WK>), dim3((unsigned)blocks),         \
                         dim3(256), lds, s, in_feat, c_in, dout, c_out, K, pair_in, pair_out,     \
                         kstart, tile_start, tile_pairs, n_ntile, n_ctile, dweight, part);        \
    } else                                                                                        \
      hipLaunchKernelGGL((spconv_wgrad_lds_kernel<WN, WC, WK, false>), dim3((unsigned)blocks),    \
                         dim3(256), 0, s, in_feat, c_in, dout, c_out, K, pair_in, pair_out,       \
                         kstart, tile_start, tile_pairs, n_ntile, n_ctile, dweight, part);        \
  } while (0)
    if (blocks > 0) {
      if (big_n && big_c) PV2_LAUNCH_WGRAD_LDS(2, 2, 1);
      else if (big_n) PV2_LAUNCH_WGRAD_LDS(2, 1, 2);
      else if (big_c) PV2_LAUNCH_WGRAD_LDS(1, 2, 2);
      else PV2_LAUNCH_WGRAD_LDS(1, 1, 4);
    }
#undef PV2_LAUNCH_WGRAD_LDS
    if (part != nullptr) {
      const int64_t total4 = (int64_t)c_out * K * (c_in / 4);
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(pv2::grid_for(total4, 64)), dim3(256), 0, s,
                         (const float4*)part, tile_start, K, wgrad_split_k(c_in, c_out), c_out,
                         c_in / 4, (float4*)dweight);
    }
    return pv2::check_launch("spconv_wgrad_lds");
  }
  PV2_REQUIRE(tile_pairs == PV2_WGRAD_TILE, "pv2_spconv_backward_weight: generic path needs 512");
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

}  // namespace pv2

namespace pv2 {

// Stage one of the product-row convolution: prod[p, :] = W[k(p)] . in[pair_in[p], :] for every pair
// p of the canonical list (trans: weight given reduction-major, as pv2_spconv_forward_wt).
int spconv_products(bool trans, const float* in_feat, int c_in, const float* weight, int K,
                    int c_out, const int32_t* pair_in, const int32_t* kstart,
                    const int32_t* tile_start, int64_t n_tiles, float* prod, hipStream_t s) {
  PV2_REQUIRE(c_in >= kKC && (c_in % kKC) == 0 && c_out >= 1 && K >= 1,
              "pv2_spconv_products: c_in must be a multiple of 32");
  PV2_REQUIRE(!trans || (c_out % 4) == 0, "pv2_spconv_products: c_out must be a multiple of 4");
  if (n_tiles == 0) return PV2_OK;
  const int nblk = (c_out + 31) / 32;
#define PV2_PROD_ARGS in_feat, c_in, weight, K, c_out, pair_in, nullptr, kstart, tile_start, n_tiles, prod, s, 0, 0, true
  if (trans) {
    switch (nblk >= 4 ? 4 : nblk) {
      case 1: return launch_fwd_lds<1, true>(PV2_PROD_ARGS);
      case 2: return launch_fwd_lds<2, true>(PV2_PROD_ARGS);
      case 3: return launch_fwd_lds<3, true>(PV2_PROD_ARGS);
      default: return launch_fwd_lds<4, true>(PV2_PROD_ARGS);
    }
  }
  switch (nblk >= 4 ? 4 : nblk) {
    case 1: return launch_fwd_lds<1, false>(PV2_PROD_ARGS);
    case 2: return launch_fwd_lds<2, false>(PV2_PROD_ARGS);
    case 3: return launch_fwd_lds<3, false>(PV2_PROD_ARGS);
    default: return launch_fwd_lds<4, false>(PV2_PROD_ARGS);
  }
#undef PV2_PROD_ARGS
}

}  // namespace pv2

static int spconv_forward_impl(bool trans, const float* in_feat, int64_t n_in, int c_in,
                               const float* weight, int K, int c_out, const int32_t* pair_in,
                               const int32_t* pair_out, const int32_t* kstart,
                               const int32_t* tile_start, int tile_pairs, int64_t n_tiles,
                               int64_t center_tile_lo, int64_t center_tile_hi, float* out_feat,
                               int64_t n_out, pv2_stream_t stream) {
  PV2_REQUIRE(c_in >= 1 && c_out >= 1 && K >= 1, "pv2_spconv_forward: bad channel/offset count");
  PV2_REQUIRE(tile_pairs == pv2_spconv_forward_tile(c_in, c_out),
              "pv2_spconv_forward: tile_pairs must be pv2_spconv_forward_tile(c_in, c_out)");
  (void)n_in;
  (void)n_out;
  if (n_tiles == 0) return PV2_OK;
  hipStream_t s = (hipStream_t)stream;
  const int nblk = (c_out + 31) / 32;
#define PV2_FWD_ARGS in_feat, c_in, weight, K, c_out, pair_in, pair_out, kstart, tile_start, n_tiles, out_feat, s
  if (tile_pairs == kFwdTile) {
    PV2_REQUIRE(center_tile_hi <= n_tiles && center_tile_lo <= center_tile_hi,
                "pv2_spconv_forward: bad centre tile range");
    if (trans) {
      PV2_REQUIRE((c_out % 4) == 0, "pv2_spconv_forward_wt: c_out must be a multiple of 4");
      switch (nblk >= 4 ? 4 : nblk) {
        case 1: return launch_fwd_lds<1, true>(PV2_FWD_ARGS, center_tile_lo, center_tile_hi);
        case 2: return launch_fwd_lds<2, true>(PV2_FWD_ARGS, center_tile_lo, center_tile_hi);
        case 3: return launch_fwd_lds<3, true>(PV2_FWD_ARGS, center_tile_lo, center_tile_hi);
        default: return launch_fwd_lds<4, true>(PV2_FWD_ARGS, center_tile_lo, center_tile_hi);
      }
    }
    switch (nblk >= 4 ? 4 : nblk) {
      case 1: return launch_fwd_lds<1, false>(PV2_FWD_ARGS, center_tile_lo, center_tile_hi);
      case 2: return launch_fwd_lds<2, false>(PV2_FWD_ARGS, center_tile_lo, center_tile_hi);
      case 3: return launch_fwd_lds<3, false>(PV2_FWD_ARGS, center_tile_lo, center_tile_hi);
      default: return launch_fwd_lds<4, false>(PV2_FWD_ARGS, center_tile_lo, center_tile_hi);
    }
  }
  PV2_REQUIRE(!trans, "pv2_spconv_forward_wt: needs the LDS-staged kernel (c_in % 32 == 0)");
  PV2_REQUIRE(center_tile_hi <= center_tile_lo,
              "pv2_spconv_forward: the generic kernel has no store pass (pass an empty centre range)");
  // generic path (any c_in): one wave per 32-pair tile, operands straight from global memory
  int nb = nblk >= 4 ? 4 : nblk;
  if (nb == 4 && n_tiles * ((nblk + 3) / 4) < 2048) nb = 2;
  switch (nb) {
    case 1: return launch_fwd<1>(PV2_FWD_ARGS);
    case 2: return launch_fwd<2>(PV2_FWD_ARGS);
    case 3: return launch_fwd<3>(PV2_FWD_ARGS);
    default: return launch_fwd<4>(PV2_FWD_ARGS);
  }
#undef PV2_FWD_ARGS
}

extern "C" {

int pv2_spconv_forward(const float* in_feat, int64_t n_in, int c_in, const float* weight, int K,
                       int c_out, const int32_t* pair_in, const int32_t* pair_out,
                       const int32_t* kstart, const int32_t* tile_start, int tile_pairs,
                       int64_t n_tiles, int64_t center_tile_lo, int64_t center_tile_hi,
                       float* out_feat, int64_t n_out, pv2_stream_t stream) {
  return spconv_forward_impl(false, in_feat, n_in, c_in, weight, K, c_out, pair_in, pair_out, kstart,
                             tile_start, tile_pairs, n_tiles, center_tile_lo, center_tile_hi,
                             out_feat, n_out, stream);
}

int pv2_spconv_forward_wt(const float* in_feat, int64_t n_in, int c_in, const float* weight, int K,
                          int c_out, const int32_t* pair_in, const int32_t* pair_out,
                          const int32_t* kstart, const int32_t* tile_start, int tile_pairs,
                          int64_t n_tiles, float* out_feat, int64_t n_out, pv2_stream_t stream) {
  return spconv_forward_impl(true, in_feat, n_in, c_in, weight, K, c_out, pair_in, pair_out, kstart,
                             tile_start, tile_pairs, n_tiles, 0, 0, out_feat, n_out, stream);
}

int pv2_spconv_os_forward(const float* in_feat, int64_t n_in, int c_in, const float* weight, int K,
                          int c_out, const int32_t* nbr, int64_t nbr_stride, const int32_t* perm,
                          int kflip, const float* bias, float* out_feat, int64_t n_out,
                          pv2_stream_t stream) {
  PV2_REQUIRE(c_in >= 1 && c_out >= 1 && K >= 1, "pv2_spconv_os_forward: bad channel/offset count");
  PV2_REQUIRE(nbr_stride >= n_out && n_out < 0x7fffffffLL, "pv2_spconv_os_forward: bad row count");
  (void)n_in;
  if (n_out == 0) return PV2_OK;
  const int nblk = (c_out + 31) / 32;
  // wide tiles amortise the gathered rows; narrow ones give small layers enough workgroups
  const int64_t row_tiles = (n_out + 31) / 32;
  int nb = nblk >= 4 ? 4 : nblk;
  if (nb == 4 && row_tiles * ((nblk + 3) / 4) < 512) nb = 2;
  if (nb == 2 && row_tiles * ((nblk + 1) / 2) < 512) nb = 1;
  const int groups = (nblk + nb - 1) / nb;
  PV2_REQUIRE(row_tiles < 0x7fffffffLL && groups < 65536, "pv2_spconv_os_forward: grid too large");
  const bool vec = (c_in % 8) == 0;
  const dim3 grid((unsigned)row_tiles, (unsigned)groups);
  hipStream_t s = (hipStream_t)stream;
#define PV2_LAUNCH_OS(NB)                                                                          \
  do {                                                                                             \
    if (vec)                                                                                       \
      hipLaunchKernelGGL((spconv_os_kernel<NB, true>), grid, dim3(256), 0, s, in_feat, c_in,       \
                         weight, K, c_out, nbr, nbr_stride, perm, kflip, bias, n_out, out_feat);   \
    else                                                                                           \
      hipLaunchKernelGGL((spconv_os_kernel<NB, false>), grid, dim3(256), 0, s, in_feat, c_in,      \
                         weight, K, c_out, nbr, nbr_stride, perm, kflip, bias, n_out, out_feat);   \
  } while (0)
  switch (nb) {
    case 1: PV2_LAUNCH_OS(1); break;
    case 2: PV2_LAUNCH_OS(2); break;
    case 3: PV2_LAUNCH_OS(3); break;
    default: PV2_LAUNCH_OS(4); break;
  }
#undef PV2_LAUNCH_OS
  return pv2::check_launch("spconv_os_forward");
}

int pv2_spconv_backward_weight(const float* in_feat, int64_t n_in, int c_in, const float* dout,
                               int64_t n_out, int c_out, int K, const int32_t* pair_in,
                               const int32_t* pair_out, const int32_t* kstart,
                               const int32_t* tile_start, int tile_pairs, int64_t n_tiles,
                               float* dweight, pv2_stream_t stream) {
  return pv2::spconv_wgrad(in_feat, n_in, c_in, dout, n_out, c_out, K, pair_in, pair_out, kstart,
                           tile_start, tile_pairs, n_tiles, dweight, nullptr, (hipStream_t)stream);
}

int64_t pv2_spconv_wgrad_partial_floats(int c_in, int c_out, int64_t n_tiles) {
  return n_tiles * wgrad_split_k(c_in, c_out) * (int64_t)c_out * c_in;
}

int pv2_spconv_backward_weight_det(const float* in_feat, int64_t n_in, int c_in, const float* dout,
                                   int64_t n_out, int c_out, int K, const int32_t* pair_in,
                                   const int32_t* pair_out, const int32_t* kstart,
                                   const int32_t* tile_start, int tile_pairs, int64_t n_tiles,
                                   float* partial, float* dweight, pv2_stream_t stream) {
  PV2_REQUIRE(partial != nullptr, "pv2_spconv_backward_weight_det: needs the partial-sum buffer");
  return pv2::spconv_wgrad(in_feat, n_in, c_in, dout, n_out, c_out, K, pair_in, pair_out, kstart,
                           tile_start, tile_pairs, n_tiles, dweight, partial, (hipStream_t)stream);
}

int pv2_gemm_nt(const float* x, int64_t m, int k, const float* w, int n, const float* bias,
                float* y, pv2_stream_t stream) {
  PV2_REQUIRE(k >= 8 && (k % 8) == 0, "pv2_gemm_nt: k must be a positive multiple of 8");
  PV2_REQUIRE(n >= 1 && m >= 0, "pv2_gemm_nt: bad sizes");
  if (m == 0) return PV2_OK;
  const int nblk = (n + 31) / 32;
  const int nb = nblk >= 4 ? 4 : nblk;
  const int n_groups = (nblk + nb - 1) / nb;
  const int64_t blocks = ((m + kFwdTile - 1) / kFwdTile) * n_groups;
  if (blocks > 0x7fffffffLL) {
    pv2::set_error("pv2_gemm_nt: grid too large");
    return PV2_E_BADARG;
  }
#define PV2_LAUNCH_TALL(NB)                                                                      \
  hipLaunchKernelGGL((tall_gemm_nt_kernel<NB>), dim3((unsigned)blocks), dim3(256), 0,            \
                     (hipStream_t)stream, x, m, k, w, n, bias, n_groups, y)
  switch (nb) {
    case 1: PV2_LAUNCH_TALL(1); break;
    case 2: PV2_LAUNCH_TALL(2); break;
    case 3: PV2_LAUNCH_TALL(3); break;
    default: PV2_LAUNCH_TALL(4); break;
  }
#undef PV2_LAUNCH_TALL
  return pv2::check_launch("gemm_nt");
}

int pv2_gemm_tn(const float* a, const float* b, int64_t m, int k1, int k2, float* c,
                pv2_stream_t stream) {
  PV2_REQUIRE(k1 >= 4 && k2 >= 4 && (k1 % 4) == 0 && (k2 % 4) == 0,
              "pv2_gemm_tn: k1 and k2 must be positive multiples of 4");
  PV2_REQUIRE(m >= 0 && m < 0x7fffffffLL, "pv2_gemm_tn: bad row count");
  if (m == 0) return PV2_OK;
  if (k1 <= 32 && k2 <= 32) {
    // ~4096 waves in flight, every wave's run a whole number of unrolled row pairs
    const int64_t quantum = 2 * kSkinnyUnroll;
    int64_t rows = (m + 4095) / 4096;
    rows = ((rows < 128 ? 128 : rows) + quantum - 1) / quantum * quantum;
    const int64_t blocks = (m + 4 * rows - 1) / (4 * rows);
    hipLaunchKernelGGL(skinny_gemm_tn_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, a, k1, b, k2, m, rows, c);
    return pv2::check_launch("gemm_tn_skinny");
  }
  const bool big_n = k1 > 64, big_c = k2 > 64;
  const int n_ntile = (k1 + (big_n ? 127 : 63)) / (big_n ? 128 : 64);
  const int n_ctile = (k2 + (big_c ? 127 : 63)) / (big_c ? 128 : 64);
  // every workgroup ends with one atomic add per element of its block of C: very tall operands
  // (the 1x1x1 convolution over a million grid cells) take the longest run the kernel supports
  const int tile_pairs = m >= (int64_t)PV2_WGRAD_TILE * 1024 ? kMaxWgradTile : PV2_WGRAD_TILE;
  const int64_t blocks = ((m + tile_pairs - 1) / tile_pairs) * n_ntile * n_ctile;
  hipStream_t s = (hipStream_t)stream;
#define PV2_LAUNCH_TN(WN, WC, WK)                                                                \
  hipLaunchKernelGGL((spconv_wgrad_lds_kernel<WN, WC, WK, true>), dim3((unsigned)blocks),         \
                     dim3(256), 0, s, b, k2, a, k1, (int)m, nullptr, nullptr, nullptr, nullptr,  \
                     tile_pairs, n_ntile, n_ctile, c, nullptr)
  if (big_n && big_c) PV2_LAUNCH_TN(2, 2, 1);
  else if (big_n) PV2_LAUNCH_TN(2, 1, 2);
  else if (big_c) PV2_LAUNCH_TN(1, 2, 2);
  else PV2_LAUNCH_TN(1, 1, 4);
#undef PV2_LAUNCH_TN
  return pv2::check_launch("gemm_tn");
}

int pv2_spconv_wgrad_tile(int c_in, int c_out, int64_t n_pairs, int K) {
  if ((c_in % 4) != 0 || (c_out % 4) != 0 || force_generic()) return PV2_WGRAD_TILE;
  const int nt = (c_out + (c_out > 64 ? 127 : 63)) / (c_out > 64 ? 128 : 64);
  const int ct = (c_in + (c_in > 64 ? 127 : 63)) / (c_in > 64 ? 128 : 64);
  // long chunks amortise the final atomics; keep >= ~768 workgroups in flight
  const int64_t big = (n_pairs / kMaxWgradTile + K) * nt * ct;
  return big >= 768 ? kMaxWgradTile : PV2_WGRAD_TILE;
}

}  // extern "C"
