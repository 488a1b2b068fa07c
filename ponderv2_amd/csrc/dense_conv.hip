// Dense 3x3x3 convolutions of the projection network (UNet3D-v1m2) on gfx950: fp32 MFMA implicit
// GEMM over channels-last grids, the input tile and its halo staged ONCE in LDS and re-used by all
// 27 taps.
//
// Stands in for the MIOpen / composable-kernel convolutions behind nn.Conv3d(k3, p1) and
// nn.ConvTranspose3d(k3, s2, p1) of the reference's ponder/models/ponder/unet3d.py (SingleConv
// :45-156 order "bcr", Encoder :292-356, Decoder :359-444, Upsampling :447-493) - forward,
// grad-input and grad-weight - with what surrounds them folded into the load / store paths:
// the preceding BatchNorm3d's affine map (x * scale + shift, zero padding applied AFTER it, as the
// module order batchnorm -> conv demands), the following ReLU, the ReLU mask of the backward pass,
// the decoder's "skip + upsampled" sum and the transposed conv's bias.
//
// Layout: grids are (B, Z, Y, X, C) fp32 (a channels_last_3d view of (B, C, Z, Y, X)); a cell's C
// channels are contiguous.  MFMA v_mfma_f32_32x32x2_f32 (exact fp32): M = 32 cells of a tile, N =
// 32 output channels, K = input channels (x 27 taps).
//   A operand (cells x channels): ds_read_b128 from the LDS halo tile, rows padded by 4 floats
//     (16 staged channels + 4 = 20, or 8 + 4 = 12): conflict-free for every 16-lane group;
//   B operand (weights): packed once per optimiser step into MFMA fragment order
//     (dconv_pack_kernel), so a wave's load is one contiguous 1 KB read that hits L1 / L2 for all
//     but the first workgroup; prefetched one tap ahead straight into registers - the main loop
//     has no barrier except around the staging of the next 16-channel chunk.
// Three shapes of the same loop:
//   dconv_kernel<NB, MT>   one output cell per iteration cell: conv k3 s1 p1 (forward, and the
//                          grad-input = the same conv on flipped, transposed weights) and the
//                          strided conv k3 s2 p1 (grad-input of the transposed conv);
//   dconvT_kernel          transposed conv k3 s2 p1 (out = 2 x in): 8 output parity classes per
//                          coarse cell, each with its own accumulator; a tap feeds exactly one class;
//   dconv_wgrad_kernel     weight gradient of both: K = cells, per-workgroup partial slabs then an
//                          ordered sum (dconv_wgrad_reduce_kernel): no atomics, bitwise repeatable.
#include <mutex>
#include <type_traits>
#include <unordered_map>

#include "common.h"
#include "mfma_split.h"

namespace {

using pv2::f32x16;

struct DGeom {
  int B, Zi, Yi, Xi;  // input grid (what is staged)
  int Zo, Yo, Xo;     // output grid
  int Zt, Yt, Xt;     // iteration grid: M runs over its cells
  int TZ, TY, TX;     // cell decode of a workgroup tile (powers of two, TZ*TY*TX = 128*MT); cells
  int eTZ;            // with z >= eTZ do not exist (tiles of grids smaller than a full tile)
  int lTX, lTY;       // log2(TX), log2(TY)
  int nTZ, nTY, nTX;  // tiles per axis (z in steps of eTZ)
  int HZ, HY, HX;     // staged halo box (input cells)
  int in_mul, in_off; // halo origin = tile origin * in_mul + in_off
  int cell_mul;       // LDS row of iteration cell (z, y, x) = ((z*cm)*HY + y*cm)*HXr + xrow(x*cm)
  int xsplit, HXh;    // strided conv: the box's x axis is stored odd / even de-interleaved (HXh each)
  int HXr;            // row count of the x axis in LDS (HX, or 2*HXh)
  float rHX, rHY;     // 1 / HX, 1 / HY: box row -> (hz, hy, hx) without integer division
};

__device__ __forceinline__ float4 ld4g(const float* __restrict__ p) {
  return *reinterpret_cast<const float4*>(p);
}

__device__ __forceinline__ void keep_positive(float4& v, const float4& m) {
  if (!(m.x > 0.f)) v.x = 0.f;
  if (!(m.y > 0.f)) v.y = 0.f;
  if (!(m.z > 0.f)) v.z = 0.f;
  if (!(m.w > 0.f)) v.w = 0.f;
}

__device__ __forceinline__ void affine4(float4& v, const float4& sc, const float4& sh) {
  v.x = v.x * sc.x + sh.x;
  v.y = v.y * sc.y + sh.y;
  v.z = v.z * sc.z + sh.z;
  v.w = v.w * sc.w + sh.w;
}

// row -> (hz, hy, hx) of a box of HY x HX rows per z-slice.  floor((row + 0.5) * (1 / HX)) == row / HX
// for the few thousand rows of a box: the product is off by < 3e-4, (row + 0.5) / HX is >= 0.5 / HX
// away from an integer.  (Runtime divisors: an integer division is ~40 VALU instructions, and the
// staging loops did three per float4.)
__device__ __forceinline__ void decode_row(int row, int HX, int HY, float rHX, float rHY, int& hz,
                                           int& hy, int& hx) {
  const int t2 = (int)(((float)row + 0.5f) * rHX);
  hx = row - t2 * HX;
  hz = (int)(((float)t2 + 0.5f) * rHY);
  hy = t2 - hz * HY;
}

// Stage the CK-channel chunk `ck` of the halo box into LDS: row = halo cell, CK/4 float4 per row,
// rows padded to CK + 4 floats.  Outside the grid: zeros (the padding of the convolution, applied
// after the affine map).  Four loads (eight with a mask) are in flight per thread before the first
// LDS store: the loop is latency-bound otherwise (one L2 / MALL round trip per iteration).
template <int CK>
__device__ __forceinline__ void stage_chunk(float* __restrict__ sX, const float* __restrict__ X,
                                            const DGeom& g, int b, int hz0, int hy0, int hx0,
                                            int c_in, int ck, const float* __restrict__ in_scale,
                                            const float* __restrict__ in_shift,
                                            const float* __restrict__ mask_src, int tid,
                                            int nthreads) {
  constexpr int QPR = CK / 4, LDR = CK + 4;
  const int quad = tid & (QPR - 1);
  const int c0 = ck * CK + quad * 4;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (in_scale != nullptr) {
    sc = ld4g(in_scale + c0);
    sh = ld4g(in_shift + c0);
  }
  const int total = g.HZ * g.HY * g.HX * QPR;
  constexpr int U = 4;
  for (int base = tid; base < total; base += nthreads * U) {
    float4 v[U], m[U];
    int dst[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * nthreads;
      const int row = idx / QPR;
      int hx, hy, hz;
      decode_row(row, g.HX, g.HY, g.rHX, g.rHY, hz, hy, hx);
      const int iz = hz0 + hz, iy = hy0 + hy, ix = hx0 + hx;
      const int xr = g.xsplit ? (hx & 1) * g.HXh + (hx >> 1) : hx;
      dst[u] = idx < total ? ((hz * g.HY + hy) * g.HXr + xr) * LDR + quad * 4 : -1;
      ok[u] = idx < total && iz >= 0 && iz < g.Zi && iy >= 0 && iy < g.Yi && ix >= 0 && ix < g.Xi;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      m[u] = make_float4(1.f, 1.f, 1.f, 1.f);
      if (ok[u]) {
        const int64_t off = ((((int64_t)b * g.Zi + iz) * g.Yi + iy) * g.Xi + ix) * c_in + c0;
        v[u] = ld4g(X + off);
        if (mask_src != nullptr) m[u] = ld4g(mask_src + off);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (dst[u] < 0) continue;
      if (ok[u]) {
        if (in_scale != nullptr) affine4(v[u], sc, sh);
        if (mask_src != nullptr) keep_positive(v[u], m[u]);  // ReLU backward
      }
      *reinterpret_cast<float4*>(&sX[dst[u]]) = v[u];
    }
  }
}

constexpr int kStagePad = 36;  // row stride of the epilogue's 32 x 32 staging tiles

// acc (MFMA layout: column n = lane & 31, rows m = (r & 3) + 8 (r >> 2) + 4 h) -> the wave's LDS
// staging tile, so the block can leave as 16-byte rows
__device__ __forceinline__ void acc_to_stage(float* __restrict__ stage, const f32x16& acc, int i, int h) {
#pragma unroll
  for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * h) * kStagePad + i] = acc[r];
}

// ---------------------------------------------------------------------------------------------
// One output cell per iteration cell.  Workgroup = 4 waves; wave w owns MT M-tiles of 32 cells
// (cells (w*MT + mt)*32 .. +32 of the tile in x-fastest order) and NB 32-wide output channel
// blocks: MT*NB independent accumulators.  Packed weights (dconv_pack_kernel):
//   Wp[t][r8][nb][lane][4]: lane (i, h) holds W[out = nb*32 + i][red = r8*8 + 4h + q][tap t], q = 0..3
//   - the B fragments of the four MFMAs of one 8-channel reduction step.
// CK: channels staged per chunk (16; 8 for the strided conv, whose halo box is 8x the tile).
// A workgroup walks `tiles_per_wg` consecutive tiles x c_in / CK chunks as ONE pipeline of items:
// while the 27 taps of an item run on the matrix pipe, the next item's box is on its way from L2 /
// HBM into registers (PFN float4 per thread, twice that with the ReLU mask); it is written to LDS
// when the taps are done.  Without that, two co-resident workgroups that start together stay in
// phase - both staging, then both computing - and the matrix pipe idles a third of the time
// (profiles/r04_pmc_dense_conv_v3.txt: 61 % busy).
// ---------------------------------------------------------------------------------------------
template <int NB, int MT, int CK, int PFN, bool MASKED>
__global__ __launch_bounds__(256, 2) void dconv_kernel(
    const float* __restrict__ X, DGeom g, int c_in, const float* __restrict__ Wp, int c_out,
    int n_groups, int tiles_per_wg, const float* __restrict__ in_scale,
    const float* __restrict__ in_shift, const float* __restrict__ mask_src,
    const float* __restrict__ bias, const float* __restrict__ addend, int relu,
    const float* __restrict__ out_mask_src, float* __restrict__ Y) {
  constexpr int S = CK / 8, LDR = CK + 4, QPR = CK / 4;
  extern __shared__ __attribute__((aligned(16))) float sX[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
  const int grp = blockIdx.x % n_groups;
  const int n_tiles = g.B * g.nTZ * g.nTY * g.nTX;
  const int tile_first = (blockIdx.x / n_groups) * tiles_per_wg;
  const int tile_count = min(tiles_per_wg, n_tiles - tile_first);
  const int nchunks = c_in / CK;
  const int n_items = tile_count * nchunks;
  // (plain scalars: what the always_inline lambdas below capture stays in registers)
  const int Zi = g.Zi, Yi = g.Yi, Xi = g.Xi, HX = g.HX, HY = g.HY, HXr = g.HXr, HXh = g.HXh;
  const int xsplit = g.xsplit, in_mul = g.in_mul, in_off = g.in_off;
  const int nTX = g.nTX, nTY = g.nTY, nTZ = g.nTZ, eTZ = g.eTZ, TYs = g.TY, TXs = g.TX;
  const int TXm = g.TX - 1, TYm = g.TY - 1, lTX = g.lTX, lTXY = g.lTX + g.lTY;
  const float rHX = g.rHX, rHY = g.rHY;
  const int total = g.HZ * HY * HX * QPR;
  const int quad = tid & (QPR - 1);

  int rowbase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int q = (wave * MT + mt) * 32 + i;
    const int cx = q & TXm, cy = (q >> lTX) & TYm, cz = q >> lTXY;
    // (strided conv: even box column 2 cx -> row cx of the even half)
    rowbase[mt] = ((cz * g.cell_mul) * HY + cy * g.cell_mul) * HXr + (xsplit ? cx : cx * g.cell_mul);
    if (cz >= eTZ) rowbase[mt] = 0;  // no such cell: any staged row, the result is dropped
  }

  f32x16 acc[MT][NB];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nb][r] = 0.f;

  const int nbtot = c_out >> 5;
  const float4* __restrict__ Wp4 = reinterpret_cast<const float4*>(Wp) + (grp * NB) * 64 + lane;
  const int64_t wr8 = (int64_t)nbtot * 64;             // float4 per 8-channel step
  const int64_t wtap = (int64_t)(c_in >> 3) * wr8;     // float4 per tap

  auto tile_origin = [&](int tile, int& b, int& z0, int& y0, int& x0) __attribute__((always_inline)) {
    const int tx = tile % nTX;
    tile /= nTX;
    const int ty = tile % nTY;
    tile /= nTY;
    b = tile / nTZ;
    z0 = (tile % nTZ) * eTZ, y0 = ty * TYs, x0 = tx * TXs;
  };

  // ---- the box of the NEXT item, global -> registers
  float4 pf[PFN], pm[MASKED ? PFN : 1], psc, psh;
  unsigned okbits = 0;
  auto fetch = [&](int tl, int ck) __attribute__((always_inline)) {
    int b, z0, y0, x0;
    tile_origin(tile_first + tl, b, z0, y0, x0);
    const int hz0 = z0 * in_mul + in_off, hy0 = y0 * in_mul + in_off, hx0 = x0 * in_mul + in_off;
    const int c0 = ck * CK + quad * 4;
    if (in_scale != nullptr) {
      psc = ld4g(in_scale + c0);
      psh = ld4g(in_shift + c0);
    }
    okbits = 0;
#pragma unroll
    for (int u = 0; u < PFN; ++u) {
      const int idx = tid + u * 256;
      int hx, hy, hz;
      decode_row(idx / QPR, HX, HY, rHX, rHY, hz, hy, hx);
      const int iz = hz0 + hz, iy = hy0 + hy, ix = hx0 + hx;
      const bool ok = idx < total && iz >= 0 && iz < Zi && iy >= 0 && iy < Yi && ix >= 0 && ix < Xi;
      pf[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (MASKED) pm[u] = make_float4(1.f, 1.f, 1.f, 1.f);
      if (ok) {
        const int64_t off = ((((int64_t)b * Zi + iz) * Yi + iy) * Xi + ix) * c_in + c0;
        pf[u] = ld4g(X + off);
        if (MASKED) pm[u] = ld4g(mask_src + off);
        okbits |= 1u << u;
      }
    }
  };
  // ---- registers -> LDS (affine map and ReLU mask applied here; zeros outside the grid)
  auto deposit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < PFN; ++u) {
      const int idx = tid + u * 256;
      if (idx >= total) continue;
      int hx, hy, hz;
      decode_row(idx / QPR, HX, HY, rHX, rHY, hz, hy, hx);
      const int xr = xsplit ? (hx & 1) * HXh + (hx >> 1) : hx;
      float4 v = pf[u];
      if ((okbits >> u) & 1u) {
        if (in_scale != nullptr) affine4(v, psc, psh);
        if (MASKED) keep_positive(v, pm[u]);
      }
      *reinterpret_cast<float4*>(&sX[((hz * HY + hy) * HXr + xr) * LDR + quad * 4]) = v;
    }
  };

  float* stage = sX + wave * (32 * kStagePad);
  const int c4 = lane & 7, r8 = lane >> 3;

  float4 bq[2][NB][S], aq[2][MT][S];
  fetch(0, 0);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int s = 0; s < S; ++s) bq[0][nb][s] = Wp4[s * wr8 + nb * 64];   // item 0, tap 0
  int tl = 0, ck = 0;  // tile (local) and chunk of the current item
  for (int item = 0; item < n_items; ++item) {
    __syncthreads();  // every wave is done with the previous item (taps and epilogue)
    deposit();
    __syncthreads();

    // ---- the 27 taps of this item: 13 x 2 + 1, operands ping-ponged between two register sets (no
    // copies).  Tap t's MFMAs are interleaved ONE BY ONE with the requests for tap t + 1 (weights
    // from L1 / L2, cells from LDS): a wave issues in order, so whatever sits in a block before or
    // after the MFMAs is time its SIMD's matrix pipe can only fill from the partner wave - and two
    // workgroups that started together stay in phase, both in such a block at the same time
    // (tools/micro/mfma_peak.hip: 16 MFMAs + 4 ds_read_b128 + 2 global loads interleaved sustain
    // 134 TFLOP/s with two waves per SIMD, 127 with a dozen register copies behind them).
    const float4* __restrict__ wck = Wp4 + (int64_t)ck * S * wr8;
    const int ntl = ck + 1 < nchunks ? tl : tl + 1, nck = ck + 1 < nchunks ? ck + 1 : 0;
    const float4* __restrict__ wnext = Wp4 + (int64_t)nck * S * wr8;   // the next item's first tap
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int s = 0; s < S; ++s)
        aq[0][mt][s] = *reinterpret_cast<const float4*>(&sX[rowbase[mt] * LDR + 8 * s + 4 * h]);
    if (item + 1 < n_items) fetch(ntl, nck);   // the next box: in flight during all 27 taps
    __builtin_amdgcn_sched_barrier(0);

    // one tap: requests for tap `t1` (or, past the last tap, the next item's first weights) into
    // register set `nxt`, MFMAs on set `cur`
    auto tap = [&](int t1, auto cur_tag) __attribute__((always_inline)) {
      constexpr int cur = decltype(cur_tag)::value, nxt = cur ^ 1;
      constexpr int kOther = 0x002 | 0x004 | 0x020 | 0x100;  // VALU, SALU, VMEM read, DS read
      if (t1 < 27) {
        const int kz = t1 / 9, ky = (t1 / 3) % 3, kx = t1 % 3;
        const int delta = (kz * HY + ky) * HXr + (xsplit ? (kx & 1) * HXh + (kx >> 1) : kx);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int s = 0; s < S; ++s) bq[nxt][nb][s] = wck[t1 * wtap + s * wr8 + nb * 64];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int s = 0; s < S; ++s)
            aq[nxt][mt][s] = *reinterpret_cast<const float4*>(
                &sX[(rowbase[mt] + delta) * LDR + 8 * s + 4 * h]);
      } else {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int s = 0; s < S; ++s) bq[nxt][nb][s] = wnext[s * wr8 + nb * 64];
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mt][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[cur][mt][s].x, bq[cur][nb][s].x, acc[mt][nb], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mt][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[cur][mt][s].y, bq[cur][nb][s].y, acc[mt][nb], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mt][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[cur][mt][s].z, bq[cur][nb][s].z, acc[mt][nb], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mt][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[cur][mt][s].w, bq[cur][nb][s].w, acc[mt][nb], 0, 0, 0);
      }
      // issue order: one MFMA, then up to two of the other requests, 4 * S * MT * NB times
#pragma unroll
      for (int k = 0; k < 4 * S * MT * NB; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(kOther, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll 1
    for (int t = 0; t < 26; t += 2) {
      tap(t + 1, std::integral_constant<int, 0>());
      tap(t + 2, std::integral_constant<int, 1>());
    }
    tap(27, std::integral_constant<int, 0>());   // tap 26: set 0; set 1 gets the next item's first weights
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int s = 0; s < S; ++s) bq[0][nb][s] = bq[1][nb][s];

    if (ck + 1 == nchunks) {
      // Epilogue of this tile: every 32 x 32 block goes through a wave-private LDS tile and leaves
      // as 16-byte pieces of output rows (4 store instructions per block instead of 16; the addend
      // arrives the same way).  The next item's box stays in flight meanwhile.
      __syncthreads();  // the halo tile is free
      int b, z0, y0, x0;
      tile_origin(tile_first + tl, b, z0, y0, x0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          acc_to_stage(stage, acc[mt][nb], i, h);
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][nb][r] = 0.f;
          __builtin_amdgcn_wave_barrier();
          const int n = (grp * NB + nb) * 32 + 4 * c4;
          float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
          if (bias != nullptr) bv = ld4g(bias + n);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int m = r8 + 8 * j;
            const int q = (wave * MT + mt) * 32 + m;
            const int cz = q >> lTXY;
            const int oz = z0 + cz, oy = y0 + ((q >> lTX) & TYm), ox = x0 + (q & TXm);
            float4 v = *reinterpret_cast<const float4*>(&stage[m * kStagePad + 4 * c4]);
            if (cz >= eTZ || oz >= g.Zt || oy >= g.Yt || ox >= g.Xt) continue;
            const int64_t off = ((((int64_t)b * g.Zo + oz) * g.Yo + oy) * g.Xo + ox) * c_out + n;
            v.x += bv.x, v.y += bv.y, v.z += bv.z, v.w += bv.w;
            if (addend != nullptr) {
              const float4 a = ld4g(addend + off);
              v.x += a.x, v.y += a.y, v.z += a.z, v.w += a.w;
            }
            if (relu) {
              v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
            }
            // the result is a gradient that next passes a ReLU backwards: masked here, once, instead
            // of by each of its two consumers while they stage it (grad-input and grad-weight)
            if (out_mask_src != nullptr) keep_positive(v, ld4g(out_mask_src + off));
            *reinterpret_cast<float4*>(Y + off) = v;
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    tl = ntl, ck = nck;
  }
}

// ---------------------------------------------------------------------------------------------
// dconv_kernel for the conv k3 s1 p1 with the products on the bf16 matrix cores (mfma_split.h): the
// halo box is cut into its three bf16 pieces on the way into LDS - a cell's row holds 3 x 16 bf16
// (+ 16 bytes of padding: 28 dwords, conflict-free ds_read_b128 for 16 consecutive rows) -, the
// weights arrive pre-cut in fragment order (dconv_pack_split_kernel), and six
// v_mfma_f32_32x32x16_bf16 per tap, 16-channel chunk and 32 x 32 block replace eight fp32 MFMAs of
// twice the length.  Same pipeline of items otherwise (see dconv_kernel).
//   Wq[t][c16][nb][piece][lane] (16 bytes): lane (i, h) holds the piece of
//   W[out = nb*32 + i][red = c16*16 + 8h .. + 7][tap t]
// ---------------------------------------------------------------------------------------------
constexpr int kRowW = 28;   // dwords per cell row of the split halo tile

template <int NB, int MT, int PFN, bool MASKED, bool ONE>
__global__ __launch_bounds__(256, 2) void dconv_split_kernel(
    const float* __restrict__ X, DGeom g, int c_in, const pv2::bf16x8* __restrict__ Wq, int c_out,
    int n_groups, int tiles_per_wg, const float* __restrict__ in_scale,
    const float* __restrict__ in_shift, const float* __restrict__ mask_src,
    const float* __restrict__ bias, const float* __restrict__ addend, int relu,
    const float* __restrict__ out_mask_src, float* __restrict__ Y, int ksplit, int64_t part_stride) {
  constexpr int CK = 16, QPR = 4;
  extern __shared__ __attribute__((aligned(16))) float sX[];
  unsigned* sU = reinterpret_cast<unsigned*>(sX);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
  // ksplit > 1 (round 6, the coarse levels: 64 - 128 workgroups of 8 - 16 chunks each on 256 CUs): the channel
  // chunks of a tile are dealt to `ksplit` workgroups; each writes its RAW partial sums to plane blockIdx.x %
  // ksplit of Y (= a scratch of ksplit x part_stride floats) and dconv_splitk_finish_kernel adds the planes in
  // order and applies the epilogue.
  const int s_idx = ksplit > 1 ? blockIdx.x % ksplit : 0;
  const int bid = ksplit > 1 ? blockIdx.x / ksplit : blockIdx.x;
  const int grp = bid % n_groups;
  const int n_tiles = g.B * g.nTZ * g.nTY * g.nTX;
  const int tile_first = (bid / n_groups) * tiles_per_wg;
  const int tile_count = min(tiles_per_wg, n_tiles - tile_first);
  const int nchunks_all = c_in / CK;
  const int nchunks = nchunks_all / ksplit;      // chunks of THIS workgroup: ckb .. ckb + nchunks - 1
  const int ckb = s_idx * nchunks;
  const int n_items = tile_count * nchunks;
  const int Zi = g.Zi, Yi = g.Yi, Xi = g.Xi, HX = g.HX, HY = g.HY;
  const int in_off = g.in_off;
  const int nTX = g.nTX, nTY = g.nTY, nTZ = g.nTZ, eTZ = g.eTZ, TYs = g.TY, TXs = g.TX;
  const int TXm = g.TX - 1, TYm = g.TY - 1, lTX = g.lTX, lTXY = g.lTX + g.lTY;
  const float rHX = g.rHX, rHY = g.rHY;
  const int total = g.HZ * HY * HX * QPR;
  const int quad = tid & (QPR - 1);

  int rowbase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int q = (wave * MT + mt) * 32 + i;
    const int cx = q & TXm, cy = (q >> lTX) & TYm, cz = q >> lTXY;
    rowbase[mt] = (cz * HY + cy) * HX + cx;
    if (cz >= eTZ) rowbase[mt] = 0;  // no such cell: any staged row, the result is dropped
  }

  f32x16 acc[MT][NB];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nb][r] = 0.f;

  const int nbtot = c_out >> 5;
  const int64_t wc16 = (int64_t)nbtot * 3 * 64;       // fragments per 16-channel chunk
  const int64_t wtap = (int64_t)nchunks_all * wc16;   // fragments per tap
  const pv2::bf16x8* __restrict__ Wl = Wq + (grp * NB) * 3 * 64 + lane + (int64_t)ckb * wc16;

  auto tile_origin = [&](int tile, int& b, int& z0, int& y0, int& x0) __attribute__((always_inline)) {
    const int tx = tile % nTX;
    tile /= nTX;
    const int ty = tile % nTY;
    tile /= nTY;
    b = tile / nTZ;
    z0 = (tile % nTZ) * eTZ, y0 = ty * TYs, x0 = tx * TXs;
  };

  float4 pf[PFN], pm[MASKED ? PFN : 1], psc, psh;
  unsigned okbits = 0;
  auto fetch = [&](int tl, int ck) __attribute__((always_inline)) {
    int b, z0, y0, x0;
    tile_origin(tile_first + tl, b, z0, y0, x0);
    const int hz0 = z0 + in_off, hy0 = y0 + in_off, hx0 = x0 + in_off;
    const int c0 = (ckb + ck) * CK + quad * 4;
    if (in_scale != nullptr) {
      psc = ld4g(in_scale + c0);
      psh = ld4g(in_shift + c0);
    }
    okbits = 0;
#pragma unroll
    for (int u = 0; u < PFN; ++u) {
      const int idx = tid + u * 256;
      int hx, hy, hz;
      decode_row(idx / QPR, HX, HY, rHX, rHY, hz, hy, hx);
      const int iz = hz0 + hz, iy = hy0 + hy, ix = hx0 + hx;
      const bool ok = idx < total && iz >= 0 && iz < Zi && iy >= 0 && iy < Yi && ix >= 0 && ix < Xi;
      pf[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (MASKED) pm[u] = make_float4(1.f, 1.f, 1.f, 1.f);
      if (ok) {
        const int64_t off = ((((int64_t)b * Zi + iz) * Yi + iy) * Xi + ix) * c_in + c0;
        pf[u] = ld4g(X + off);
        if (MASKED) pm[u] = ld4g(mask_src + off);
        okbits |= 1u << u;
      }
    }
  };
  // registers -> LDS: affine map, ReLU mask, then the three bf16 pieces (8 bytes each per float4)
  auto deposit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < PFN; ++u) {
      const int idx = tid + u * 256;
      if (idx >= total) continue;
      float4 v = pf[u];
      if ((okbits >> u) & 1u) {
        if (in_scale != nullptr) affine4(v, psc, psh);
        if (MASKED) keep_positive(v, pm[u]);
      }
      const float x[4] = {v.x, v.y, v.z, v.w};
      float r1[4], r2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) r1[j] = pv2::bf16_rest(x[j]), r2[j] = pv2::bf16_rest(r1[j]);
      unsigned* d = sU + (idx / QPR) * kRowW + 2 * quad;   // (box rows are stored in box order)
      *reinterpret_cast<uint2*>(d) = make_uint2(pv2::pack_hi(x[0], x[1]), pv2::pack_hi(x[2], x[3]));
      *reinterpret_cast<uint2*>(d + 8) = make_uint2(pv2::pack_hi(r1[0], r1[1]), pv2::pack_hi(r1[2], r1[3]));
      *reinterpret_cast<uint2*>(d + 16) = make_uint2(pv2::pack_hi(r2[0], r2[1]), pv2::pack_hi(r2[2], r2[3]));
    }
  };

  float* stage = sX + wave * (32 * kStagePad);
  const int c4 = lane & 7, r8 = lane >> 3;

  // Weight fragments run TWO taps ahead (three register sets, tap T lives in set T % 3 - 27 % 3 == 0, so
  // the numbering carries over from item to item): a tap is six MFMAs per block, ~200 cycles - less
  // than a trip to L2, where most of the 27 x 3 KB of a chunk's fragments live.  Cells come from LDS
  // one tap ahead (two sets).
  pv2::bf16x8 bq[3][NB][3], aq[2][MT][3];
  fetch(0, 0);
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) bq[T][nb][pc] = Wl[T * wtap + (nb * 3 + pc) * 64];   // item 0, taps 0, 1
  int tl = 0, ck = 0;
  for (int item = 0; item < n_items; ++item) {
    __syncthreads();  // every wave is done with the previous item (taps and epilogue)
    deposit();
    __syncthreads();

    const pv2::bf16x8* __restrict__ wck = Wl + (int64_t)ck * wc16;
    const int ntl = ck + 1 < nchunks ? tl : tl + 1, nck = ck + 1 < nchunks ? ck + 1 : 0;
    const pv2::bf16x8* __restrict__ wnext = Wl + (int64_t)nck * wc16;   // the next item's taps
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        aq[0][mt][pc] = *reinterpret_cast<const pv2::bf16x8*>(&sU[rowbase[mt] * kRowW + 8 * pc + 4 * h]);
    if (item + 1 < n_items) fetch(ntl, nck);   // the next box: in flight during all 27 taps
    __builtin_amdgcn_sched_barrier(0);

    // tap t: MFMAs on cell set t % 2 and weight set t % 3; requests: cells of tap t + 1, weights of
    // tap t + 2 (past the item's last tap: the next item's taps 0 and 1)
    auto tap = [&](int t, auto a_tag, auto w_tag) __attribute__((always_inline)) {
      constexpr int ac = decltype(a_tag)::value, an = ac ^ 1;
      constexpr int wc = decltype(w_tag)::value, wn = (wc + 2) % 3;
      constexpr int kOther = 0x002 | 0x004 | 0x020 | 0x100;  // VALU, SALU, VMEM read, DS read
      {
        const int T = t + 2;
        const pv2::bf16x8* __restrict__ src = T < 27 ? wck + T * wtap : wnext + (T - 27) * wtap;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) bq[wn][nb][pc] = src[(nb * 3 + pc) * 64];
      }
      if (t + 1 < 27) {
        const int t1 = t + 1;
        const int kz = t1 / 9, ky = (t1 / 3) % 3, kx = t1 % 3;
        const int delta = (kz * HY + ky) * HX + kx;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int pc = 0; pc < 3; ++pc)
            aq[an][mt][pc] = *reinterpret_cast<const pv2::bf16x8*>(
                &sU[(rowbase[mt] + delta) * kRowW + 8 * pc + 4 * h]);
      }
#define PV2_TERM(ta, tb)                                      \
  _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)           \
  _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)           \
    acc[mt][nb] = pv2::mfma_bf16(aq[ac][mt][ta], bq[wc][nb][tb], acc[mt][nb]);
      if constexpr (ONE) { PV2_TERM(0, 0) } else { PV2_SPLIT_TERMS(PV2_TERM) }
#undef PV2_TERM
      // issue order: one MFMA, then up to two of the other requests
#pragma unroll
      for (int k = 0; k < (ONE ? 1 : 6) * MT * NB; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(kOther, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
#pragma unroll 1
    for (int t = 0; t < 24; t += 6) {
      tap(t, I0(), I0());
      tap(t + 1, I1(), I1());
      tap(t + 2, I0(), I2());
      tap(t + 3, I1(), I0());
      tap(t + 4, I0(), I1());
      tap(t + 5, I1(), I2());
    }
    tap(24, I0(), I0());
    tap(25, I1(), I1());
    tap(26, I0(), I2());

    if (ck + 1 == nchunks) {
      __syncthreads();  // the halo tile is free
      int b, z0, y0, x0;
      tile_origin(tile_first + tl, b, z0, y0, x0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          acc_to_stage(stage, acc[mt][nb], i, h);
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][nb][r] = 0.f;
          __builtin_amdgcn_wave_barrier();
          const int n = (grp * NB + nb) * 32 + 4 * c4;
          float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
          if (bias != nullptr) bv = ld4g(bias + n);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int m = r8 + 8 * j;
            const int q = (wave * MT + mt) * 32 + m;
            const int cz = q >> lTXY;
            const int oz = z0 + cz, oy = y0 + ((q >> lTX) & TYm), ox = x0 + (q & TXm);
            float4 v = *reinterpret_cast<const float4*>(&stage[m * kStagePad + 4 * c4]);
            if (cz >= eTZ || oz >= g.Zt || oy >= g.Yt || ox >= g.Xt) continue;
            const int64_t off = ((((int64_t)b * g.Zo + oz) * g.Yo + oy) * g.Xo + ox) * c_out + n;
            if (ksplit > 1) {   // raw partial sums of this workgroup's chunks
              *reinterpret_cast<float4*>(Y + (int64_t)s_idx * part_stride + off) = v;
              continue;
            }
            v.x += bv.x, v.y += bv.y, v.z += bv.z, v.w += bv.w;
            if (addend != nullptr) {
              const float4 a = ld4g(addend + off);
              v.x += a.x, v.y += a.y, v.z += a.z, v.w += a.w;
            }
            if (relu) {
              v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
            }
            if (out_mask_src != nullptr) keep_positive(v, ld4g(out_mask_src + off));
            *reinterpret_cast<float4*>(Y + off) = v;
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    tl = ntl, ck = nck;
  }
}

// ---------------------------------------------------------------------------------------------
// Transposed conv k3 s2 p1, out = 2 x in per axis: out[o] = sum_t W[t] x[(o + 1 - t) / 2] over the
// taps with o + 1 - t even.  Per axis: o = 2j -> t = 1 (x[j]);  o = 2j + 1 -> t = 0 (x[j + 1]) and
// t = 2 (x[j]).  A wave owns 32 coarse cells j and ONE 32-wide output channel block; the 8 output
// parity classes have one accumulator each; tap (kz, ky, kx) belongs to class p = (k != 1) per
// axis and reads the coarse cell j + (k == 0).  Weights packed as for dconv_kernel (out = C_out of
// the transposed conv, red = its C_in).  Epilogue: + bias + addend (the decoder's skip features).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void dconvT_kernel(
    const float* __restrict__ X, DGeom g, int c_in, const float* __restrict__ Wp, int c_out,
    int n_groups, const float* __restrict__ bias, const float* __restrict__ addend,
    float* __restrict__ Y) {
  constexpr int CK = 16, LDR = CK + 4;
  extern __shared__ __attribute__((aligned(16))) float sX[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
  const int grp = blockIdx.x % n_groups;
  int tile = blockIdx.x / n_groups;
  const int tx = tile % g.nTX;
  tile /= g.nTX;
  const int ty = tile % g.nTY;
  tile /= g.nTY;
  const int tz = tile % g.nTZ;
  const int b = tile / g.nTZ;
  const int z0 = tz * g.eTZ, y0 = ty * g.TY, x0 = tx * g.TX;

  int rowbase;
  {
    const int q = wave * 32 + i;
    const int cx = q & (g.TX - 1), cy = (q >> g.lTX) & (g.TY - 1), cz = q >> (g.lTX + g.lTY);
    rowbase = cz < g.eTZ ? (cz * g.HY + cy) * g.HX + cx : 0;
  }
  f32x16 acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  const int nchunks = c_in / CK;
  const int nbtot = c_out >> 5;
  const float4* __restrict__ Wp4 = reinterpret_cast<const float4*>(Wp) + grp * 64 + lane;
  const int64_t wr8 = (int64_t)nbtot * 64;
  const int64_t wtap = (int64_t)(c_in >> 3) * wr8;

  for (int ck = 0; ck < nchunks; ++ck) {
    if (ck) __syncthreads();
    stage_chunk<CK>(sX, X, g, b, z0, y0, x0, c_in, ck, nullptr, nullptr, nullptr, tid, 256);
    __syncthreads();
    const float4* __restrict__ wck = Wp4 + (int64_t)ck * 2 * wr8;
    float4 a[8][2];  // the 8 coarse neighbours j + (dz, dy, dx)
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const int delta = (((d >> 2) & 1) * g.HY + ((d >> 1) & 1)) * g.HX + (d & 1);
#pragma unroll
      for (int s = 0; s < 2; ++s)
        a[d][s] = *reinterpret_cast<const float4*>(&sX[(rowbase + delta) * LDR + 8 * s + 4 * h]);
    }
    float4 b0 = wck[0], b1 = wck[wr8];
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const int kz = t / 9, ky = (t / 3) % 3, kx = t % 3;
      const int cls = ((kz != 1) << 2) | ((ky != 1) << 1) | (kx != 1);
      const int d = ((kz == 0) << 2) | ((ky == 0) << 1) | (kx == 0);
      float4 n0 = b0, n1 = b1;
      if (t + 1 < 27) {  // the next tap's fragments, in flight during this tap's MFMAs (pinned: the
        n0 = wck[(t + 1) * wtap];        // scheduler would sink the loads to their first use)
        n1 = wck[(t + 1) * wtap + wr8];
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][0].x, b0.x, acc[cls], 0, 0, 0);
      acc[cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][0].y, b0.y, acc[cls], 0, 0, 0);
      acc[cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][0].z, b0.z, acc[cls], 0, 0, 0);
      acc[cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][0].w, b0.w, acc[cls], 0, 0, 0);
      acc[cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][1].x, b1.x, acc[cls], 0, 0, 0);
      acc[cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][1].y, b1.y, acc[cls], 0, 0, 0);
      acc[cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][1].z, b1.z, acc[cls], 0, 0, 0);
      acc[cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][1].w, b1.w, acc[cls], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      b0 = n0, b1 = n1;
    }
  }

  __syncthreads();
  float* stage = sX + wave * (32 * kStagePad);
  const int c4 = lane & 7, r8 = lane >> 3;
  const int n = grp * 32 + 4 * c4;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias != nullptr) bv = ld4g(bias + n);
#pragma unroll
  for (int cls = 0; cls < 8; ++cls) {
    const int pz = (cls >> 2) & 1, py = (cls >> 1) & 1, px = cls & 1;
    acc_to_stage(stage, acc[cls], i, h);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = r8 + 8 * j;
      const int q = wave * 32 + m;
      const int cz = q >> (g.lTX + g.lTY);
      const int jz = z0 + cz, jy = y0 + ((q >> g.lTX) & (g.TY - 1)), jx = x0 + (q & (g.TX - 1));
      float4 v = *reinterpret_cast<const float4*>(&stage[m * kStagePad + 4 * c4]);
      if (cz >= g.eTZ || jz >= g.Zt || jy >= g.Yt || jx >= g.Xt) continue;
      const int64_t off =
          ((((int64_t)b * g.Zo + 2 * jz + pz) * g.Yo + 2 * jy + py) * g.Xo + 2 * jx + px) * c_out + n;
      v.x += bv.x, v.y += bv.y, v.z += bv.z, v.w += bv.w;
      if (addend != nullptr) {
        const float4 a = ld4g(addend + off);
        v.x += a.x, v.y += a.y, v.z += a.z, v.w += a.w;
      }
      *reinterpret_cast<float4*>(Y + off) = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------
// The strided conv k3 s2 p1 (grad-input of the transposed conv) on the bf16 matrix cores.  Its halo
// box is 8 x the tile, so only 8 channels are staged per chunk - half a 16-step MFMA.  The other half
// is a second TAP: lanes h = 0 / 1 of v_mfma_f32_32x32x16_bf16 own reduction steps 0..7 / 8..15, here
// (tap 2 tp, channels 0..7) / (tap 2 tp + 1, channels 0..7) - each half reads its own box row, the
// weights are packed by tap pair (dconv_pack_split2_kernel; tap 27 = zeros).  A cell row is three
// pieces x 8 bf16 = 12 dwords: 16 consecutive rows hit 64 distinct banks without padding.  The box's
// x axis is stored odd / even de-interleaved as in dconv_kernel.
//   Wq[tp][c8][nb][piece][lane]: lane (i, h) holds W[out = nb*32 + i][red = c8*8 .. + 7][tap 2 tp + h]
// ---------------------------------------------------------------------------------------------
constexpr int kRowS = 12;

template <int NB, int PFN, bool MASKED, bool ONE>
__global__ __launch_bounds__(256, 2) void dconv_strided_split_kernel(
    const float* __restrict__ X, DGeom g, int c_in, const pv2::bf16x8* __restrict__ Wq, int c_out,
    int n_groups, int tiles_per_wg, const float* __restrict__ mask_src,
    const float* __restrict__ out_mask_src, float* __restrict__ Y, int ksplit, int64_t part_stride) {
  constexpr int CK = 8, QPR = 2, NPAIR = 14;
  extern __shared__ __attribute__((aligned(16))) float sX[];
  unsigned* sU = reinterpret_cast<unsigned*>(sX);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
  // (ksplit: as in dconv_split_kernel)
  const int s_idx = ksplit > 1 ? blockIdx.x % ksplit : 0;
  const int bid = ksplit > 1 ? blockIdx.x / ksplit : blockIdx.x;
  const int grp = bid % n_groups;
  const int n_tiles = g.B * g.nTZ * g.nTY * g.nTX;
  const int tile_first = (bid / n_groups) * tiles_per_wg;
  const int tile_count = min(tiles_per_wg, n_tiles - tile_first);
  const int nchunks_all = c_in / CK;
  const int nchunks = nchunks_all / ksplit;
  const int ckb = s_idx * nchunks;
  const int n_items = tile_count * nchunks;
  const int Zi = g.Zi, Yi = g.Yi, Xi = g.Xi, HX = g.HX, HY = g.HY, HXr = g.HXr, HXh = g.HXh;
  const int nTX = g.nTX, nTY = g.nTY, nTZ = g.nTZ, eTZ = g.eTZ, TYs = g.TY, TXs = g.TX;
  const int TXm = g.TX - 1, TYm = g.TY - 1, lTX = g.lTX, lTXY = g.lTX + g.lTY;
  const float rHX = g.rHX, rHY = g.rHY;
  const int total = g.HZ * HY * HX * QPR;
  const int quad = tid & (QPR - 1);

  int rowbase;
  {
    const int q = wave * 32 + i;
    const int cx = q & TXm, cy = (q >> lTX) & TYm, cz = q >> lTXY;
    rowbase = ((cz * 2) * HY + cy * 2) * HXr + cx;   // even box column 2 cx -> row cx of the even half
    if (cz >= eTZ) rowbase = 0;
  }
  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  const int nbtot = c_out >> 5;
  const int64_t wc8 = (int64_t)nbtot * 3 * 64;        // fragments per 8-channel chunk
  const int64_t wpair = (int64_t)nchunks_all * wc8;   // fragments per tap pair
  const pv2::bf16x8* __restrict__ Wl = Wq + (grp * NB) * 3 * 64 + lane + (int64_t)ckb * wc8;

  auto tile_origin = [&](int tile, int& b, int& z0, int& y0, int& x0) __attribute__((always_inline)) {
    const int tx = tile % nTX;
    tile /= nTX;
    const int ty = tile % nTY;
    tile /= nTY;
    b = tile / nTZ;
    z0 = (tile % nTZ) * eTZ, y0 = ty * TYs, x0 = tx * TXs;
  };
  // box row of tap t relative to the cell's row (taps past 26: any row, their weights are zero)
  auto tap_delta = [&](int t) __attribute__((always_inline)) {
    t = t < 27 ? t : 26;
    const int kz = t / 9, ky = (t / 3) % 3, kx = t % 3;
    return (kz * HY + ky) * HXr + (kx & 1) * HXh + (kx >> 1);
  };

  float4 pf[PFN], pm[MASKED ? PFN : 1];
  unsigned okbits = 0;
  auto fetch = [&](int tl, int ck) __attribute__((always_inline)) {
    int b, z0, y0, x0;
    tile_origin(tile_first + tl, b, z0, y0, x0);
    const int hz0 = 2 * z0 - 1, hy0 = 2 * y0 - 1, hx0 = 2 * x0 - 1;
    const int c0 = (ckb + ck) * CK + quad * 4;
    okbits = 0;
#pragma unroll
    for (int u = 0; u < PFN; ++u) {
      const int idx = tid + u * 256;
      int hx, hy, hz;
      decode_row(idx / QPR, HX, HY, rHX, rHY, hz, hy, hx);
      const int iz = hz0 + hz, iy = hy0 + hy, ix = hx0 + hx;
      const bool ok = idx < total && iz >= 0 && iz < Zi && iy >= 0 && iy < Yi && ix >= 0 && ix < Xi;
      pf[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (MASKED) pm[u] = make_float4(1.f, 1.f, 1.f, 1.f);
      if (ok) {
        const int64_t off = ((((int64_t)b * Zi + iz) * Yi + iy) * Xi + ix) * c_in + c0;
        pf[u] = ld4g(X + off);
        if (MASKED) pm[u] = ld4g(mask_src + off);
        okbits |= 1u << u;
      }
    }
  };
  auto deposit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < PFN; ++u) {
      const int idx = tid + u * 256;
      if (idx >= total) continue;
      int hx, hy, hz;
      decode_row(idx / QPR, HX, HY, rHX, rHY, hz, hy, hx);
      const int xr = (hx & 1) * HXh + (hx >> 1);
      float4 v = pf[u];
      if (MASKED && ((okbits >> u) & 1u)) keep_positive(v, pm[u]);
      const float x[4] = {v.x, v.y, v.z, v.w};
      float r1[4], r2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) r1[j] = pv2::bf16_rest(x[j]), r2[j] = pv2::bf16_rest(r1[j]);
      unsigned* d = sU + ((hz * HY + hy) * HXr + xr) * kRowS + 2 * quad;
      *reinterpret_cast<uint2*>(d) = make_uint2(pv2::pack_hi(x[0], x[1]), pv2::pack_hi(x[2], x[3]));
      *reinterpret_cast<uint2*>(d + 4) = make_uint2(pv2::pack_hi(r1[0], r1[1]), pv2::pack_hi(r1[2], r1[3]));
      *reinterpret_cast<uint2*>(d + 8) = make_uint2(pv2::pack_hi(r2[0], r2[1]), pv2::pack_hi(r2[2], r2[3]));
    }
  };

  float* stage = sX + wave * (32 * kStagePad);
  const int c4 = lane & 7, r8 = lane >> 3;

  pv2::bf16x8 bq[2][NB][3], aq[2][3];
  fetch(0, 0);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) bq[0][nb][pc] = Wl[(nb * 3 + pc) * 64];   // item 0, pair 0
  int tl = 0, ck = 0;
  for (int item = 0; item < n_items; ++item) {
    __syncthreads();
    deposit();
    __syncthreads();
    const pv2::bf16x8* __restrict__ wck = Wl + (int64_t)ck * wc8;
    const int ntl = ck + 1 < nchunks ? tl : tl + 1, nck = ck + 1 < nchunks ? ck + 1 : 0;
    const pv2::bf16x8* __restrict__ wnext = Wl + (int64_t)nck * wc8;
    {
      const int row = rowbase + (h ? tap_delta(1) : tap_delta(0));
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        aq[0][pc] = *reinterpret_cast<const pv2::bf16x8*>(&sU[row * kRowS + 4 * pc]);
    }
    if (item + 1 < n_items) fetch(ntl, nck);
    __builtin_amdgcn_sched_barrier(0);

    // tap pair tp on sets `cur`; requests for pair tp + 1 (past the last: the next item's first weights)
    auto pair = [&](int tp1, auto cur_tag) __attribute__((always_inline)) {
      constexpr int cur = decltype(cur_tag)::value, nxt = cur ^ 1;
      constexpr int kOther = 0x002 | 0x004 | 0x020 | 0x100;
      if (tp1 < NPAIR) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) bq[nxt][nb][pc] = wck[tp1 * wpair + (nb * 3 + pc) * 64];
        const int row = rowbase + (h ? tap_delta(2 * tp1 + 1) : tap_delta(2 * tp1));
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
          aq[nxt][pc] = *reinterpret_cast<const pv2::bf16x8*>(&sU[row * kRowS + 4 * pc]);
      } else {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) bq[nxt][nb][pc] = wnext[(nb * 3 + pc) * 64];
      }
#define PV2_TERM(ta, tb) \
  _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) acc[nb] = pv2::mfma_bf16(aq[cur][ta], bq[cur][nb][tb], acc[nb]);
      if constexpr (ONE) { PV2_TERM(0, 0) } else { PV2_SPLIT_TERMS(PV2_TERM) }
#undef PV2_TERM
#pragma unroll
      for (int k = 0; k < (ONE ? 1 : 6) * NB; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(kOther, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll 1
    for (int tp = 0; tp < NPAIR; tp += 2) {
      pair(tp + 1, std::integral_constant<int, 0>());
      pair(tp + 2, std::integral_constant<int, 1>());   // (tp + 2 == 14: the next item's first weights into set 0)
    }

    if (ck + 1 == nchunks) {
      __syncthreads();
      int b, z0, y0, x0;
      tile_origin(tile_first + tl, b, z0, y0, x0);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        acc_to_stage(stage, acc[nb], i, h);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        __builtin_amdgcn_wave_barrier();
        const int n = (grp * NB + nb) * 32 + 4 * c4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int m = r8 + 8 * j;
          const int q = wave * 32 + m;
          const int cz = q >> lTXY;
          const int oz = z0 + cz, oy = y0 + ((q >> lTX) & TYm), ox = x0 + (q & TXm);
          float4 v = *reinterpret_cast<const float4*>(&stage[m * kStagePad + 4 * c4]);
          if (cz >= eTZ || oz >= g.Zt || oy >= g.Yt || ox >= g.Xt) continue;
          const int64_t off = ((((int64_t)b * g.Zo + oz) * g.Yo + oy) * g.Xo + ox) * c_out + n;
          if (ksplit > 1) {
            *reinterpret_cast<float4*>(Y + (int64_t)s_idx * part_stride + off) = v;
            continue;
          }
          if (out_mask_src != nullptr) keep_positive(v, ld4g(out_mask_src + off));
          *reinterpret_cast<float4*>(Y + off) = v;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    tl = ntl, ck = nck;
  }
}

// ---------------------------------------------------------------------------------------------
// dconvT_kernel on the bf16 matrix cores (mfma_split.h): the coarse box in three bf16 pieces per cell
// (28-dword rows, as dconv_split_kernel), weights pre-cut (dconv_pack_split_kernel, out = C_out of the
// transposed conv); per tap ONE class accumulator takes six MFMAs; the neighbour's cells come from LDS
// one tap ahead, the weight fragments two taps ahead.
// ---------------------------------------------------------------------------------------------
template <bool ONE>
__global__ __launch_bounds__(256, 2) void dconvT_split_kernel(
    const float* __restrict__ X, DGeom g, int c_in, const pv2::bf16x8* __restrict__ Wq, int c_out,
    int n_groups, const float* __restrict__ bias, const float* __restrict__ addend,
    float* __restrict__ Y, int ksplit, int64_t part_stride) {
  constexpr int CK = 16, QPR = 4;
  extern __shared__ __attribute__((aligned(16))) float sX[];
  unsigned* sU = reinterpret_cast<unsigned*>(sX);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
  // (ksplit: as in dconv_split_kernel - the channel chunks of a tile dealt to ksplit workgroups, raw partial
  // sums into plane s_idx, dconv_splitk_finish_kernel adds bias and the skip features)
  const int s_idx = ksplit > 1 ? blockIdx.x % ksplit : 0;
  const int bid = ksplit > 1 ? blockIdx.x / ksplit : blockIdx.x;
  const int grp = bid % n_groups;
  int tile = bid / n_groups;
  const int tx = tile % g.nTX;
  tile /= g.nTX;
  const int ty = tile % g.nTY;
  tile /= g.nTY;
  const int tz = tile % g.nTZ;
  const int b = tile / g.nTZ;
  const int z0 = tz * g.eTZ, y0 = ty * g.TY, x0 = tx * g.TX;
  const int HX = g.HX, HY = g.HY;

  int rowbase;
  {
    const int q = wave * 32 + i;
    const int cx = q & (g.TX - 1), cy = (q >> g.lTX) & (g.TY - 1), cz = q >> (g.lTX + g.lTY);
    rowbase = cz < g.eTZ ? (cz * HY + cy) * HX + cx : 0;
  }
  f32x16 acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  const int nchunks = c_in / CK;
  const int nbtot = c_out >> 5;
  const pv2::bf16x8* __restrict__ Wl = Wq + grp * 3 * 64 + lane;
  const int64_t wc16 = (int64_t)nbtot * 3 * 64;
  const int64_t wtap = (int64_t)nchunks * wc16;
  const int total = g.HZ * HY * HX * QPR;
  const int quad = tid & 3;

  const int nper = nchunks / ksplit, ckb = s_idx * nper;
  for (int cki = 0; cki < nper; ++cki) {
    const int ck = ckb + cki;
    if (cki) __syncthreads();
    // the chunk's box -> three bf16 pieces per cell (four loads in flight per thread)
    for (int base = tid; base < total; base += 256 * 4) {
      float4 v[4];
      int row[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = base + u * 256;
        row[u] = idx < total ? idx / QPR : -1;
        int hx, hy, hz;
        decode_row(idx / QPR, HX, HY, g.rHX, g.rHY, hz, hy, hx);
        const int iz = z0 + hz, iy = y0 + hy, ix = x0 + hx;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < total && iz < g.Zi && iy < g.Yi && ix < g.Xi)
          v[u] = ld4g(X + ((((int64_t)b * g.Zi + iz) * g.Yi + iy) * g.Xi + ix) * c_in + ck * CK + quad * 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (row[u] < 0) continue;
        const float x[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
        float r1[4], r2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r1[j] = pv2::bf16_rest(x[j]), r2[j] = pv2::bf16_rest(r1[j]);
        unsigned* d = sU + row[u] * kRowW + 2 * quad;
        *reinterpret_cast<uint2*>(d) = make_uint2(pv2::pack_hi(x[0], x[1]), pv2::pack_hi(x[2], x[3]));
        *reinterpret_cast<uint2*>(d + 8) = make_uint2(pv2::pack_hi(r1[0], r1[1]), pv2::pack_hi(r1[2], r1[3]));
        *reinterpret_cast<uint2*>(d + 16) = make_uint2(pv2::pack_hi(r2[0], r2[1]), pv2::pack_hi(r2[2], r2[3]));
      }
    }
    __syncthreads();
    const pv2::bf16x8* __restrict__ wck = Wl + (int64_t)ck * wc16;
    auto cells_of = [&](int t, pv2::bf16x8 (&a)[3]) __attribute__((always_inline)) {
      const int kz = t / 9, ky = (t / 3) % 3, kx = t % 3;
      const int delta = ((kz == 0) * HY + (ky == 0)) * HX + (kx == 0);   // coarse neighbour j + (k == 0)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        a[pc] = *reinterpret_cast<const pv2::bf16x8*>(&sU[(rowbase + delta) * kRowW + 8 * pc + 4 * h]);
    };
    pv2::bf16x8 wq[3][3], aq[2][3];
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) wq[T][pc] = wck[T * wtap + pc * 64];
    cells_of(0, aq[0]);
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const int kz = t / 9, ky = (t / 3) % 3, kx = t % 3;
      const int cls = ((kz != 1) << 2) | ((ky != 1) << 1) | (kx != 1);
      if (t + 2 < 27) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) wq[(t + 2) % 3][pc] = wck[(t + 2) * wtap + pc * 64];
      }
      if (t + 1 < 27) cells_of(t + 1, aq[(t + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);   // (the requests stay in front of the MFMAs)
#define PV2_TERM(ta, tb) acc[cls] = pv2::mfma_bf16(aq[t & 1][ta], wq[t % 3][tb], acc[cls]);
      if constexpr (ONE) { PV2_TERM(0, 0) } else { PV2_SPLIT_TERMS(PV2_TERM) }
#undef PV2_TERM
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  __syncthreads();
  float* stage = sX + wave * (32 * kStagePad);
  const int c4 = lane & 7, r8 = lane >> 3;
  const int n = grp * 32 + 4 * c4;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias != nullptr) bv = ld4g(bias + n);
#pragma unroll
  for (int cls = 0; cls < 8; ++cls) {
    const int pz = (cls >> 2) & 1, py = (cls >> 1) & 1, px = cls & 1;
    acc_to_stage(stage, acc[cls], i, h);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = r8 + 8 * j;
      const int q = wave * 32 + m;
      const int cz = q >> (g.lTX + g.lTY);
      const int jz = z0 + cz, jy = y0 + ((q >> g.lTX) & (g.TY - 1)), jx = x0 + (q & (g.TX - 1));
      float4 v = *reinterpret_cast<const float4*>(&stage[m * kStagePad + 4 * c4]);
      if (cz >= g.eTZ || jz >= g.Zt || jy >= g.Yt || jx >= g.Xt) continue;
      const int64_t off =
          ((((int64_t)b * g.Zo + 2 * jz + pz) * g.Yo + 2 * jy + py) * g.Xo + 2 * jx + px) * c_out + n;
      if (ksplit > 1) {
        *reinterpret_cast<float4*>(Y + (int64_t)s_idx * part_stride + off) = v;
        continue;
      }
      v.x += bv.x, v.y += bv.y, v.z += bv.z, v.w += bv.w;
      if (addend != nullptr) {
        const float4 a = ld4g(addend + off);
        v.x += a.x, v.y += a.y, v.z += a.z, v.w += a.w;
      }
      *reinterpret_cast<float4*>(Y + off) = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------
// Weight gradient: dW[n][c][t] = sum over iteration cells q of G[cellG(q, t)][n] * X[cellX(q, t)][c]
//   conv k3 s1 p1:        G = dy at q,            X = input at q + t - 1
//   transposed k3 s2 p1:  G = dy at 2q + (t != 1), X = input at q + (t == 0)       (per axis)
// MFMA: M = n (32 output channels of G), N = c (32 input channels of X), K = cells, two per
// instruction (lanes h = 0 / 1 take consecutive cells).  A workgroup (8 waves) owns one (n block,
// c block) pair and a strided list of tiles; every wave walks all cells of a tile for its own 2 - 4
// taps (one accumulator each).  Both operands come from LDS with plain ds_read_b32: a lane group reads 32
// consecutive floats of one row.  The boxes of the NEXT tile are fetched into registers while the
// MFMAs of the current one run (XI / GI: float4 per thread of the X / G box).  The result leaves as
// partial slabs part[wg][28][32][32].
// ---------------------------------------------------------------------------------------------
struct WGeom {
  int B, Zx, Yx, Xx;  // X grid
  int Zg, Yg, Xg;     // G grid
  int Zt, Yt, Xt;     // iteration grid
  int TZ, TY, TX, eTZ, lTX, lTY, nTZ, nTY, nTX;
  int HZ, HY, HX;     // staged X box; origin = tile origin * x_mul + x_off
  int x_mul, x_off;
  int GZ, GY, GX;     // staged G box; origin = tile origin * g_mul; LDS row of cell = cell * g_mul
  int g_mul;
  int transposed;     // which of the two tap -> row-delta rules applies (no tables in here: a
};                    // dynamically indexed array would move the whole argument to scratch memory)

constexpr int kWgThreads = 512;

// SHARED_A: every tap reads the same G row (conv k3 s1 p1: dG == 0) - one A fragment per cell pair.
template <int XI, int GI, bool SHARED_A>
__global__ __launch_bounds__(kWgThreads) void dconv_wgrad_kernel(
    const float* __restrict__ X, int c_x, const float* __restrict__ in_scale,
    const float* __restrict__ in_shift, const float* __restrict__ G, int c_g,
    const float* __restrict__ mask_src, WGeom geom, int n_tiles, int n_nblk, int n_cblk,
    float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // (the lambdas below are always_inline: a closure that is CALLED keeps what it captured by
  // reference in scratch memory, and every use becomes a scratch load inside the MFMA loop)
  const int Zx = geom.Zx, Yx = geom.Yx, Xx = geom.Xx, Zg = geom.Zg, Yg = geom.Yg, Xg = geom.Xg;
  const int TXm = geom.TX - 1, TYm = geom.TY - 1, lTX = geom.lTX, lTXY = geom.lTX + geom.lTY;
  const int TYs = geom.TY, TXs = geom.TX, eTZ = geom.eTZ;
  const int nTX = geom.nTX, nTY = geom.nTY, nTZ = geom.nTZ;
  const int HY = geom.HY, HX = geom.HX, GY = geom.GY, GX = geom.GX;
  const int x_mul = geom.x_mul, x_off = geom.x_off, g_mul = geom.g_mul;
  const int xrows = geom.HZ * HY * HX, grows = geom.GZ * GY * GX;
  const int ncell = geom.TZ * geom.TY * geom.TX;
  const bool transposed = geom.transposed != 0;
  float* sXw = smem;               // [xrows][32]
  float* sG = smem + xrows * 32;   // [grows][32]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
  // taps of this wave: waves 0-3 take four each (taps 0..15), waves 4-7 three, three, three, two
  // (16..26) - waves w and w + 4 share a SIMD, whose matrix pipe then serves 7, 7, 7, 6 taps
  const int t0 = __builtin_amdgcn_readfirstlane(wave < 4 ? 4 * wave : 16 + 3 * (wave - 4));
  const int cnt = __builtin_amdgcn_readfirstlane(wave < 4 ? 4 : (wave < 7 ? 3 : 2));
  const int per_slot = n_nblk * n_cblk;
  const int slot = blockIdx.x / per_slot, sub = blockIdx.x % per_slot;
  const int n_slots = gridDim.x / per_slot;
  const int nblk = sub / n_cblk, cblk = sub % n_cblk;
  const int n0 = nblk * 32, c0 = cblk * 32;

  f32x16 acc[4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
  int dXu[4], dGu[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {   // LDS row deltas of this wave's taps (tap 27 = padding, unused)
    const int t = t0 + u < 27 ? t0 + u : 26;
    const int kz = t / 9, ky = (t / 3) % 3, kx = t % 3;
    if (transposed) {
      dXu[u] = (((kz == 0) * HY + (ky == 0)) * HX + (kx == 0)) * 32;
      dGu[u] = (((kz != 1) * GY + (ky != 1)) * GX + (kx != 1)) * 32;
    } else {
      dXu[u] = ((kz * HY + ky) * HX + kx) * 32;
      dGu[u] = 0;
    }
  }

  const int oct = tid & 7;  // float4 column of a 32-channel row
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (in_scale != nullptr) {
    sc = ld4g(in_scale + c0 + oct * 4);
    sh = ld4g(in_shift + c0 + oct * 4);
  }

  float4 px[XI], pg[GI], pm[GI];
  // global -> registers: the boxes of `tile` (zeros outside the grids; the X box with the preceding
  // BatchNorm's affine map, the G box with the ReLU mask applied when they are stored)
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    int tt = tile;
    const int tx = tt % nTX;
    tt /= nTX;
    const int ty = tt % nTY;
    tt /= nTY;
    const int tz = tt % nTZ;
    const int b = tt / nTZ;
    const int z0 = tz * eTZ, y0 = ty * TYs, x0 = tx * TXs;
    const int hz0 = z0 * x_mul + x_off, hy0 = y0 * x_mul + x_off, hx0 = x0 * x_mul + x_off;
#pragma unroll
    for (int u = 0; u < XI; ++u) {
      const int idx = tid + u * kWgThreads;
      const int row = idx >> 3;
      const int hx = row % HX;
      const int t2 = row / HX;
      const int hy = t2 % HY, hz = t2 / HY;
      const int iz = hz0 + hz, iy = hy0 + hy, ix = hx0 + hx;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < xrows && iz >= 0 && iz < Zx && iy >= 0 && iy < Yx && ix >= 0 && ix < Xx) {
        v = ld4g(X + ((((int64_t)b * Zx + iz) * Yx + iy) * Xx + ix) * c_x + c0 + oct * 4);
        if (in_scale != nullptr) affine4(v, sc, sh);
      }
      px[u] = v;
    }
    const int gz0 = z0 * g_mul, gy0 = y0 * g_mul, gx0 = x0 * g_mul;
#pragma unroll
    for (int u = 0; u < GI; ++u) {
      const int idx = tid + u * kWgThreads;
      const int row = idx >> 3;
      const int hx = row % GX;
      const int t2 = row / GX;
      const int hy = t2 % GY, hz = t2 / GY;
      const int iz = gz0 + hz, iy = gy0 + hy, ix = gx0 + hx;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f), m = make_float4(1.f, 1.f, 1.f, 1.f);
      if (row < grows && iz < Zg && iy < Yg && ix < Xg) {
        const int64_t off = ((((int64_t)b * Zg + iz) * Yg + iy) * Xg + ix) * c_g + n0 + oct * 4;
        v = ld4g(G + off);
        if (mask_src != nullptr) m = ld4g(mask_src + off);
      }
      pg[u] = v;
      pm[u] = m;
    }
  };
  auto deposit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < XI; ++u) {
      const int idx = tid + u * kWgThreads;
      if ((idx >> 3) < xrows) *reinterpret_cast<float4*>(&sXw[idx * 4]) = px[u];
    }
#pragma unroll
    for (int u = 0; u < GI; ++u) {
      const int idx = tid + u * kWgThreads;
      if ((idx >> 3) < grows) {
        float4 v = pg[u];
        keep_positive(v, pm[u]);
        *reinterpret_cast<float4*>(&sG[idx * 4]) = v;
      }
    }
  };

  // The loop over this workgroup's tiles, compiled once per tap count (2, 3 or 4: a wave-uniform
  // choice made outside the loops, so the MFMA stream has no branches in it).
  auto run = [&](auto cnt_tag) __attribute__((always_inline)) {
    constexpr int CNT = decltype(cnt_tag)::value;
    if (slot < n_tiles) fetch(slot);
    const int npair = ncell >> 1;
    for (int tile = slot; tile < n_tiles; tile += n_slots) {
      __syncthreads();  // the previous tile's reads are done
      deposit();
      __syncthreads();
      if (tile + n_slots < n_tiles) fetch(tile + n_slots);  // in flight during the MFMAs below
      __builtin_amdgcn_sched_barrier(0);                    // (not to be sunk to the deposit)
      // operands of cell pair jj + 1 are read from LDS while the MFMAs of pair jj run
      float a_c[CNT], b_c[CNT], a_n[CNT], b_n[CNT];
      // the pair's first cell is wave-uniform (scalar unit); lanes add their own h / channel part
      const int lane_x = h * 32 + i, lane_g = h * g_mul * 32 + i;
      auto read_pair = [&](int jj, float (&a)[CNT], float (&bb)[CNT]) __attribute__((always_inline)) {
        const int q0 = 2 * jj;
        const int cx = q0 & TXm, cy = (q0 >> lTX) & TYm, cz = q0 >> lTXY;
        const bool real = cz < eTZ;  // (tiles of grids smaller than a full tile)
        const int rx = (real ? ((cz * HY + cy) * HX + cx) * 32 : 0) + lane_x;
        const int rg = (real ? (((cz * g_mul) * GY + cy * g_mul) * GX + cx * g_mul) * 32 : 0) + lane_g;
        if (SHARED_A) a[0] = sG[rg];
#pragma unroll
        for (int u = 0; u < CNT; ++u) {
          if (!SHARED_A) a[u] = sG[rg + dGu[u]];
          bb[u] = sXw[rx + dXu[u]];
        }
        if (!real) {  // wave-uniform
#pragma unroll
          for (int u = 0; u < CNT; ++u) a[u] = 0.f;
        }
      };
      auto mfmas = [&](const float (&a)[CNT], const float (&bb)[CNT]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < CNT; ++u)
          acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(SHARED_A ? a[0] : a[u], bb[u], acc[u], 0, 0, 0);
      };
      // two pairs per iteration, ping-pong operand sets (no register copies); npair is even
      read_pair(0, a_c, b_c);
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0) only (see dconv_kernel): the tile prefetch stays in flight
#pragma unroll 1
      for (int jj = 0; jj < npair; jj += 2) {
        read_pair(jj + 1, a_n, b_n);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(a_c, b_c);
        __builtin_amdgcn_sched_barrier(0);
        read_pair(jj + 2 < npair ? jj + 2 : 0, a_c, b_c);   // (the last one is read and not used)
        __builtin_amdgcn_sched_barrier(0);
        mfmas(a_n, b_n);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // part[wg][t][n_local][c_local]; D[m][j]: j = lane & 31 (c), m -> n
    float* dst = part + ((int64_t)blockIdx.x * 28 + t0) * 1024;
#pragma unroll
    for (int u = 0; u < CNT; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dst[u * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[u][r];
  };
  if (cnt == 4) run(std::integral_constant<int, 4>());
  else if (cnt == 3) run(std::integral_constant<int, 3>());
  else run(std::integral_constant<int, 2>());
}

// ---------------------------------------------------------------------------------------------
// The weight gradient on the bf16 matrix cores.  The reduction runs over CELLS, and a lane of
// v_mfma_f32_32x32x16_bf16 owns eight consecutive reduction steps of one channel - a column of the
// [cell][channel] boxes.  The boxes are stored as bf16 pieces, [cell][piece][32 channels] (192 bytes
// per cell), and a fragment is gathered with eight 2-byte LDS reads (lanes = channels: 64 contiguous
// bytes per cell, no bank conflicts; immediate offsets - the eight cells are consecutive along x)
// and four v_lshl_or: any tap shift costs nothing, nothing is transposed on the way in.  Per 16-cell
// step and tap: six MFMAs against 24 + 24 narrow reads - the matrix pipe stays the bound (1536
// against ~960 cycles per step and CU for the eight waves).  Same tiling, partial slabs and ordered
// reduction as dconv_wgrad_kernel; tiles are half as large (the pieces take 1.5 x the LDS).
// ---------------------------------------------------------------------------------------------
constexpr int kCellU16 = 96;   // ushorts per cell row: 3 pieces x 32 channels

template <int XI, int GI, bool SHARED_A, bool ONE>
__global__ __launch_bounds__(kWgThreads) void dconv_wgrad_split_kernel(
    const float* __restrict__ X, int c_x, const float* __restrict__ in_scale,
    const float* __restrict__ in_shift, const float* __restrict__ G, int c_g,
    const float* __restrict__ mask_src, WGeom geom, int n_tiles, int n_nblk, int n_cblk,
    float* __restrict__ part) {
  constexpr int GMUL = SHARED_A ? 1 : 2;   // G rows per iteration cell along each axis
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Zx = geom.Zx, Yx = geom.Yx, Xx = geom.Xx, Zg = geom.Zg, Yg = geom.Yg, Xg = geom.Xg;
  const int TXm = geom.TX - 1, TYm = geom.TY - 1, lTX = geom.lTX, lTXY = geom.lTX + geom.lTY;
  const int TYs = geom.TY, TXs = geom.TX, eTZ = geom.eTZ;
  const int nTX = geom.nTX, nTY = geom.nTY, nTZ = geom.nTZ;
  const int HY = geom.HY, HX = geom.HX, GY = geom.GY, GX = geom.GX;
  const int x_mul = geom.x_mul, x_off = geom.x_off;
  const int xrows = geom.HZ * HY * HX, grows = geom.GZ * GY * GX;
  const int ncell = geom.TZ * geom.TY * geom.TX;
  const bool transposed = geom.transposed != 0;
  unsigned short* sXw = reinterpret_cast<unsigned short*>(smem);   // [xrows][3][32]
  unsigned short* sG = sXw + xrows * kCellU16;                      // [grows][3][32]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
  const int t0 = __builtin_amdgcn_readfirstlane(wave < 4 ? 4 * wave : 16 + 3 * (wave - 4));
  const int cnt = __builtin_amdgcn_readfirstlane(wave < 4 ? 4 : (wave < 7 ? 3 : 2));
  const int per_slot = n_nblk * n_cblk;
  const int slot = blockIdx.x / per_slot, sub = blockIdx.x % per_slot;
  const int n_slots = gridDim.x / per_slot;
  const int nblk = sub / n_cblk, cblk = sub % n_cblk;
  const int n0 = nblk * 32, c0 = cblk * 32;

  f32x16 acc[4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
  int dXu[4], dGu[4];   // row deltas of this wave's taps, in ushorts
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int t = t0 + u < 27 ? t0 + u : 26;
    const int kz = t / 9, ky = (t / 3) % 3, kx = t % 3;
    if (transposed) {
      dXu[u] = (((kz == 0) * HY + (ky == 0)) * HX + (kx == 0)) * kCellU16;
      dGu[u] = (((kz != 1) * GY + (ky != 1)) * GX + (kx != 1)) * kCellU16;
    } else {
      dXu[u] = ((kz * HY + ky) * HX + kx) * kCellU16;
      dGu[u] = 0;
    }
  }

  const int oct = tid & 7;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (in_scale != nullptr) {
    sc = ld4g(in_scale + c0 + oct * 4);
    sh = ld4g(in_shift + c0 + oct * 4);
  }

  float4 px[XI], pg[GI], pm[GI];
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    int tt = tile;
    const int tx = tt % nTX;
    tt /= nTX;
    const int ty = tt % nTY;
    tt /= nTY;
    const int tz = tt % nTZ;
    const int b = tt / nTZ;
    const int z0 = tz * eTZ, y0 = ty * TYs, x0 = tx * TXs;
    const int hz0 = z0 * x_mul + x_off, hy0 = y0 * x_mul + x_off, hx0 = x0 * x_mul + x_off;
#pragma unroll
    for (int u = 0; u < XI; ++u) {
      const int idx = tid + u * kWgThreads;
      const int row = idx >> 3;
      const int hx = row % HX;
      const int t2 = row / HX;
      const int hy = t2 % HY, hz = t2 / HY;
      const int iz = hz0 + hz, iy = hy0 + hy, ix = hx0 + hx;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < xrows && iz >= 0 && iz < Zx && iy >= 0 && iy < Yx && ix >= 0 && ix < Xx) {
        v = ld4g(X + ((((int64_t)b * Zx + iz) * Yx + iy) * Xx + ix) * c_x + c0 + oct * 4);
        if (in_scale != nullptr) affine4(v, sc, sh);
      }
      px[u] = v;
    }
    const int gz0 = z0 * GMUL, gy0 = y0 * GMUL, gx0 = x0 * GMUL;
#pragma unroll
    for (int u = 0; u < GI; ++u) {
      const int idx = tid + u * kWgThreads;
      const int row = idx >> 3;
      const int hx = row % GX;
      const int t2 = row / GX;
      const int hy = t2 % GY, hz = t2 / GY;
      const int iz = gz0 + hz, iy = gy0 + hy, ix = gx0 + hx;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f), m = make_float4(1.f, 1.f, 1.f, 1.f);
      if (row < grows && iz < Zg && iy < Yg && ix < Xg) {
        const int64_t off = ((((int64_t)b * Zg + iz) * Yg + iy) * Xg + ix) * c_g + n0 + oct * 4;
        v = ld4g(G + off);
        if (mask_src != nullptr) m = ld4g(mask_src + off);
      }
      pg[u] = v;
      pm[u] = m;
    }
  };
  auto put = [&](unsigned short* box, int idx, const float4& v) __attribute__((always_inline)) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    float r1[4], r2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) r1[j] = pv2::bf16_rest(x[j]), r2[j] = pv2::bf16_rest(r1[j]);
    unsigned* d = reinterpret_cast<unsigned*>(box + (idx >> 3) * kCellU16) + 2 * (idx & 7);
    *reinterpret_cast<uint2*>(d) = make_uint2(pv2::pack_hi(x[0], x[1]), pv2::pack_hi(x[2], x[3]));
    *reinterpret_cast<uint2*>(d + 16) = make_uint2(pv2::pack_hi(r1[0], r1[1]), pv2::pack_hi(r1[2], r1[3]));
    *reinterpret_cast<uint2*>(d + 32) = make_uint2(pv2::pack_hi(r2[0], r2[1]), pv2::pack_hi(r2[2], r2[3]));
  };
  auto deposit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < XI; ++u) {
      const int idx = tid + u * kWgThreads;
      if ((idx >> 3) < xrows) put(sXw, idx, px[u]);
    }
#pragma unroll
    for (int u = 0; u < GI; ++u) {
      const int idx = tid + u * kWgThreads;
      if ((idx >> 3) < grows) {
        float4 v = pg[u];
        keep_positive(v, pm[u]);
        put(sG, idx, v);
      }
    }
  };
  // eight cells of one piece: cells along x are STEP rows apart
  auto frag = [&](const unsigned short* p, int step_u16) __attribute__((always_inline)) {
    pv2::u32x4 a;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      a[j] = (unsigned)p[(2 * j) * step_u16] | ((unsigned)p[(2 * j + 1) * step_u16] << 16);
    return __builtin_bit_cast(pv2::bf16x8, a);
  };

  auto run = [&](auto cnt_tag) __attribute__((always_inline)) {
    constexpr int CNT = decltype(cnt_tag)::value;
    if (slot < n_tiles) fetch(slot);
    const int nstep = ncell >> 4;   // 16-cell reduction steps (TX is a multiple of 16)
    for (int tile = slot; tile < n_tiles; tile += n_slots) {
      __syncthreads();
      deposit();
      __syncthreads();
      if (tile + n_slots < n_tiles) fetch(tile + n_slots);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
      for (int sidx = 0; sidx < nstep; ++sidx) {
        const int q0 = 16 * sidx;
        const int cx = q0 & TXm, cy = (q0 >> lTX) & TYm, cz = q0 >> lTXY;
        if (cz >= eTZ) break;   // (tiles of grids smaller than a full tile: wave-uniform)
        const unsigned short* xr = sXw + ((cz * HY + cy) * HX + cx + 8 * h) * kCellU16 + i;
        const unsigned short* gr = sG + (((cz * GMUL) * GY + cy * GMUL) * GX + (cx + 8 * h) * GMUL) * kCellU16 + i;
        pv2::bf16x8 fa[SHARED_A ? 1 : CNT][3], fb[CNT][3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          if (SHARED_A) fa[0][pc] = frag(gr + pc * 32, GMUL * kCellU16);
#pragma unroll
          for (int u = 0; u < CNT; ++u) {
            if (!SHARED_A) fa[u][pc] = frag(gr + dGu[u] + pc * 32, GMUL * kCellU16);
            fb[u][pc] = frag(xr + dXu[u] + pc * 32, kCellU16);
          }
        }
#define PV2_TERM(ta, tb) \
  _Pragma("unroll") for (int u = 0; u < CNT; ++u) acc[u] = pv2::mfma_bf16(fa[SHARED_A ? 0 : u][ta], fb[u][tb], acc[u]);
        if constexpr (ONE) { PV2_TERM(0, 0) } else { PV2_SPLIT_TERMS(PV2_TERM) }
#undef PV2_TERM
      }
    }
    float* dst = part + ((int64_t)blockIdx.x * 28 + t0) * 1024;
#pragma unroll
    for (int u = 0; u < CNT; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dst[u * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[u][r];
  };
  if (cnt == 4) run(std::integral_constant<int, 4>());
  else if (cnt == 3) run(std::integral_constant<int, 3>());
  else run(std::integral_constant<int, 2>());
}

// dW[n, c, t] = sum over the slots' partial slabs, in ascending order; written with the
// strides of the caller's weight tensor (Conv3d [n, c, kz, ky, kx] or ConvTranspose3d [c, n, ...],
// either memory format).
__global__ __launch_bounds__(256) void dconv_wgrad_reduce_kernel(
    const float* __restrict__ part, int n_slots, int n_nblk, int n_cblk, int c_g, int c_x,
    int64_t s_n, int64_t s_c, int64_t s_z, int64_t s_y, int64_t s_x, float* __restrict__ dW) {
  const int64_t total = (int64_t)27 * c_g * c_x;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % c_x);
  const int n = (int)((e / c_x) % c_g);
  const int t = (int)(e / ((int64_t)c_x * c_g));
  const int per_slot = n_nblk * n_cblk;
  const int sub = (n >> 5) * n_cblk + (c >> 5);
  const int local = (n & 31) * 32 + (c & 31);
  float s = 0.f;
  for (int slot = 0; slot < n_slots; ++slot) {
    const int64_t wg = (int64_t)slot * per_slot + sub;
    s += part[(wg * 28 + t) * 1024 + local];
  }
  dW[n * s_n + c * s_c + (t / 9) * s_z + ((t / 3) % 3) * s_y + (t % 3) * s_x] = s;
}

// packed[t][r8][nb][lane][q] = W[out = nb*32 + (lane & 31)][red = r8*8 + 4(lane >> 5) + q][tap]
// with tap = flip ? 26 - t : t.
__global__ __launch_bounds__(256) void dconv_pack_kernel(
    const float* __restrict__ W, int n_out, int n_red, int64_t s_out, int64_t s_red, int64_t s_z,
    int64_t s_y, int64_t s_x, int flip, float* __restrict__ packed) {
  const int64_t total = (int64_t)27 * n_out * n_red;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int q = (int)(e & 3);
  const int lane = (int)((e >> 2) & 63);
  int64_t rest = e >> 8;
  const int nbtot = n_out >> 5, nred8 = n_red >> 3;
  const int nb = (int)(rest % nbtot);
  rest /= nbtot;
  const int r8 = (int)(rest % nred8);
  const int t = (int)(rest / nred8);
  const int tap = flip ? 26 - t : t;
  const int out = nb * 32 + (lane & 31);
  const int red = r8 * 8 + 4 * (lane >> 5) + q;
  packed[e] = W[out * s_out + red * s_red + (tap / 9) * s_z + ((tap / 3) % 3) * s_y + (tap % 3) * s_x];
}

// Wq[t][c16][nb][piece][lane][d] (dword d = reduction steps 2d, 2d + 1 of the lane's eight): the three
// bf16 pieces of W[out = nb*32 + (lane & 31)][red = c16*16 + 8 (lane >> 5) + 2d (+1)][tap], tap as above.
__global__ __launch_bounds__(256) void dconv_pack_split_kernel(
    const float* __restrict__ W, int n_out, int n_red, int64_t s_out, int64_t s_red, int64_t s_z,
    int64_t s_y, int64_t s_x, int flip, unsigned* __restrict__ packed) {
  const int64_t total = (int64_t)27 * n_out * n_red / 2;   // (out, red pair) items; three dwords each
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int d = (int)(e & 3);
  const int lane = (int)((e >> 2) & 63);
  int64_t rest = e >> 8;
  const int nbtot = n_out >> 5, nc16 = n_red >> 4;
  const int nb = (int)(rest % nbtot);
  rest /= nbtot;
  const int c16 = (int)(rest % nc16);
  const int t = (int)(rest / nc16);
  const int tap = flip ? 26 - t : t;
  const int out = nb * 32 + (lane & 31);
  const int red = c16 * 16 + 8 * (lane >> 5) + 2 * d;
  const float* src = W + out * s_out + red * s_red + (tap / 9) * s_z + ((tap / 3) % 3) * s_y + (tap % 3) * s_x;
  const float a = src[0], b = src[s_red];
  const float a1 = pv2::bf16_rest(a), b1 = pv2::bf16_rest(b);
  unsigned* dst = packed + ((((int64_t)t * nc16 + c16) * nbtot + nb) * 3 * 64 + lane) * 4 + d;
  dst[0] = pv2::pack_hi(a, b);
  dst[64 * 4] = pv2::pack_hi(a1, b1);
  dst[2 * 64 * 4] = pv2::pack_hi(pv2::bf16_rest(a1), pv2::bf16_rest(b1));
}

// Wq[tp][c8][nb][piece][lane][d]: the pieces of W[out = nb*32 + (lane & 31)][red = c8*8 + 2d (+1)][tap],
// tap = 2 tp + (lane >> 5); zeros for tap 27 (dconv_strided_split_kernel).
__global__ __launch_bounds__(256) void dconv_pack_split2_kernel(
    const float* __restrict__ W, int n_out, int n_red, int64_t s_out, int64_t s_red, int64_t s_z,
    int64_t s_y, int64_t s_x, int flip, unsigned* __restrict__ packed) {
  const int nbtot = n_out >> 5, nc8 = n_red >> 3;
  const int64_t total = (int64_t)14 * nc8 * nbtot * 256;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int d = (int)(e & 3);
  const int lane = (int)((e >> 2) & 63);
  int64_t rest = e >> 8;
  const int nb = (int)(rest % nbtot);
  rest /= nbtot;
  const int c8 = (int)(rest % nc8);
  const int tp = (int)(rest / nc8);
  const int t = 2 * tp + (lane >> 5);
  float a = 0.f, b = 0.f;
  if (t < 27) {
    const int tap = flip ? 26 - t : t;
    const int out = nb * 32 + (lane & 31);
    const int red = c8 * 8 + 2 * d;
    const float* src = W + out * s_out + red * s_red + (tap / 9) * s_z + ((tap / 3) % 3) * s_y + (tap % 3) * s_x;
    a = src[0], b = src[s_red];
  }
  const float a1 = pv2::bf16_rest(a), b1 = pv2::bf16_rest(b);
  unsigned* dst = packed + ((((int64_t)tp * nc8 + c8) * nbtot + nb) * 3 * 64 + lane) * 4 + d;
  dst[0] = pv2::pack_hi(a, b);
  dst[64 * 4] = pv2::pack_hi(a1, b1);
  dst[2 * 64 * 4] = pv2::pack_hi(pv2::bf16_rest(a1), pv2::bf16_rest(b1));
}

int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

int pow2ceil(int v) { return 1 << ilog2(v); }

// Tile of `cells` iteration cells: x first (up to 32), then y and z alternately, none beyond the
// (power-of-two cover of the) grid; what is left over becomes cells that do not exist (*TZ > *eTZ).
void pick_tile(int cells, int Zt, int Yt, int Xt, int* TZ, int* eTZ, int* TY, int* TX) {
  const int cz = pow2ceil(Zt), cy = pow2ceil(Yt), cx = pow2ceil(Xt);
  int tx = cx < 32 ? cx : 32;
  int rest = cells / tx, ty = 1, tz = 1;
  while (rest > 1) {
    if (ty <= tz && ty < cy) ty <<= 1;
    else if (tz < cz) tz <<= 1;
    else if (ty < cy) ty <<= 1;
    else break;
    rest >>= 1;
  }
  *eTZ = tz;
  *TZ = tz * rest;
  *TY = ty;
  *TX = tx;
}

// dynamic LDS above 64 KB has to be allowed per kernel; once per kernel and size is enough
template <typename K>
int set_lds(K kernel, size_t bytes) {
  if (bytes <= 65536) return PV2_OK;
  static std::mutex mu;
  static std::unordered_map<const void*, size_t> allowed;
  const void* key = reinterpret_cast<const void*>(kernel);
  std::lock_guard<std::mutex> lock(mu);
  auto it = allowed.find(key);
  if (it != allowed.end() && it->second >= bytes) return PV2_OK;
  if (int e = pv2::hip_status(hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  (int)bytes)))
    return e;
  allowed[key] = bytes;
  return PV2_OK;
}

int env_int(const char* name, int fallback) {
  const char* e = getenv(name);
  return e ? atoi(e) : fallback;
}

// every dense convolution runs on the bf16 matrix cores (dconv_split_kernel, dconvT_split_kernel,
// dconv_strided_split_kernel) unless PV2_FP32_MFMA=1
bool g_one_term = false;   // pv2_dconv3_set_one_term

bool split_conv(int mode) {
  static const bool on = env_int("PV2_FP32_MFMA", 0) != 1;
  return on && mode >= 0 && mode <= 2;
}

}  // namespace

extern "C" {

// (mode: the pv2_dconv3_forward mode the packed weight is for - the formats differ)
int64_t pv2_dconv3_packed_floats(int c_out, int c_in, int mode) {
  const int64_t n = (int64_t)27 * c_out * c_in;
  if (split_conv(mode)) return (mode == 2 ? (int64_t)28 * c_out * c_in : n) * 3 / 2;   // three bf16 pieces
  return n;                                                       // per weight; mode 2: 14 tap pairs
}

int pv2_dconv3_pack_weights(const float* w, int n_out, int n_red, int64_t s_out, int64_t s_red,
                            int64_t s_z, int64_t s_y, int64_t s_x, int flip, int mode, float* packed,
                            pv2_stream_t stream) {
  PV2_REQUIRE(w != nullptr && packed != nullptr, "dconv3_pack_weights: null pointer");
  PV2_REQUIRE(n_out > 0 && n_out % 32 == 0 && n_red > 0 && n_red % 16 == 0,
              "dconv3_pack_weights: output channels must be a multiple of 32, reduction channels of 16");
  PV2_REQUIRE(mode >= 0 && mode <= 2, "dconv3_pack_weights: mode must be 0, 1 or 2");
  if (split_conv(mode) && mode == 2) {
    const int64_t items = (int64_t)14 * (n_red / 8) * (n_out / 32) * 256;
    hipLaunchKernelGGL(dconv_pack_split2_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, w, n_out, n_red, s_out, s_red, s_z, s_y, s_x, flip,
                       reinterpret_cast<unsigned*>(packed));
    return pv2::check_launch("dconv3_pack_weights(split, tap pairs)");
  }
  if (split_conv(mode)) {
    const int64_t items = (int64_t)27 * n_out * n_red / 2;
    hipLaunchKernelGGL(dconv_pack_split_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, w, n_out, n_red, s_out, s_red, s_z, s_y, s_x, flip,
                       reinterpret_cast<unsigned*>(packed));
    return pv2::check_launch("dconv3_pack_weights(split)");
  }
  const int64_t total = (int64_t)27 * n_out * n_red;
  hipLaunchKernelGGL(dconv_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, w, n_out, n_red, s_out, s_red, s_z, s_y, s_x, flip, packed);
  return pv2::check_launch("dconv3_pack_weights");
}

// mode 0: conv k3 s1 p1 (out grid = in grid); mode 1: transposed conv k3 s2 p1 (out = 2 x in);
// mode 2: strided conv k3 s2 p1 (out = in / 2, in even): the grad-input of mode 1.
}  // extern "C"

// Split-K finish (dconv_split_kernel / dconv_strided_split_kernel with ksplit > 1): the partial planes added in
// order, then the convolution's epilogue - bias, addend, ReLU, the ReLU mask of the tensor the result is a gradient of.
__global__ __launch_bounds__(256) void dconv_splitk_finish_kernel(
    const float4* __restrict__ part, int ksplit, int64_t n4, int c4, const float4* __restrict__ bias,
    const float4* __restrict__ addend, int relu, const float4* __restrict__ out_mask, float4* __restrict__ Y) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += stride) {
    float4 v = part[e];
    for (int k = 1; k < ksplit; ++k) {
      const float4 u = part[(int64_t)k * n4 + e];
      v.x += u.x, v.y += u.y, v.z += u.z, v.w += u.w;
    }
    if (bias != nullptr) {
      const float4 b = bias[e % c4];
      v.x += b.x, v.y += b.y, v.z += b.z, v.w += b.w;
    }
    if (addend != nullptr) {
      const float4 a = addend[e];
      v.x += a.x, v.y += a.y, v.z += a.z, v.w += a.w;
    }
    if (relu) {
      v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
    }
    if (out_mask != nullptr) keep_positive(v, out_mask[e]);
    Y[e] = v;
  }
}

// Scratch of a stream for the split-K partial planes, grown on demand.
static int splitk_workspace(hipStream_t s, int64_t floats, float** out) {
  struct Entry {
    int dev;
    hipStream_t s;
    float* p;
    int64_t cap;
  };
  static Entry table[64];
  static int used = 0;
  int dev = 0;
  if (int e = pv2::hip_status(hipGetDevice(&dev))) return e;
  Entry* hit = nullptr;
  for (int k = 0; k < used; ++k)
    if (table[k].dev == dev && table[k].s == s) hit = &table[k];
  if (hit == nullptr) {
    if (used == 64) {
      pv2::set_error("dconv3_forward: more than 64 (device, stream) pairs");
      return PV2_E_UNSUPPORTED;
    }
    table[used] = Entry{dev, s, nullptr, 0};
    hit = &table[used++];
  }
  if (hit->cap < floats) {
    if (hit->p) {   // (kernels of this stream may still read the old buffer)
      if (int e = pv2::hip_status(hipStreamSynchronize(s))) return e;
      (void)hipFree(hit->p);
      hit->p = nullptr;
      hit->cap = 0;
    }
    if (int e = pv2::hip_status(hipMalloc(reinterpret_cast<void**>(&hit->p), sizeof(float) * floats))) return e;
    hit->cap = floats;
  }
  *out = hit->p;
  return PV2_OK;
}

template <bool ONE>
static int dconv3_forward_t(const float* x, int b, int z, int y, int xx, int c_in, const float* packed_w,
                       int c_out, int mode, const float* in_scale, const float* in_shift,
                       const float* in_mask_src, const float* bias, const float* addend, int relu,
                       const float* out_mask_src, float* out, pv2_stream_t stream) {
  PV2_REQUIRE(x != nullptr && packed_w != nullptr && out != nullptr, "dconv3_forward: null pointer");
  PV2_REQUIRE(mode >= 0 && mode <= 2, "dconv3_forward: mode must be 0, 1 or 2");
  PV2_REQUIRE(c_in % 16 == 0 && c_out % 32 == 0 && c_in > 0 && c_out > 0,
              "dconv3_forward: c_in must be a multiple of 16, c_out of 32");
  PV2_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "dconv3_forward: scale and shift come together");
  PV2_REQUIRE(b > 0 && z > 0 && y > 0 && xx > 0, "dconv3_forward: empty grid");
  if (mode == 2) PV2_REQUIRE(z % 2 == 0 && y % 2 == 0 && xx % 2 == 0, "dconv3_forward: mode 2 needs even input sizes");
  if (mode == 1)
    PV2_REQUIRE(in_scale == nullptr && in_mask_src == nullptr && relu == 0 && out_mask_src == nullptr,
                "dconv3_forward: mode 1 takes bias / addend only");
  hipStream_t s = (hipStream_t)stream;
  DGeom g;
  g.B = b;
  g.Zi = z, g.Yi = y, g.Xi = xx;
  const int64_t cells_t = (int64_t)b * z * y * xx / (mode == 2 ? 8 : 1);
  if (mode == 0) {
    g.Zo = z, g.Yo = y, g.Xo = xx;
    g.Zt = z, g.Yt = y, g.Xt = xx;
  } else if (mode == 1) {
    g.Zo = 2 * z, g.Yo = 2 * y, g.Xo = 2 * xx;
    g.Zt = z, g.Yt = y, g.Xt = xx;
  } else {
    g.Zo = z / 2, g.Yo = y / 2, g.Xo = xx / 2;
    g.Zt = g.Zo, g.Yt = g.Yo, g.Xt = g.Xo;
  }
  const int nbtot = c_out / 32;
  const bool split = split_conv(mode);
  int mt = 1, nb = 1;
  // (two M-tiles per wave for the big grids; with a ReLU mask the register prefetch is twice as
  // wide - the 256-cell tile's 13 + 13 float4 spill, the 128-cell tile's 9 + 9 fit.  The split
  // kernel's halo tile is 112 bytes per cell: the 256-cell tile's box would leave one workgroup per CU)
  if (mode == 0 && cells_t >= 262144 && in_mask_src == nullptr && !split) mt = 2;
  static const int split_mt = env_int("PV2_DSPLIT_MT", 1);
  if (split && cells_t >= 262144 && in_mask_src == nullptr) mt = split_mt;
  static const int force_mt = env_int("PV2_DCONV_MT", 0), force_nb = env_int("PV2_DCONV_NB", 0);
  // (the forced two-tile variant has no masked form on the bf16-piece path: with a ReLU mask the knob is
  // ignored there, as the automatic choice already does - it used to drop the mask silently)
  if (mode == 0 && force_mt && !(split && force_mt == 2 && in_mask_src != nullptr)) mt = force_mt;
  pick_tile(128 * mt, g.Zt, g.Yt, g.Xt, &g.TZ, &g.eTZ, &g.TY, &g.TX);
  g.lTX = ilog2(g.TX), g.lTY = ilog2(g.TY);
  g.nTZ = (g.Zt + g.eTZ - 1) / g.eTZ, g.nTY = (g.Yt + g.TY - 1) / g.TY, g.nTX = (g.Xt + g.TX - 1) / g.TX;
  const int64_t n_tiles = (int64_t)b * g.nTZ * g.nTY * g.nTX;
  if (mode != 1 && mt == 1 && nbtot % 2 == 0 && n_tiles * (nbtot / 2) >= 512) nb = 2;
  // the strided form stages 8 channels of a 2x-per-axis box per round - most of its time -, and every channel
  // group of a tile repeats that staging: two output blocks per workgroup wherever they exist (round 6: 158 -> 85
  // and 66 -> 44 us on the two coarse levels, the library's 109 / 82)
  if (mode == 2 && split && mt == 1 && nbtot % 2 == 0) nb = 2;
  if (mode != 1 && mt == 1 && (force_nb == 1 || force_nb == 2) && nbtot % force_nb == 0) nb = force_nb;
  g.xsplit = 0, g.HXh = 0;
  int ck = 16;
  if (mode == 0) {
    g.HZ = g.eTZ + 2, g.HY = g.TY + 2, g.HX = g.TX + 2;
    g.in_mul = 1, g.in_off = -1, g.cell_mul = 1;
  } else if (mode == 1) {
    g.HZ = g.eTZ + 1, g.HY = g.TY + 1, g.HX = g.TX + 1;
    g.in_mul = 1, g.in_off = 0, g.cell_mul = 1;
  } else {
    g.HZ = 2 * g.eTZ + 1, g.HY = 2 * g.TY + 1, g.HX = 2 * g.TX + 1;
    g.in_mul = 2, g.in_off = -1, g.cell_mul = 2;
    g.xsplit = 1, g.HXh = g.TX + 1;  // box column hx -> row (hx & 1) * HXh + (hx >> 1): the 32 cells of
    ck = 8;                          // an M-tile read CONSECUTIVE rows (stride-2 rows conflict 2-way)
  }
  g.HXr = g.xsplit ? 2 * g.HXh : g.HX;
  g.rHX = 1.0f / (float)g.HX, g.rHY = 1.0f / (float)g.HY;
  size_t lds = (size_t)g.HZ * g.HY * g.HXr * (!split ? ck + 4 : mode == 2 ? kRowS : kRowW) * sizeof(float);
  const size_t epilogue = (size_t)4 * 32 * kStagePad * sizeof(float);
  if (lds < epilogue) lds = epilogue;
  PV2_REQUIRE(lds <= 160 * 1024, "dconv3_forward: halo tile does not fit the LDS");
  const int n_groups = nbtot / nb;
  if (mode == 1 && split) {
    int ks = 1;   // split-K where the launch would leave CUs empty (see below)
    {
      static const int force_ks = env_int("PV2_DCONV_KSPLIT", 0);
      static const int split_below = env_int("PV2_DCONV_KSPLIT_WGS", 256);
      const int nchunks = c_in / 16;
      const int64_t wgs = n_tiles * n_groups;
      // (a transposed conv writes EIGHT fine cells per coarse one: at 256 workgroups the planes are 134 MB and the
      // split loses, 69 -> 94 us; at 64 it wins, 92 -> 55)
      if (force_ks == 0 && wgs <= split_below / 2) {
        for (int k : {8, 4, 2})
          if (nchunks % k == 0 && nchunks / k >= 2 && wgs * k <= 1024) {
            ks = k;
            break;
          }
      } else if (force_ks > 1 && nchunks % force_ks == 0) {
        ks = force_ks;
      }
    }
    const int64_t out_floats_t = (int64_t)b * g.Zo * g.Yo * g.Xo * c_out;
    float* planes_t = nullptr;
    if (ks > 1)
      if (int e = splitk_workspace(s, (int64_t)ks * out_floats_t, &planes_t)) return e;
    if (int e = set_lds(dconvT_split_kernel<ONE>, lds)) return e;
    hipLaunchKernelGGL(dconvT_split_kernel<ONE>, dim3((unsigned)(n_tiles * n_groups * ks)), dim3(256), lds, s, x,
                       g, c_in, reinterpret_cast<const pv2::bf16x8*>(packed_w), c_out, n_groups, bias, addend,
                       ks > 1 ? planes_t : out, ks, out_floats_t);
    if (ks > 1) {
      const int64_t n4 = out_floats_t / 4;
      hipLaunchKernelGGL(dconv_splitk_finish_kernel, dim3(pv2::grid_for(n4, 256)), dim3(256), 0, s,
                         reinterpret_cast<const float4*>(planes_t), ks, n4, c_out / 4,
                         reinterpret_cast<const float4*>(bias), reinterpret_cast<const float4*>(addend), 0,
                         static_cast<const float4*>(nullptr), reinterpret_cast<float4*>(out));
    }
    return pv2::check_launch("dconv3_forward(transposed, split)");
  }
  if (mode == 1) {
    if (int e = set_lds(dconvT_kernel, lds)) return e;
    hipLaunchKernelGGL(dconvT_kernel, dim3((unsigned)(n_tiles * n_groups)), dim3(256), lds, s, x, g, c_in,
                       packed_w, c_out, n_groups, bias, addend, out);
    return pv2::check_launch("dconv3_forward(transposed)");
  }
  // consecutive tiles per workgroup: enough workgroups for two full rounds of the 256 CUs x 2, and
  // the deeper the pipeline the smaller the share of its un-overlapped first box
  int tpw = 1;
  static const int force_tpw = env_int("PV2_DCONV_TPW", 0);
  while (tpw < 8 && n_tiles * n_groups / (tpw * 2) >= 1024) tpw *= 2;
  if (force_tpw) tpw = force_tpw;
  // Split-K for launches that would leave CUs empty (the coarse levels: 16 tiles x a few channel groups): the
  // channel chunks of a tile go to `ksplit` workgroups, dconv_splitk_finish_kernel adds their planes in order.
  // PV2_DCONV_KSPLIT: 1 = never, k > 1 = that split wherever it divides the chunks.
  int ksplit = 1;
  if (split && (mode == 0 || mode == 2) && mt == 1) {
    static const int force_ks = env_int("PV2_DCONV_KSPLIT", 0);
    const int nchunks = c_in / ck;
    const int64_t wgs = ((n_tiles + tpw - 1) / tpw) * n_groups;
    const int min_chunks = ck == 8 ? 4 : 2;
    static const int split_below = env_int("PV2_DCONV_KSPLIT_WGS", 256);   // (launches of more workgroups stay whole)
    if (force_ks == 0 && wgs <= split_below) {
      for (int k : {8, 4, 2})
        if (nchunks % k == 0 && nchunks / k >= min_chunks && wgs * k <= 1024) {
          ksplit = k;
          break;
        }
    } else if (force_ks > 1 && nchunks % force_ks == 0) {
      ksplit = force_ks;
    }
  }
  const int64_t out_floats = (int64_t)b * g.Zo * g.Yo * g.Xo * c_out;
  float* planes = nullptr;
  if (ksplit > 1)
    if (int e = splitk_workspace(s, (int64_t)ksplit * out_floats, &planes)) return e;
  auto finish = [&](const float* f_bias, const float* f_addend, int f_relu, const float* f_mask) -> int {
    if (ksplit == 1) return PV2_OK;
    const int64_t n4 = out_floats / 4;
    hipLaunchKernelGGL(dconv_splitk_finish_kernel, dim3(pv2::grid_for(n4, 256)), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(planes), ksplit, n4, c_out / 4,
                       reinterpret_cast<const float4*>(f_bias), reinterpret_cast<const float4*>(f_addend), f_relu,
                       reinterpret_cast<const float4*>(f_mask), reinterpret_cast<float4*>(out));
    return PV2_OK;
  };
  const dim3 grid((unsigned)(((n_tiles + tpw - 1) / tpw) * n_groups * ksplit));
  const int pfn = (g.HZ * g.HY * g.HX * (ck / 4) + 255) / 256;  // float4 of the box per thread
  const bool masked = in_mask_src != nullptr;
#define PV2_DCONV_LAUNCH(NB_, MT_, CK_, PFN_, MASKED_)                                                  \
  do {                                                                                                  \
    if (int e = set_lds(dconv_kernel<NB_, MT_, CK_, PFN_, MASKED_>, lds)) return e;                     \
    hipLaunchKernelGGL((dconv_kernel<NB_, MT_, CK_, PFN_, MASKED_>), grid, dim3(256), lds, s, x, g,     \
                       c_in, packed_w, c_out, n_groups, tpw, in_scale, in_shift, in_mask_src, bias,     \
                       addend, relu, out_mask_src, out);                                                \
  } while (0)
#define PV2_DCONV_MASK(NB_, MT_, CK_, PFN_)               \
  do {                                                    \
    if (masked) PV2_DCONV_LAUNCH(NB_, MT_, CK_, PFN_, true);   \
    else PV2_DCONV_LAUNCH(NB_, MT_, CK_, PFN_, false);         \
  } while (0)
  PV2_REQUIRE(pfn <= 13, "dconv3_forward: halo box larger than the register prefetch");
  if (split && mode == 2) {
    PV2_REQUIRE(in_scale == nullptr && bias == nullptr && addend == nullptr && relu == 0,
                "dconv3_forward: mode 2 takes the masks only");
    const pv2::bf16x8* wq = reinterpret_cast<const pv2::bf16x8*>(packed_w);
#define PV2_DSTRIDE_LAUNCH(NB_, MASKED_)                                                                \
  do {                                                                                                  \
    if (int e = set_lds(dconv_strided_split_kernel<NB_, 13, MASKED_, ONE>, lds)) return e;                   \
    hipLaunchKernelGGL((dconv_strided_split_kernel<NB_, 13, MASKED_, ONE>), grid, dim3(256), lds, s, x, g,   \
                       c_in, wq, c_out, n_groups, tpw, in_mask_src, out_mask_src,                       \
                       ksplit > 1 ? planes : out, ksplit, out_floats);                                  \
  } while (0)
    if (nb == 2) {
      if (masked) PV2_DSTRIDE_LAUNCH(2, true);
      else PV2_DSTRIDE_LAUNCH(2, false);
    } else {
      if (masked) PV2_DSTRIDE_LAUNCH(1, true);
      else PV2_DSTRIDE_LAUNCH(1, false);
    }
#undef PV2_DSTRIDE_LAUNCH
    if (int e = finish(nullptr, nullptr, 0, out_mask_src)) return e;
    return pv2::check_launch("dconv3_forward(strided, split)");
  }
  if (split) {
    const pv2::bf16x8* wq = reinterpret_cast<const pv2::bf16x8*>(packed_w);
#define PV2_DSPLIT_LAUNCH_MT(NB_, MT_, PFN_, MASKED_)                                                     \
  do {                                                                                                   \
    if (int e = set_lds(dconv_split_kernel<NB_, MT_, PFN_, MASKED_, ONE>, lds)) return e;                     \
    hipLaunchKernelGGL((dconv_split_kernel<NB_, MT_, PFN_, MASKED_, ONE>), grid, dim3(256), lds, s, x, g, c_in, \
                       wq, c_out, n_groups, tpw, in_scale, in_shift, in_mask_src, bias, addend, relu,    \
                       out_mask_src, ksplit > 1 ? planes : out, ksplit, out_floats);                     \
  } while (0)
#define PV2_DSPLIT_LAUNCH(NB_, PFN_, MASKED_) PV2_DSPLIT_LAUNCH_MT(NB_, 1, PFN_, MASKED_)
    if (mt == 2) {
      PV2_DSPLIT_LAUNCH_MT(1, 2, 13, false);
      return pv2::check_launch("dconv3_forward(split)");
    }
#define PV2_DSPLIT_MASK(NB_, PFN_)                    \
  do {                                                \
    if (masked) PV2_DSPLIT_LAUNCH(NB_, PFN_, true);   \
    else PV2_DSPLIT_LAUNCH(NB_, PFN_, false);         \
  } while (0)
    if (pfn <= 9) {
      if (nb == 2) PV2_DSPLIT_MASK(2, 9);
      else PV2_DSPLIT_MASK(1, 9);
    } else {
      if (nb == 2) PV2_DSPLIT_MASK(2, 13);
      else PV2_DSPLIT_MASK(1, 13);
    }
#undef PV2_DSPLIT_MASK
#undef PV2_DSPLIT_LAUNCH
#undef PV2_DSPLIT_LAUNCH_MT
    if (int e = finish(bias, addend, relu, out_mask_src)) return e;
    return pv2::check_launch("dconv3_forward(split)");
  }
  if (mode == 2) {
    if (nb == 2) PV2_DCONV_MASK(2, 1, 8, 13);
    else PV2_DCONV_MASK(1, 1, 8, 13);
  } else if (mt == 2) {
    PV2_DCONV_MASK(1, 2, 16, 13);
  } else if (pfn <= 9) {
    if (nb == 2) PV2_DCONV_MASK(2, 1, 16, 9);
    else PV2_DCONV_MASK(1, 1, 16, 9);
  } else {
    if (nb == 2) PV2_DCONV_MASK(2, 1, 16, 13);
    else PV2_DCONV_MASK(1, 1, 16, 13);
  }
#undef PV2_DCONV_MASK
#undef PV2_DCONV_LAUNCH
  return pv2::check_launch("dconv3_forward");
}

extern "C" {

int pv2_dconv3_forward(const float* x, int b, int z, int y, int xx, int c_in, const float* packed_w,
                       int c_out, int mode, const float* in_scale, const float* in_shift,
                       const float* in_mask_src, const float* bias, const float* addend, int relu,
                       const float* out_mask_src, float* out, pv2_stream_t stream) {
  return g_one_term ? dconv3_forward_t<true>(x, b, z, y, xx, c_in, packed_w, c_out, mode, in_scale, in_shift,
                                             in_mask_src, bias, addend, relu, out_mask_src, out, stream)
                    : dconv3_forward_t<false>(x, b, z, y, xx, c_in, packed_w, c_out, mode, in_scale, in_shift,
                                              in_mask_src, bias, addend, relu, out_mask_src, out, stream);
}

// Geometry shared by the partial-size query and the launch.
static int wgrad_geometry(int b, int z, int y, int xx, int c_x, int c_g, int mode, WGeom* g,
                          int* n_tiles, int* n_slots) {
  g->B = b;
  g->Zx = z, g->Yx = y, g->Xx = xx;
  g->Zt = z, g->Yt = y, g->Xt = xx;
  const bool split = split_conv(mode);
  if (mode == 0) {
    g->Zg = z, g->Yg = y, g->Xg = xx;
    pick_tile(split ? 128 : 256, z, y, xx, &g->TZ, &g->eTZ, &g->TY, &g->TX);
    g->HZ = g->eTZ + 2, g->HY = g->TY + 2, g->HX = g->TX + 2;
    g->x_mul = 1, g->x_off = -1;
    g->GZ = g->eTZ, g->GY = g->TY, g->GX = g->TX;
    g->g_mul = 1;
  } else {
    g->Zg = 2 * z, g->Yg = 2 * y, g->Xg = 2 * xx;
    pick_tile(64, z, y, xx, &g->TZ, &g->eTZ, &g->TY, &g->TX);
    g->HZ = g->eTZ + 1, g->HY = g->TY + 1, g->HX = g->TX + 1;
    g->x_mul = 1, g->x_off = 0;
    g->GZ = 2 * g->eTZ, g->GY = 2 * g->TY, g->GX = 2 * g->TX;
    g->g_mul = 2;
  }
  g->lTX = ilog2(g->TX), g->lTY = ilog2(g->TY);
  g->nTZ = (z + g->eTZ - 1) / g->eTZ, g->nTY = (y + g->TY - 1) / g->TY, g->nTX = (xx + g->TX - 1) / g->TX;
  g->transposed = mode == 1;
  *n_tiles = b * g->nTZ * g->nTY * g->nTX;
  const int blocks = (c_g / 32) * (c_x / 32);
  int slots = 256 / blocks;   // one workgroup per CU (the LDS tile is > 80 KB)
  if (slots < 1) slots = 1;
  if (slots > *n_tiles) slots = *n_tiles;
  *n_slots = slots;
  return PV2_OK;
}

int64_t pv2_dconv3_wgrad_partial_floats(int b, int z, int y, int xx, int c_x, int c_g, int mode) {
  WGeom g;
  int n_tiles, n_slots;
  wgrad_geometry(b, z, y, xx, c_x, c_g, mode, &g, &n_tiles, &n_slots);
  return (int64_t)n_slots * (c_g / 32) * (c_x / 32) * 28 * 1024;
}

// mode 0: conv k3 s1 p1 - x [b,z,y,xx,c_x] the conv input (in_scale / in_shift: the BatchNorm in
//         front of it), gy [b,z,y,xx,c_g] the output gradient (gy_mask_src: the output, for the ReLU mask);
// mode 1: transposed conv k3 s2 p1 - x the coarse input, gy [b,2z,2y,2xx,c_g].
// dw[n*s_n + c*s_c + kz*s_z + ky*s_y + kx*s_x], n over c_g, c over c_x.
}  // extern "C"

template <bool ONE>
static int dconv3_backward_weight_t(const float* x, int b, int z, int y, int xx, int c_x,
                               const float* in_scale, const float* in_shift, const float* gy,
                               int c_g, const float* gy_mask_src, int mode, float* partial_ws,
                               float* dw, int64_t s_n, int64_t s_c, int64_t s_z, int64_t s_y,
                               int64_t s_x, pv2_stream_t stream) {
  PV2_REQUIRE(x != nullptr && gy != nullptr && partial_ws != nullptr && dw != nullptr,
              "dconv3_backward_weight: null pointer");
  PV2_REQUIRE(mode == 0 || mode == 1, "dconv3_backward_weight: mode must be 0 or 1");
  PV2_REQUIRE(c_x % 32 == 0 && c_g % 32 == 0 && c_x > 0 && c_g > 0,
              "dconv3_backward_weight: channel counts must be multiples of 32");
  PV2_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "dconv3_backward_weight: scale and shift come together");
  hipStream_t s = (hipStream_t)stream;
  WGeom g;
  int n_tiles, n_slots;
  wgrad_geometry(b, z, y, xx, c_x, c_g, mode, &g, &n_tiles, &n_slots);
  const int n_nblk = c_g / 32, n_cblk = c_x / 32;
  const int xrows = g.HZ * g.HY * g.HX, grows = g.GZ * g.GY * g.GX;
  // (mode 0 only: the transposed conv's taps each read their own gradient rows - 48 narrow LDS reads per
  // six MFMAs is more than the LDS issues; measured 1.3 x slower than the fp32 kernel)
  const bool split = split_conv(mode) && mode == 0 && g.TX % 16 == 0;
  const size_t lds = ((size_t)xrows + grows) * (split ? kCellU16 * sizeof(unsigned short) : 32 * sizeof(float));
  PV2_REQUIRE(lds <= 160 * 1024, "dconv3_backward_weight: tiles do not fit the LDS");
  const int xi = (xrows * 8 + kWgThreads - 1) / kWgThreads, gi = (grows * 8 + kWgThreads - 1) / kWgThreads;
  const dim3 grid((unsigned)(n_slots * n_nblk * n_cblk));
  if (split) {
#define PV2_WSPLIT_LAUNCH(XI_, GI_, SA_)                                                                   \
  do {                                                                                                     \
    if (int e = set_lds(dconv_wgrad_split_kernel<XI_, GI_, SA_, ONE>, lds)) return e;                           \
    hipLaunchKernelGGL((dconv_wgrad_split_kernel<XI_, GI_, SA_, ONE>), grid, dim3(kWgThreads), lds, s, x, c_x,  \
                       in_scale, in_shift, gy, c_g, gy_mask_src, g, n_tiles, n_nblk, n_cblk, partial_ws);  \
  } while (0)
    bool launched = true;
    if (xi <= 9 && gi <= 2) PV2_WSPLIT_LAUNCH(9, 2, true);
    else if (xi <= 13 && gi <= 4) PV2_WSPLIT_LAUNCH(13, 4, true);
    else launched = false;
#undef PV2_WSPLIT_LAUNCH
    if (launched) {
      const int64_t total = (int64_t)27 * c_g * c_x;
      hipLaunchKernelGGL(dconv_wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                         partial_ws, n_slots, n_nblk, n_cblk, c_g, c_x, s_n, s_c, s_z, s_y, s_x, dw);
      return pv2::check_launch("dconv3_backward_weight(split)");
    }
    pv2::set_error("dconv3_backward_weight: no split kernel variant for this tile geometry");
    return PV2_E_UNSUPPORTED;
  }
#define PV2_WGRAD_LAUNCH(XI_, GI_, SA_)                                                              \
  do {                                                                                               \
    if (int e = set_lds(dconv_wgrad_kernel<XI_, GI_, SA_>, lds)) return e;                           \
    hipLaunchKernelGGL((dconv_wgrad_kernel<XI_, GI_, SA_>), grid, dim3(kWgThreads), lds, s, x, c_x,  \
                       in_scale, in_shift, gy, c_g, gy_mask_src, g, n_tiles, n_nblk, n_cblk,         \
                       partial_ws);                                                                  \
  } while (0)
  if (mode == 1 && xi <= 4 && gi <= 8) PV2_WGRAD_LAUNCH(4, 8, false);       // the transposed conv's boxes
  else if (mode == 0 && xi <= 13 && gi <= 4) PV2_WGRAD_LAUNCH(13, 4, true); // (2, 4, 32)-cell tiles + halo
  else if (xi <= 13 && gi <= 8) PV2_WGRAD_LAUNCH(13, 8, false);             // anything else that fits
  else {
    pv2::set_error("dconv3_backward_weight: no kernel variant for this tile geometry");
    return PV2_E_UNSUPPORTED;
  }
#undef PV2_WGRAD_LAUNCH
  const int64_t total = (int64_t)27 * c_g * c_x;
  hipLaunchKernelGGL(dconv_wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                     partial_ws, n_slots, n_nblk, n_cblk, c_g, c_x, s_n, s_c, s_z, s_y, s_x, dw);
  return pv2::check_launch("dconv3_backward_weight");
}

extern "C" {

int pv2_dconv3_backward_weight(const float* x, int b, int z, int y, int xx, int c_x,
                               const float* in_scale, const float* in_shift, const float* gy,
                               int c_g, const float* gy_mask_src, int mode, float* partial_ws,
                               float* dw, int64_t s_n, int64_t s_c, int64_t s_z, int64_t s_y,
                               int64_t s_x, pv2_stream_t stream) {
  return g_one_term ? dconv3_backward_weight_t<true>(x, b, z, y, xx, c_x, in_scale, in_shift, gy, c_g, gy_mask_src,
                                                     mode, partial_ws, dw, s_n, s_c, s_z, s_y, s_x, stream)
                    : dconv3_backward_weight_t<false>(x, b, z, y, xx, c_x, in_scale, in_shift, gy, c_g,
                                                      gy_mask_src, mode, partial_ws, dw, s_n, s_c, s_z, s_y, s_x,
                                                      stream);
}

// Products of the dense convolutions on the LEADING bf16 piece of each operand only (one MFMA where the fp32
// mode issues six): the reduced-precision training mode (the reference's enable_amp = True runs these layers
// through the library's 16-bit convolutions, ponder/engines/train.py:183-196).  Operands are cut to bf16 by
// truncation, sums stay fp32, results are stored as fp32.  Process-wide; 0 restores the fp32 products.
int pv2_dconv3_set_one_term(int on) {
  g_one_term = on != 0;
  return PV2_OK;
}

}  // extern "C"
