// Output-stationary sparse convolution over MASK-GROUPED rows on gfx950: no product rows, no row
// reduce, no atomics, every output element written once in a fixed summation order.
//
// Stands in for spconv 2.x's indice_conv / indice_conv_backward behind SubMConv3d / SparseConv3d /
// SparseInverseConv3d (ponder/models/sparse_unet/spconv_unet_v1m1_base.py:47-83,112-121,135-146,
// 171-181) - the same call sites as sparse_conv.hip / sparse_conv_pr.hip, for the layers where a voxel
// has several neighbours (levels 1-4 of the U-Net: 3.5 - 11.5 of 27 offsets present).
//
// Why.  The product-row form (sparse_conv_pr.hip) keeps the matrix cores busy but writes one fp32 row
// per PAIR (37 MB per launch on the bench geometry) and needs a second, latency-bound launch to sum
// them: measured 5.9x the algorithmic traffic and 0.14 of the bf16-piece MFMA roofline for the
// operation.  An output-stationary kernel keeps the sums in the MFMA accumulators, but with rows in
// storage order a 32-row tile touches almost every offset (2.1 - 8.9x wasted matrix work here).
// Sorting the rows by the bit mask of their present offsets fixes that: on the ScanNet-shaped bench
// geometry a 32-row tile of equal-or-adjacent masks wastes 1.2 - 1.6x (tools/analyze_tiles.py), less
// than the row reduce alone cost.
//
//   plan (once per rulebook and orientation, on the geometry stream: pv2_osm_plan)
//       mask[o]   = bits k with tbl[k][o] >= 0
//       perm      = rows sorted stably by mask (rocPRIM radix sort; rows past the valid count last)
//       tblp[k][r] = tbl[k][perm[r]]  - the gather table in sorted-row order: coalesced per tile
//       tmask[t]  = OR of the masks of rows 32 t .. 32 t + 31
//   conv (spconv_osm_kernel<NB, WR, TRANS>)
//       a workgroup = WR waves = WR consecutive 32-row tiles x 32 NB output channels.  It walks the
//       offsets present in ANY of its tiles in ascending order; per offset and 32-wide reduction slab
//       the weight slab goes global -> registers -> three bf16 piece planes in LDS (shared by the WR
//       waves), every wave gathers its own 32 rows (absent neighbours read a row of zeros) and - if ITS
//       tile has the offset at all - issues the six bf16 MFMAs per 16 reduction steps and column
//       block of mfma_split.h into accumulators that live across offsets.  Loads run two slabs ahead,
//       across offset boundaries.  Epilogue: (+ addend) -> 16-byte stores of whole row segments to
//       out[perm[r]]; optionally the BatchNorm statistics of the block (sum and centred sum of
//       squares per channel: Chan's parallel form, combined in double by rownorm.hip).
//
// Summation order of an output element: offsets ascending (table order), reduction steps in MFMA
// order, pieces smallest first - fixed by the plan, not by scheduling: bitwise reproducible.
#include <stdlib.h>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"
#include "mfma_split.h"

namespace {

using pv2::f32x16;

constexpr int kKC = 32;      // reduction steps per slab
constexpr int kRowDw = 20;   // dwords per row of a piece plane: 32 bf16 + 16 bytes of padding
constexpr int kStagePad = 36;
constexpr int kMaxK = 32;

// ---------------------------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------------------------
__global__ void osm_keys_kernel(const int32_t* __restrict__ tbl, int K, int64_t n_cols, int64_t stride,
                                const int32_t* __restrict__ n_cols_dev, uint32_t* __restrict__ keys,
                                int32_t* __restrict__ vals) {
  const int64_t cols = n_cols_dev ? min((int64_t)*n_cols_dev, n_cols) : n_cols;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_cols; i += step) {
    uint32_t m = 0;
    if (i < cols) {
      for (int k = 0; k < K; ++k) m |= (uint32_t)(tbl[(int64_t)k * stride + i] >= 0) << k;
    } else {
      m = 0xffffffffu;   // rows past the valid count sort last (a valid mask has at most 31... K <= 31 bits)
    }
    keys[i] = m;
    vals[i] = (int32_t)i;
  }
}

// one thread per sorted position r < n_pad
__global__ __launch_bounds__(256) void osm_table_kernel(
    const int32_t* __restrict__ tbl, int K, int64_t n_cols, int64_t stride,
    const int32_t* __restrict__ n_cols_dev, const uint32_t* __restrict__ keys_sorted,
    const int32_t* __restrict__ perm, int64_t n_pad, int32_t* __restrict__ tblp,
    uint32_t* __restrict__ tmask) {
  const int64_t cols = n_cols_dev ? min((int64_t)*n_cols_dev, n_cols) : n_cols;
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_pad) return;   // (n_pad is a multiple of 256: whole waves)
  const bool valid = r < cols;
  const int src = valid ? perm[r] : -1;
  uint32_t m = valid ? keys_sorted[r] : 0u;
  for (int k = 0; k < K; ++k)
    tblp[(int64_t)k * n_pad + r] = ((m >> k) & 1u) ? tbl[(int64_t)k * stride + src] : -1;
#pragma unroll
  for (int d = 1; d < 32; d <<= 1) m |= (uint32_t)__shfl_xor((int)m, d);
  if ((threadIdx.x & 31) == 0) tmask[r >> 5] = m;
}

// ---------------------------------------------------------------------------------------------
// conv
// ---------------------------------------------------------------------------------------------
struct Cursor {   // position in the (present offset, slab) sequence of a workgroup: all uniform
  uint32_t rem;   // offsets not yet finished (lowest set bit = the current one)
  int k, s;
};

template <int NB, int WR, bool TRANS>
__global__ __launch_bounds__(64 * WR) void spconv_osm_kernel(
    const float* __restrict__ X, int c_in, const float* __restrict__ W, int K, int c_out,
    const int32_t* __restrict__ tblp, int64_t n_pad, const int32_t* __restrict__ perm,
    const uint32_t* __restrict__ tmask, int kflip, int64_t n_out, const float* __restrict__ zero_row,
    const float* addend, float* Y, float* __restrict__ partial, int n_groups) {
  constexpr int NT = 32 * NB, THREADS = 64 * WR, ROWS = 32 * WR;
  constexpr int kSlabDw = 3 * NT * kRowDw;
  constexpr int kStageDw = WR * 32 * kStagePad;
  constexpr int kBufDw = 2 * kSlabDw > kStageDw ? 2 * kSlabDw : kStageDw;
  extern __shared__ __attribute__((aligned(16))) unsigned sMem[];
  unsigned* sP = sMem;
  int* s_idx = reinterpret_cast<int*>(sMem + kBufDw);   // [K][ROWS]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 31, h = lane >> 5;
  const int rowblk = blockIdx.x / n_groups, grp = blockIdx.x % n_groups;
  const int64_t r0 = (int64_t)rowblk * ROWS;
  const int n0 = grp * NT;

  // offsets present in the workgroup's tiles / in this wave's tile
  uint32_t um = 0, mine = 0;
#pragma unroll
  for (int w = 0; w < WR; ++w) {
    const uint32_t m = tmask[(int64_t)rowblk * WR + w];
    um |= m;
    if (w == wave) mine = m;
  }
  const uint32_t my_mask = (uint32_t)__builtin_amdgcn_readfirstlane((int)mine);
  // this lane's output row (sorted position r0 + 32 wave + i)
  const int64_t rpos = r0 + wave * 32 + i;
  const int orow = rpos < n_out ? perm[rpos] : -1;

  {
    // the tile's slice of the gather table -> LDS: ALL of a thread's entries are requested before the first
    // is stored (a load-store loop paid one memory round trip per trip: 5 - 10 us of prologue per launch)
    constexpr int NE = kMaxK * ROWS / THREADS;
    int v[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const int e = tid + THREADS * j;
      const int k = e / ROWS, r = e - k * ROWS;
      v[j] = e < K * ROWS ? tblp[(int64_t)k * n_pad + r0 + r] : -1;
    }
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const int e = tid + THREADS * j;
      if (e < K * ROWS) s_idx[e] = v[j];
    }
  }

  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  const int nslab = c_in / kKC;
  const int total = __builtin_popcount(um) * nslab;
  __syncthreads();   // s_idx

  if (total > 0) {
    // weight staging items.  Plain ([c_out, K, c_in]): item q = (row q >> 3, 16-byte column q & 7) of
    // the [NT x 32] slab.  TRANS ([c_red, K, c_out], the forward weight read in place by grad-input):
    // item q = step pair q & 15 of the four output channels 4 (q >> 4) .. + 3.
    constexpr int ITEMS = TRANS ? 128 * NB : 256 * NB;
    constexpr int NW = (ITEMS + THREADS - 1) / THREADS;
    struct Slab {
      float4 a[4];
      float4 wa[NW], wb[TRANS ? NW : 1];
    };
    Slab sl[2];
    auto ldu4 = [](const float* q) __attribute__((always_inline)) { return *reinterpret_cast<const float4*>(q); };

    auto advance = [&](Cursor& c) __attribute__((always_inline)) {   // branch-free, clamps at the end
      const bool wrap = (c.s + 1 == nslab);
      const uint32_t nrem = c.rem & (c.rem - 1);
      const bool adv = wrap && nrem != 0;
      const bool hold = wrap && nrem == 0;
      c.s = adv ? 0 : (hold ? c.s : c.s + 1);
      c.rem = adv ? nrem : c.rem;
      c.k = __builtin_ctz(c.rem);
    };
    // (UNCONDITIONAL loads, as in spconv_fwd_lds_kernel: rows without the neighbour read the zero row,
    // weight rows / columns past c_out are clamped to the last ones - none of it is stored)
    auto load_slab = [&](Slab& d, const Cursor& c) __attribute__((always_inline)) {
      const int idx = s_idx[c.k * ROWS + wave * 32 + i];
      const float* xr = (idx >= 0 ? X + (int64_t)idx * c_in : zero_row) + 8 * h + c.s * kKC;
#pragma unroll
      for (int s = 0; s < 4; ++s) d.a[s] = ldu4(xr + 16 * (s >> 1) + 4 * (s & 1));
      const int kw = kflip ? K - 1 - c.k : c.k;
#pragma unroll
      for (int u = 0; u < NW; ++u) {
        const int q = min(tid + THREADS * u, ITEMS - 1);
        if (!TRANS) {
          const int row = min(n0 + (q >> 3), c_out - 1);
          d.wa[u] = ldu4(W + ((int64_t)row * K + kw) * c_in + c.s * kKC + 4 * (q & 7));
        } else {
          const float* src = W + ((int64_t)(c.s * kKC + 2 * (q & 15)) * K + kw) * c_out +
                             min(n0 + 4 * (q >> 4), c_out - 4);
          d.wa[u] = ldu4(src);
          d.wb[u] = ldu4(src + (int64_t)K * c_out);
        }
      }
    };
    auto store_items = [&](const Slab& d, int buf) __attribute__((always_inline)) {
      unsigned* dst = sP + buf * kSlabDw;
#pragma unroll
      for (int u = 0; u < NW; ++u) {
        const int q = tid + THREADS * u;
        if (ITEMS % THREADS != 0 && q >= ITEMS) continue;
        if (!TRANS) {
          const float x[4] = {d.wa[u].x, d.wa[u].y, d.wa[u].z, d.wa[u].w};
          float r1[4], r2[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) r1[j] = pv2::bf16_rest(x[j]), r2[j] = pv2::bf16_rest(r1[j]);
          unsigned* o = dst + (q >> 3) * kRowDw + 2 * (q & 7);
          *reinterpret_cast<uint2*>(o) = make_uint2(pv2::pack_hi(x[0], x[1]), pv2::pack_hi(x[2], x[3]));
          *reinterpret_cast<uint2*>(o + NT * kRowDw) =
              make_uint2(pv2::pack_hi(r1[0], r1[1]), pv2::pack_hi(r1[2], r1[3]));
          *reinterpret_cast<uint2*>(o + 2 * NT * kRowDw) =
              make_uint2(pv2::pack_hi(r2[0], r2[1]), pv2::pack_hi(r2[2], r2[3]));
        } else {
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
    };

    Cursor lc, cc;   // load cursor (two slabs ahead), compute cursor
    lc.rem = cc.rem = um;
    lc.k = cc.k = __builtin_ctz(um);
    lc.s = cc.s = 0;
    float4 a_cur[4];
    auto iteration = [&](int t, Slab& nxt, Slab& far) __attribute__((always_inline)) {
      const int buf = t & 1;
      load_slab(far, lc);
      advance(lc);
      __builtin_amdgcn_sched_barrier(0);   // (those loads stay in front of the MFMAs)
      if ((my_mask >> cc.k) & 1u) {        // (wave-uniform: this tile has the offset)
        const unsigned* src = sP + buf * kSlabDw + i * kRowDw + 4 * h;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const pv2::Split8 pa = pv2::split8(a_cur[2 * st], a_cur[2 * st + 1]);
          pv2::bf16x8 pb[NB][3];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
              pb[nb][pc] = *reinterpret_cast<const pv2::bf16x8*>(src + (pc * NT + nb * 32) * kRowDw + 8 * st);
#define PV2_TERM(ta, tb) \
  _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) acc[nb] = pv2::mfma_bf16(pa.p[ta], pb[nb][tb], acc[nb]);
          PV2_SPLIT_TERMS(PV2_TERM)
#undef PV2_TERM
        }
      }
      advance(cc);
      __builtin_amdgcn_sched_barrier(0);
      store_items(nxt, buf ^ 1);
#pragma unroll
      for (int s = 0; s < 4; ++s) a_cur[s] = nxt.a[s];
      __syncthreads();
    };
    load_slab(sl[0], lc);
    advance(lc);
    store_items(sl[0], 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) a_cur[s] = sl[0].a[s];
    load_slab(sl[1], lc);
    advance(lc);
    __syncthreads();
    int t = 0;
    for (; t + 1 < total; t += 2) {
      iteration(t, sl[1], sl[0]);
      iteration(t + 1, sl[0], sl[1]);
    }
    if (t < total) iteration(t, sl[1], sl[0]);
  }

  // ---- epilogue.  acc[nb][r] = row (r & 3) + 8 (r >> 2) + 4 h of the wave's tile, channel n0 + 32 nb + i.
  float* sF = reinterpret_cast<float*>(sP);
  if (partial != nullptr) {
    // BatchNorm statistics of this block of rows: partial[blk][0..c) = sum, [c..2c) = sum of squares
    // about the BLOCK's mean (rows past n_out do not count); fixed order: wave halves, then waves
    const int64_t left = n_out - r0;
    const int nrows = left < ROWS ? (int)left : ROWS;
    const int wrow0 = wave * 32 + 4 * h;
    float* red = sF;            // [WR][NT]
    float* mean = sF + WR * NT; // [NT]
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (wrow0 + (r & 3) + 8 * (r >> 2) < nrows) s += acc[nb][r];
      s += __shfl_xor(s, 32);
      if (h == 0) red[wave * NT + nb * 32 + i] = s;
    }
    __syncthreads();
    float* prow = partial + (int64_t)rowblk * 2 * c_out;
    for (int c = tid; c < NT; c += THREADS) {
      float s = red[c];
#pragma unroll
      for (int w = 1; w < WR; ++w) s += red[w * NT + c];
      mean[c] = s / (float)nrows;
      if (n0 + c < c_out) prow[n0 + c] = s;
    }
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const float m = mean[nb * 32 + i];
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (wrow0 + (r & 3) + 8 * (r >> 2) < nrows) {
          const float d = acc[nb][r] - m;
          s += d * d;
        }
      s += __shfl_xor(s, 32);
      if (h == 0) red[wave * NT + nb * 32 + i] = s;
    }
    __syncthreads();
    for (int c = tid; c < NT; c += THREADS) {
      float s = red[c];
#pragma unroll
      for (int w = 1; w < WR; ++w) s += red[w * NT + c];
      if (n0 + c < c_out) prow[c_out + n0 + c] = s;
    }
    __syncthreads();
  }
  // rows leave through a wave-private LDS tile as 16-byte pieces: 8 lanes per 128-byte row segment
  float* stage = sF + wave * 32 * kStagePad;
  const int c4 = lane & 7, r8 = lane >> 3;
  int orow_j[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) orow_j[j] = __shfl(orow, r8 + 8 * j);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * h) * kStagePad + i] = acc[nb][r];
    __builtin_amdgcn_wave_barrier();
    const int n = n0 + nb * 32 + 4 * c4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 v = *reinterpret_cast<const float4*>(&stage[(r8 + 8 * j) * kStagePad + 4 * c4]);
      if (orow_j[j] >= 0 && n < c_out) {
        float* dst = Y + (int64_t)orow_j[j] * c_out + n;
        if (addend != nullptr) {
          const float4 a = *reinterpret_cast<const float4*>(addend + (int64_t)orow_j[j] * c_out + n);
          v.x += a.x, v.y += a.y, v.z += a.z, v.w += a.w;
        }
        *reinterpret_cast<float4*>(dst) = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e && e[0] ? atoi(e) : dflt;
}

// (NB, WR) of a launch: the widest column group (gathered rows are split into bf16 pieces once per
// group) and four row tiles per workgroup (one weight slab serves four waves) that still give the
// 256 CUs about two workgroups each; small layers trade reuse for workgroups.
int g_force_nb = env_int("PV2_OSM_NB", 0), g_force_wr = env_int("PV2_OSM_WR", 0);
int g_min_wgs = env_int("PV2_OSM_MIN_WGS", 448);
int g_mode = -1;   // PV2_CONV_OSM: 0 never, 1 every planned conv, 2 auto; -1: not read yet

void osm_pick(int64_t n_out, int c_out, int* nb_out, int* wr_out) {
  const int force_nb = g_force_nb, force_wr = g_force_wr, want = g_min_wgs;
  const int nblk = (c_out + 31) / 32;
  const int64_t tiles = (n_out + 31) / 32;
  int best_nb = 1, best_wr = 2;
  bool found = false;
  for (int wr = 4; wr >= 2 && !found; wr -= 2)
    for (int nb = 4; nb >= 1 && !found; --nb) {
      if (nblk % nb) continue;
      const int64_t wgs = ((tiles + wr - 1) / wr) * (nblk / nb);
      if (wgs >= want) best_nb = nb, best_wr = wr, found = true;
    }
  if (force_nb >= 1 && force_nb <= 4) best_nb = force_nb;
  if (force_wr == 2 || force_wr == 4) best_wr = force_wr;
  *nb_out = best_nb;
  *wr_out = best_wr;
}

template <int NB, int WR, bool TRANS>
int launch_osm(const float* X, int c_in, const float* W, int K, int c_out, const int32_t* tblp,
               int64_t n_pad, const int32_t* perm, const uint32_t* tmask, int kflip, int64_t n_out,
               const float* zero_row, const float* addend, float* Y, float* partial, hipStream_t s) {
  constexpr int NT = 32 * NB, ROWS = 32 * WR;
  constexpr int kSlabDw = 3 * NT * kRowDw;
  constexpr int kStageDw = WR * 32 * kStagePad;
  constexpr int kBufDw = 2 * kSlabDw > kStageDw ? 2 * kSlabDw : kStageDw;
  const size_t lds = ((size_t)kBufDw + (size_t)K * ROWS) * 4;
  const int n_groups = (c_out + NT - 1) / NT;
  const int64_t blocks = ((n_out + ROWS - 1) / ROWS) * n_groups;
  if (blocks > 0x7fffffffLL) {
    pv2::set_error("pv2_spconv_osm: grid too large");
    return PV2_E_BADARG;
  }
  if (lds > 65536) {
    static bool allowed = false;   // (per instantiation)
    if (!allowed) {
      if (int e = pv2::hip_status(hipFuncSetAttribute(
              reinterpret_cast<const void*>(&spconv_osm_kernel<NB, WR, TRANS>),
              hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)))
        return e;
      allowed = true;
    }
  }
  hipLaunchKernelGGL((spconv_osm_kernel<NB, WR, TRANS>), dim3((unsigned)blocks), dim3(64 * WR), lds, s,
                     X, c_in, W, K, c_out, tblp, n_pad, perm, tmask, kflip, n_out, zero_row, addend, Y,
                     partial, n_groups);
  return pv2::check_launch("spconv_osm");
}

template <bool TRANS>
int dispatch_osm(int nb, int wr, const float* X, int c_in, const float* W, int K, int c_out,
                 const int32_t* tblp, int64_t n_pad, const int32_t* perm, const uint32_t* tmask,
                 int kflip, int64_t n_out, const float* zero_row, const float* addend, float* Y,
                 float* partial, hipStream_t s) {
#define PV2_OSM_ARGS X, c_in, W, K, c_out, tblp, n_pad, perm, tmask, kflip, n_out, zero_row, addend, Y, partial, s
  if (wr == 4) {
    switch (nb) {
      case 1: return launch_osm<1, 4, TRANS>(PV2_OSM_ARGS);
      case 2: return launch_osm<2, 4, TRANS>(PV2_OSM_ARGS);
      case 3: return launch_osm<3, 4, TRANS>(PV2_OSM_ARGS);
      default: return launch_osm<4, 4, TRANS>(PV2_OSM_ARGS);
    }
  }
  switch (nb) {
    case 1: return launch_osm<1, 2, TRANS>(PV2_OSM_ARGS);
    case 2: return launch_osm<2, 2, TRANS>(PV2_OSM_ARGS);
    case 3: return launch_osm<3, 2, TRANS>(PV2_OSM_ARGS);
    default: return launch_osm<4, 2, TRANS>(PV2_OSM_ARGS);
  }
#undef PV2_OSM_ARGS
}

}  // namespace

namespace pv2 {

int osm_rows_per_block(int64_t n_out, int c_out) {
  int nb, wr;
  osm_pick(n_out, c_out, &nb, &wr);
  return 32 * wr;
}

// PV2_CONV_OSM = 0: never; 1: every conv that carries a plan; auto (default): the shapes where it
// measured faster than the product-row route on MI355X (profiles/r05_spconv_ab.txt).
// PV2_FP32_MFMA=1 (every product on the fp32 MFMA) keeps the product-row route: this kernel has the
// bf16-piece form only.
bool use_osm(const pv2_osm_plan_t* plan, const float* zero_row, int K, int64_t n_rows, int64_t n_other,
             int c_red, int c_cols) {
  if (g_mode < 0) {
    const char* f = getenv("PV2_FP32_MFMA");
    const char* e = getenv("PV2_CONV_OSM");
    g_mode = (f != nullptr && f[0] == '1') ? 0
             : (e == nullptr || e[0] == 0) ? 2 : e[0] == '0' ? 0 : e[0] == '1' ? 1 : 2;
  }
  const int mode = g_mode;
  if (mode == 0 || plan == nullptr || plan->tblp == nullptr || zero_row == nullptr) return false;
  if (K < 1 || K >= kMaxK || c_red < kKC || (c_red % kKC) != 0 || (c_cols % 4) != 0) return false;
  if (n_rows < 1 || n_rows > plan->n_pad || (n_rows + 63) / 64 > PV2_BN_MAX_PARTIAL_BLOCKS) return false;
  if (mode == 1) return true;
  // auto, from the A/B on the bench geometry (profiles/r05_spconv_ab.txt).  Per launch this route wins
  // for the strided / inverse convs whose walked rows are the FINE side (every row has exactly one
  // offset: the mask order makes every tile a dense single-offset GEMM - grad-input of a strided conv,
  // forward of an inverse conv) and for the narrow strided forward passes; the 27-offset submanifold convs
  // lose (their workgroups walk 10 - 17 offsets x c_in / 32 slabs serially: 1.0 - 4x slower, DESIGN.md
  // section 3.2c).  The host side builds plans only when asked to (PV2_CONV_OSM=1): the eight plans of
  // the strided levels cost more on the geometry stream than the 62 us per step those launches save.
  if (K > 8) return false;
  return n_rows >= n_other || c_red <= 64;
}

int spconv_osm(bool trans, const float* in_feat, int c_in, const float* weight, int K, int c_out,
               const pv2_osm_plan_t* plan, int64_t n_out, const float* zero_row, const float* addend,
               float* out, float* bn_partial, int* bn_blocks, int* bn_rows_per_block, hipStream_t s) {
  PV2_REQUIRE(plan != nullptr && plan->tblp != nullptr && plan->perm != nullptr && plan->tmask != nullptr,
              "pv2_spconv_osm: the rulebook carries no output-stationary plan");
  PV2_REQUIRE(K >= 1 && K < kMaxK, "pv2_spconv_osm: 1 <= K <= 31");
  PV2_REQUIRE(c_in >= kKC && (c_in % kKC) == 0, "pv2_spconv_osm: c_in must be a multiple of 32");
  PV2_REQUIRE(c_out >= 4 && (c_out % 4) == 0, "pv2_spconv_osm: c_out must be a multiple of 4");
  PV2_REQUIRE(zero_row != nullptr, "pv2_spconv_osm: needs the row of zeros");
  PV2_REQUIRE(n_out >= 0 && n_out <= plan->n_pad && (plan->n_pad % 256) == 0,
              "pv2_spconv_osm: plan smaller than the output");
  if (bn_blocks) *bn_blocks = 0;
  if (n_out == 0) return PV2_OK;
  int nb, wr;
  osm_pick(n_out, c_out, &nb, &wr);
  if (bn_partial) {
    const int64_t blocks = (n_out + 32 * wr - 1) / (32 * wr);
    PV2_REQUIRE(blocks <= PV2_BN_MAX_PARTIAL_BLOCKS, "pv2_spconv_osm: too many statistics blocks");
    if (bn_blocks) *bn_blocks = (int)blocks;
    if (bn_rows_per_block) *bn_rows_per_block = 32 * wr;
  }
  if (trans)
    return dispatch_osm<true>(nb, wr, in_feat, c_in, weight, K, c_out, plan->tblp, plan->n_pad, plan->perm,
                              plan->tmask, plan->kflip, n_out, zero_row, addend, out, bn_partial, s);
  return dispatch_osm<false>(nb, wr, in_feat, c_in, weight, K, c_out, plan->tblp, plan->n_pad, plan->perm,
                             plan->tmask, plan->kflip, n_out, zero_row, addend, out, bn_partial, s);
}

}  // namespace pv2

extern "C" {

int pv2_debug_set_osm(int mode, int nb, int wr, int min_wgs) {
  if (mode >= 0) g_mode = mode > 2 ? 2 : mode;
  if (nb >= 0) g_force_nb = nb;
  if (wr >= 0) g_force_wr = wr;
  if (min_wgs > 0) g_min_wgs = min_wgs;
  return PV2_OK;
}

size_t pv2_osm_plan_workspace_bytes(int64_t n_cols) {
  size_t need = 0;
  const size_t n = (size_t)(n_cols > 0 ? n_cols : 1);
  (void)rocprim::radix_sort_pairs(nullptr, need, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                  (int32_t*)nullptr, (int32_t*)nullptr, n);
  // keys, sorted keys, values (256-byte aligned pieces), then rocPRIM's own scratch
  const size_t piece = (n * 4 + 255) / 256 * 256;
  return 3 * piece + need + 256;
}

int pv2_osm_plan(const int32_t* tbl, int K, int64_t n_cols, int64_t stride, const int32_t* n_cols_dev,
                 int32_t* perm, int32_t* tblp, uint32_t* tmask, int64_t n_pad, void* workspace,
                 size_t workspace_bytes, pv2_stream_t stream) {
  PV2_REQUIRE(K >= 1 && K < kMaxK, "pv2_osm_plan: 1 <= K <= 31");
  PV2_REQUIRE(n_cols >= 0 && stride >= n_cols && n_pad >= n_cols && (n_pad % 256) == 0 && n_pad > 0,
              "pv2_osm_plan: n_pad must be a positive multiple of 256 covering the table");
  PV2_REQUIRE(workspace_bytes >= pv2_osm_plan_workspace_bytes(n_cols), "pv2_osm_plan: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)n_cols;
  const size_t piece = ((n > 0 ? n : 1) * 4 + 255) / 256 * 256;
  char* ws = static_cast<char*>(workspace);
  uint32_t* keys = reinterpret_cast<uint32_t*>(ws);
  uint32_t* keys_sorted = reinterpret_cast<uint32_t*>(ws + piece);
  int32_t* vals = reinterpret_cast<int32_t*>(ws + 2 * piece);
  void* scratch = ws + 3 * piece;
  if (n_cols > 0) {
    hipLaunchKernelGGL(osm_keys_kernel, dim3(pv2::grid_for(n_cols, 256)), dim3(256), 0, s, tbl, K, n_cols,
                       stride, n_cols_dev, keys, vals);
    size_t need = 0;
    (void)rocprim::radix_sort_pairs(nullptr, need, keys, keys_sorted, vals, perm, n);
    hipError_t e = rocprim::radix_sort_pairs(scratch, need, keys, keys_sorted, vals, perm, n, 0, 32, s);
    if (e != hipSuccess) return pv2::hip_status(e);
  }
  hipLaunchKernelGGL(osm_table_kernel, dim3((unsigned)(n_pad / 256)), dim3(256), 0, s, tbl, K, n_cols, stride,
                     n_cols_dev, keys_sorted, perm, n_pad, tblp, tmask);
  return pv2::check_launch("osm_plan");
}

int pv2_spconv_osm(const float* in_feat, int c_in, const float* weight, int K, int c_out,
                   int weight_reduction_major, const pv2_osm_plan_t* plan, int64_t n_out,
                   const float* zero_row, const float* addend, float* out, float* bn_partial,
                   int* bn_blocks, int* bn_rows_per_block, pv2_stream_t stream) {
  return pv2::spconv_osm(weight_reduction_major != 0, in_feat, c_in, weight, K, c_out, plan, n_out,
                         zero_row, addend, out, bn_partial, bn_blocks, bn_rows_per_block,
                         (hipStream_t)stream);
}

}  // extern "C"
