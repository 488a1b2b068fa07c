// Product-row sparse convolution on gfx950: the atomic-free, bitwise reproducible fp32 path.
//
// Stands in for spconv 2.x's indice_conv / indice_conv_backward behind SubMConv3d / SparseConv3d /
// SparseInverseConv3d (ponder/models/sparse_unet/spconv_unet_v1m1_base.py:41,47,58,112,135,171) and,
// as `pv2_convbn_*`, for the conv -> BatchNorm1d -> (+shortcut) -> ReLU units those layers form
// (:70-83 BasicBlock.forward, :108,120-121 the SparseSequential stages).
//
// Why two stages.  The pair-major scatter-add kernel (sparse_conv.hip) keeps the matrix core fed -
// pairs of one offset are compacted, one LDS weight slab serves 128 of them - but its epilogue adds
// every product row into the output with device-scope fp32 atomics, which MI355X executes at the
// memory side at ~0.9 TB/s of payload: on the ScanNet-shaped bench geometry every C >= 64 layer's
// time was its atomic bytes (P * c_out * 4) / 0.9 TB/s, the MFMA work hidden underneath
// (profiles/r02_spconv_kernel_table_v10.txt).  An output-stationary kernel needs no atomics but
// wastes the fp32 matrix core on absent neighbours (3-5 of 27 offsets are present per voxel here)
// and re-streams the weights per 32-row tile.  So:
//
//   stage 1  spconv_fwd_lds_kernel<store = 2>: prod[p, :] = W[k(p)] . in[pair_in[p], :] - the same
//            compacted pair-major GEMM, one PRODUCT ROW per pair written with plain coalesced
//            stores (the buffer lives in the 256 MiB Infinity Cache between the stages);
//   stage 2  row_reduce_kernel: out[o, :] = sum over the offsets k present at o, in ascending k, of
//            prod[pos[k][o], :] - a gather-sum that streams, with the BatchNorm statistics
//            (per-block partial column sums, rownorm.hip's format) accumulated in its epilogue.
//
// `pos` is the position table of the rulebook: pos[k * stride + o] = index p of the pair of offset k
// whose output row is o, or -1 (one more [K, N] int32 table next to the neighbour table, built by
// pair_positions_kernel).  Every output element is written exactly once in a fixed summation
// order: no zero-fill, no atomics, identical bits on every run - in the forward pass, the
// grad-input pass (same two stages over the transposed pair roles, weights read in place) and the
// weight gradient (partial slabs + ordered reduction, sparse_conv.hip).
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int kMaxK = 32;       // offsets per rulebook on this path (27 submanifold, 8 strided)

__global__ void fill_i32_kernel(int32_t* p, int64_t n, int32_t v) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

// pos_a[k * stride_a + pair_a[p]] = p (and the same for b) for every pair p < kstart[K]; a / b may
// be null.  The tables were filled with -1.  Every (k, row) occurs at most once in a conv rulebook.
__global__ void pair_positions_kernel(const int32_t* __restrict__ pair_a,
                                      const int32_t* __restrict__ pair_b,
                                      const int32_t* __restrict__ kstart, int K, int64_t stride_a,
                                      int64_t stride_b, int32_t* __restrict__ pos_a,
                                      int32_t* __restrict__ pos_b) {
  const int64_t P = kstart[K];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += stride) {
    int lo = 0, hi = K;  // invariant: kstart[lo] <= p < kstart[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (kstart[mid] <= p) lo = mid; else hi = mid;
    }
    if (pos_a) pos_a[(int64_t)lo * stride_a + pair_a[p]] = (int32_t)p;
    if (pos_b) pos_b[(int64_t)lo * stride_b + pair_b[p]] = (int32_t)p;
  }
}

__device__ __forceinline__ void accd(double (&a)[4], const float4& b) {
  a[0] += (double)b.x;
  a[1] += (double)b.y;
  a[2] += (double)b.z;
  a[3] += (double)b.w;
}

__device__ __forceinline__ void add4(float4& a, const float4& b) {
  a.x += b.x;
  a.y += b.y;
  a.z += b.z;
  a.w += b.w;
}

// Stage two.  A team of TS lanes sums one output row: out[o] = (addend[o] + bias) + sum over the
// offsets k present at o, ascending, of prod[pos[k][o]].  No shared memory and no barrier on the way:
// lane l of the team loads the position-table entries k = l, l + TS, ... of its row (one strided
// load each), a ballot turns them into the team's bit mask of present offsets, and the team walks
// the set bits in ascending order - the pair index comes from the owning lane by a shuffle, up to
// FOUR product rows per output row are in flight, and every team works on TWO output rows at a
// time (its row of this pass and of the next), so that eight 16-byte loads per lane are
// outstanding while the dependent chain table -> product rows -> store is only three round trips
// long.  (The first version staged the table through LDS and compacted it with one thread per row
// behind two barriers: 18-27 us per launch regardless of size, all of it exposed latency.)
//
// STATS: the BatchNorm forward statistics of the result in the same pass - per-block partial column
// sums of (y - s) and (y - s)^2 with the shift s = y[0, :] (every team recomputes row 0 with the
// same instruction sequence, hence the same bits - as the second row of its first pass), written
// as partial[block][0..2c) exactly as col_partials_kernel<0> (rownorm.hip) does, for
// col_combine_kernel<0> to finish.
template <int TS>
struct RowState {
  static constexpr int NE = TS >= 32 ? 1 : 32 / TS;   // table entries per lane (K <= 32)
  int e[NE];
  uint32_t mask;
};

template <int TS>
__device__ __forceinline__ void load_entries(RowState<TS>& st, const int32_t* __restrict__ pos,
                                             int64_t pos_stride, int K, int64_t row, bool valid,
                                             int l, int team_base) {
#pragma unroll
  for (int j = 0; j < RowState<TS>::NE; ++j) {
    const int k = l + j * TS;
    st.e[j] = (valid && k < K) ? pos[(int64_t)k * pos_stride + row] : -1;
  }
  uint32_t m = 0;
  if constexpr (TS >= 32) {
    m = (uint32_t)(__ballot(st.e[0] >= 0) >> team_base);   // the team's 32 (or 64) lanes: bit k = offset k
  } else {
#pragma unroll
    for (int j = 0; j < RowState<TS>::NE; ++j) {
      const uint32_t part = (uint32_t)(__ballot(st.e[j] >= 0) >> team_base) & ((1u << TS) - 1u);
      m |= part << (j * TS);
    }
  }
  st.mask = m;
}

// pair index of offset k of this row, from the lane that holds it (k is uniform within the team)
template <int TS>
__device__ __forceinline__ int entry_of(const RowState<TS>& st, int k) {
  if constexpr (RowState<TS>::NE == 1) {
    return __shfl(st.e[0], k, TS);
  } else {
    int p = -1;
#pragma unroll
    for (int j = 0; j < RowState<TS>::NE; ++j) {
      const int v = __shfl(st.e[j], k & (TS - 1), TS);
      if ((k / TS) == j) p = v;
    }
    return p;
  }
}

// next (up to) U pair indices of the row, in ascending offset order; -1 past the end
template <int TS, int U>
__device__ __forceinline__ void next_entries(RowState<TS>& st, int (&p)[U]) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (st.mask) {
      const int k = __builtin_ctz(st.mask);
      st.mask &= st.mask - 1;
      p[u] = entry_of<TS>(st, k);
    } else {
      p[u] = -1;
    }
  }
}

// STATS == 2: the BACKWARD sums of the BatchNorm that PRODUCED the activation whose gradient this launch
// completes (out = its gradient, all consumers added): per-block partial sums of g and g * xhat with
// g = out * (act > 0), xhat = (yprod - mean) * invstd - what col_partials_kernel<1> computes in a pass of
// its own over three matrices.  `sy` = the producer's conv output, `so` = its activation (null: no ReLU),
// `smi` = its {mean, invstd}.
#ifndef PV2_REDUCE_AHEAD
#define PV2_REDUCE_AHEAD 4
#endif
constexpr int kReduceAhead = PV2_REDUCE_AHEAD;
template <int TS, int NJ, int STATS>
__global__ __launch_bounds__(256) void row_reduce_kernel(
    const float* __restrict__ T, const int32_t* __restrict__ pos, int64_t pos_stride, int K, int c,
    int64_t n_rows, int64_t rows_per_block, const float* __restrict__ bias,
    const float* addend, float* Y, float* partial, const float* __restrict__ sy,
    const float* __restrict__ so, const float* __restrict__ smi) {
  constexpr int NT = 256 / TS;
  __shared__ double s_red[STATS ? 2048 * NJ : 1];
  const int tid = threadIdx.x, team = tid / TS, l = tid % TS;
  const int team_base = (tid & 63) / TS * TS;
  const int c4n = c >> 2;
  const float4* T4 = reinterpret_cast<const float4*>(T);
  const float4* bias4 = reinterpret_cast<const float4*>(bias);
  const float4* add4p = reinterpret_cast<const float4*>(addend);
  const float4* sy4 = reinterpret_cast<const float4*>(sy);
  const float4* so4 = reinterpret_cast<const float4*>(so);
  float4* Y4 = reinterpret_cast<float4*>(Y);
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r_end = min(n_rows, r_begin + rows_per_block);
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);

  // (the statistics accumulate in DOUBLE per thread and per block - a block's rows are summed exactly to
  // within the final rounding of its fp32 partial row; round 6)
  float4 sh[NJ], is4[NJ];
  double a0[NJ][4], a1[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    sh[j] = is4[j] = zero;
#pragma unroll
    for (int q = 0; q < 4; ++q) a0[j][q] = a1[j][q] = 0.0;
  }
  if (STATS == 2) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = l + j * TS;
      if (col < c4n) {
        sh[j] = reinterpret_cast<const float4*>(smi)[col];          // mean
        is4[j] = reinterpret_cast<const float4*>(smi + c)[col];     // invstd
      }
    }
  }

  // rows come in pairs (A, B); the very first B of a STATS == 1 launch is row 0 (the shift)
  bool need_shift = STATS == 1;
  for (int64_t ra = r_begin + team; ra < r_end || need_shift; ra += 2 * NT) {
    const int64_t rb = need_shift ? 0 : ra + NT;
    const bool va = ra < r_end, vb = need_shift || rb < r_end;
    RowState<TS> sa, sb;
    load_entries<TS>(sa, pos, pos_stride, K, va ? ra : 0, va, l, team_base);
    load_entries<TS>(sb, pos, pos_stride, K, vb ? rb : 0, vb, l, team_base);
    float4 accA[NJ], accB[NJ];
    float4 yA[NJ], yB[NJ], oA[NJ], oB[NJ];   // STATS == 2: the producer's rows, requested up front
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = l + j * TS;
      float4 ia = zero, ib = zero;
      yA[j] = yB[j] = zero;
      oA[j] = oB[j] = make_float4(1.f, 1.f, 1.f, 1.f);
      if (col < c4n) {
        if (add4p) {
          if (va) ia = add4p[ra * c4n + col];
          if (vb) ib = add4p[rb * c4n + col];
        }
        if (bias4) {
          const float4 bv = bias4[col];
          add4(ia, bv);
          add4(ib, bv);
        }
        if (STATS == 2) {
          if (va) yA[j] = sy4[ra * c4n + col];
          if (vb) yB[j] = sy4[rb * c4n + col];
          if (so4) {
            if (va) oA[j] = so4[ra * c4n + col];
            if (vb) oB[j] = so4[rb * c4n + col];
          }
        }
      }
      accA[j] = ia;
      accB[j] = ib;
    }
    // U product rows per row and trip, all requested before the first is added.  (Round 6: U = 8 - most rows
    // done in ONE trip, the longest in four instead of seven - measured 20.8 against 20.0 us per launch over
    // the step's 116 reduces, 18.5 against 18.3 ms per step: the kernel is not bound by its trips.  4 stays.)
    constexpr int U = NJ == 1 ? kReduceAhead : 4;
    while (sa.mask | sb.mask) {   // (uniform within the team; teams of a wave diverge)
      int pa[U], pb[U];
      next_entries<TS, U>(sa, pa);
      next_entries<TS, U>(sb, pb);
      float4 vA[U][NJ], vB[U][NJ];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int col = l + j * TS;
          vA[u][j] = (pa[u] >= 0 && col < c4n) ? T4[(int64_t)pa[u] * c4n + col] : zero;
          vB[u][j] = (pb[u] >= 0 && col < c4n) ? T4[(int64_t)pb[u] * c4n + col] : zero;
        }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (pa[u] >= 0) add4(accA[j], vA[u][j]);   // (adding +0 would turn a -0 sum into +0)
          if (pb[u] >= 0) add4(accB[j], vB[u][j]);
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = l + j * TS;
      if (col >= c4n) continue;
      if (need_shift) sh[j] = accB[j];                 // row 0: the shift (block 0 also stores it below)
      if (va) Y4[ra * c4n + col] = accA[j];
      if (!need_shift && vb) Y4[rb * c4n + col] = accB[j];
      if (STATS == 1) {
        if (va) {
          const float4 d = make_float4(accA[j].x - sh[j].x, accA[j].y - sh[j].y, accA[j].z - sh[j].z,
                                       accA[j].w - sh[j].w);
          accd(a0[j], d);
          accd(a1[j], make_float4(d.x * d.x, d.y * d.y, d.z * d.z, d.w * d.w));
        }
        if (!need_shift && vb) {
          const float4 d = make_float4(accB[j].x - sh[j].x, accB[j].y - sh[j].y, accB[j].z - sh[j].z,
                                       accB[j].w - sh[j].w);
          accd(a0[j], d);
          accd(a1[j], make_float4(d.x * d.x, d.y * d.y, d.z * d.z, d.w * d.w));
        }
      }
      if (STATS == 2) {
        auto one = [&](const float4& g_, const float4& yv, const float4& ov) {
          const float4 g = make_float4(ov.x > 0.f ? g_.x : 0.f, ov.y > 0.f ? g_.y : 0.f,
                                       ov.z > 0.f ? g_.z : 0.f, ov.w > 0.f ? g_.w : 0.f);
          accd(a0[j], g);
          accd(a1[j], make_float4(g.x * (yv.x - sh[j].x) * is4[j].x, g.y * (yv.y - sh[j].y) * is4[j].y,
                                  g.z * (yv.z - sh[j].z) * is4[j].z, g.w * (yv.w - sh[j].w) * is4[j].w));
        };
        if (va) one(accA[j], yA[j], oA[j]);
        if (vb) one(accB[j], yB[j], oB[j]);
      }
    }
    if (need_shift) {
      need_shift = false;
      ra -= NT;   // the B slot of this pass went to row 0: the next pass starts one team-stride on
    }
  }

  if (STATS) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = l + j * TS;
      if (col < c4n) {
        double* r0 = &s_red[team * 2 * c + 4 * col];
        r0[0] = a0[j][0]; r0[1] = a0[j][1]; r0[2] = a0[j][2]; r0[3] = a0[j][3];
        double* r1 = r0 + c;
        r1[0] = a1[j][0]; r1[1] = a1[j][1]; r1[2] = a1[j][2]; r1[3] = a1[j][3];
      }
    }
    __syncthreads();
    for (int t = tid; t < 2 * c; t += 256) {
      double s = 0.0;
      for (int q = 0; q < NT; ++q) s += s_red[q * 2 * c + t];  // teams in a fixed order
      partial[(int64_t)blockIdx.x * 2 * c + t] = (float)s;
    }
  }
}

// What the STATS modes of row_reduce_kernel need beyond the sum itself.
struct ReduceStats {
  int mode = 0;                   // 0 / 1 (forward statistics of the result) / 2 (backward sums, see above)
  float* partial = nullptr;       // [blocks][2 c]
  const float* sy = nullptr;      // mode 2
  const float* so = nullptr;
  const float* smi = nullptr;
};

template <int STATS>
void launch_reduce(const float* T, const int32_t* pos, int64_t pos_stride, int K, int c,
                   int64_t n_rows, int blocks, int64_t rpb, const float* bias, const float* addend,
                   float* Y, const ReduceStats& st, hipStream_t s) {
  const int c4n = c / 4;
#define PV2_RED(TS, NJ)                                                                          \
  hipLaunchKernelGGL((row_reduce_kernel<TS, NJ, STATS>), dim3(blocks), dim3(256), 0, s, T, pos,  \
                     pos_stride, K, c, n_rows, rpb, bias, addend, Y, st.partial, st.sy, st.so,   \
                     st.smi)
  if (c4n <= 8) PV2_RED(8, 1);
  else if (c4n <= 16) PV2_RED(16, 1);
  else if (c4n <= 32) PV2_RED(32, 1);
  else if (c4n <= 64) PV2_RED(64, 1);
  else PV2_RED(64, 2);
#undef PV2_RED
}

int reduce_rows_stats(const float* prod, const int32_t* pos, int64_t pos_stride, int K, int c,
                      int64_t n_rows, const float* bias, const float* addend, float* out,
                      const ReduceStats& st, int* bn_blocks, hipStream_t s) {
  PV2_REQUIRE(K >= 1 && K <= kMaxK, "pv2_spconv_reduce_rows: 1 <= K <= 32");
  PV2_REQUIRE(c >= 4 && (c % 4) == 0 && c <= 512,
              "pv2_spconv_reduce_rows: channel count must be a multiple of 4, at most 512");
  PV2_REQUIRE(n_rows >= 0 && pos_stride >= n_rows, "pv2_spconv_reduce_rows: bad row count");
  if (bn_blocks) *bn_blocks = 0;
  if (n_rows == 0) return PV2_OK;
  // Two rows per team and pass; as many workgroups as that takes, up to 2048 when the BatchNorm
  // statistics ride along (one partial row per workgroup, pv2_bn_workspace_floats) and 16384 otherwise.
  const int c4n = c / 4;
  const int nt = 256 / (c4n <= 8 ? 8 : c4n <= 16 ? 16 : c4n <= 32 ? 32 : 64);
  const int64_t cap = st.mode ? 2048 : 16384;
  int64_t rpb = (n_rows + cap - 1) / cap;
  rpb = (rpb + 2 * nt - 1) / (2 * nt) * (2 * nt);
  const int blocks = (int)((n_rows + rpb - 1) / rpb);
  if (st.mode == 1)
    launch_reduce<1>(prod, pos, pos_stride, K, c, n_rows, blocks, rpb, bias, addend, out, st, s);
  else if (st.mode == 2)
    launch_reduce<2>(prod, pos, pos_stride, K, c, n_rows, blocks, rpb, bias, addend, out, st, s);
  else
    launch_reduce<0>(prod, pos, pos_stride, K, c, n_rows, blocks, rpb, bias, addend, out, st, s);
  if (st.mode && bn_blocks) *bn_blocks = blocks;
  return pv2::check_launch("spconv_reduce_rows");
}

int reduce_rows(const float* prod, const int32_t* pos, int64_t pos_stride, int K, int c,
                int64_t n_rows, const float* bias, const float* addend, float* out,
                float* bn_partial, int* bn_blocks, hipStream_t s) {
  ReduceStats st;
  if (bn_partial) {
    st.mode = 1;
    st.partial = bn_partial;
  }
  return reduce_rows_stats(prod, pos, pos_stride, K, c, n_rows, bias, addend, out, st, bn_blocks, s);
}

// Events that order the side stream of pv2_convbn_backward behind the caller's stream.  A wait
// binds to the record that precedes it, so a small ring of reusable events is enough.
hipEvent_t fork_event() {
  constexpr int kRing = 64;
  static hipEvent_t ring[kRing];
  static int next = -1;
  if (next < 0) {
    for (int i = 0; i < kRing; ++i) (void)hipEventCreateWithFlags(&ring[i], hipEventDisableTiming);
    next = 0;
  }
  hipEvent_t e = ring[next];
  next = (next + 1) % kRing;
  return e;
}

}  // namespace

namespace pv2 {

// an event of the fork ring for other translation units (spunet_exec.hip's 16-bit units)
int fork_event_for(hipEvent_t* ev) {
  *ev = fork_event();
  return *ev ? PV2_OK : PV2_E_WORKSPACE;
}

// Backward of one conv + BatchNorm unit (the body of pv2_convbn_backward).  dx_accumulate: the
// grad-input is ADDED to what dx already holds (another consumer of the same activation wrote its
// gradient first) - the row-reduce kernel takes dx as its addend, element for element in place.
// bn_sums_ready: gsum already holds this unit's {sum g, sum g * xhat} (the launch that completed
// grad_out computed them in its epilogue): only the elementwise half of the BatchNorm backward runs.
// dx_producer: this unit's grad-input COMPLETES the gradient of its input activation, which is the
// output of the conv + BatchNorm unit described there - its backward sums are taken in the row
// reduce's epilogue and combined into dx_producer->gsum (*dx_sums_done = 1 when that happened).
int convbn_backward(const pv2_conv_geom* g, const float* grad_out, const float* x, int c_in,
                    const float* weight, int c_out, const float* y_conv, const float* out_or_null,
                    const float* mean_invstd, const float* bn_weight, float* prod_ws,
                    float* stats_ws, float* gsum, float* dy, float* dres_or_null, float* dx_or_null,
                    int dx_accumulate, float* dweight_or_null, float* part_ws, hipStream_t s,
                    hipStream_t side, int bn_sums_ready, const BnProducer* dx_producer,
                    int* dx_sums_done) {
  PV2_REQUIRE(g != nullptr && g->n_out >= 2, "pv2_convbn_backward: needs at least two output rows");
  if (dx_sums_done) *dx_sums_done = 0;
  if (bn_sums_ready) {
    if (int e = pv2::bn_backward_apply(grad_out, y_conv, out_or_null, mean_invstd, bn_weight, gsum,
                                       g->n_out, c_out, dy, dres_or_null, s))
      return e;
  } else if (int e = pv2_bn_backward(grad_out, y_conv, out_or_null, mean_invstd, bn_weight, g->n_out,
                                     c_out, stats_ws, gsum, dy, dres_or_null, (pv2_stream_t)s)) {
    return e;
  }
  // The weight gradient feeds nothing until the optimizer: off the critical chain, on the side stream.  WHERE it
  // forks decides what it shares the machine with: right behind the BatchNorm backward (PV2_WGRAD_LATE=0, rounds
  // 4 - 5) it runs beside this unit's grad-input products - two MFMA kernels slowing each other -; behind the
  // products launch (1, the default since round 6) it starts when the products end, beside the row reduce and
  // the next unit's BatchNorm kernels, which leave the matrix pipe idle: 18.14 -> 17.88 ms per step; behind the
  // row reduce (2) it meets the NEXT unit's products: 18.25.
  static const int wgrad_late = [] {
    const char* e = getenv("PV2_WGRAD_LATE");
    return e ? atoi(e) : 1;
  }();
  const bool osm_dx = dx_or_null && pv2::use_osm(&g->osm_bwd, g->zero_row, g->K, g->n_in, g->n_out, c_out, c_in);
  const bool late = wgrad_late && dweight_or_null && side != s && dx_or_null && !osm_dx;
  auto weight_gradient = [&]() -> int {
    if (side != s) {
      hipEvent_t ev = fork_event();
      if (int e = pv2::hip_status(hipEventRecord(ev, s))) return e;
      if (int e = pv2::hip_status(hipStreamWaitEvent(side, ev, 0))) return e;
    }
    return pv2::spconv_wgrad(x, g->n_in, c_in, dy, g->n_out, c_out, g->K, g->pair_in, g->pair_out,
                             g->kstart, g->tile_start_w, g->tile_pairs_w, g->n_tiles_w, dweight_or_null,
                             part_ws, side);
  };
  if (dweight_or_null && !late)
    if (int e = weight_gradient()) return e;
  if (osm_dx) {
    // grad-input, output-stationary over the INPUT rows (sparse_conv_osm.hip): one launch, the other
    // consumers' gradient added in its epilogue, forward weight read in place
    if (int e = pv2::spconv_osm(true, dy, c_out, weight, g->K, c_in, &g->osm_bwd, g->n_in, g->zero_row,
                                dx_accumulate ? dx_or_null : nullptr, dx_or_null, nullptr, nullptr,
                                nullptr, s))
      return e;
  } else if (dx_or_null) {
    // grad-input: the same two stages with the pair roles swapped, forward weight read in place
    if (int e = pv2::spconv_products(true, dy, c_out, weight, g->K, c_in, g->pair_out, g->kstart,
                                     g->tile_start, g->n_tiles, prod_ws, s))
      return e;
    if (late && wgrad_late == 1)
      if (int e = weight_gradient()) return e;
    ReduceStats st;
    if (dx_producer && dx_producer->gsum && g->n_in >= 2) {
      st.mode = 2;
      st.partial = stats_ws;
      st.sy = dx_producer->y_conv;
      st.so = dx_producer->out_or_null;
      st.smi = dx_producer->mean_invstd;
    }
    int blocks = 0;
    if (int e = reduce_rows_stats(prod_ws, g->pos_in, g->pos_in_stride, g->K, c_in, g->n_in, nullptr,
                                  dx_accumulate ? dx_or_null : nullptr, dx_or_null, st, &blocks, s))
      return e;
    if (late && wgrad_late != 1)   // (2: behind the row reduce)
      if (int e = weight_gradient()) return e;
    if (st.mode == 2) {
      // (combined at once: the workspace serves the next unit's statistics)
      if (int e = pv2::bn_backward_combine(stats_ws, blocks, c_in, dx_producer->gsum, s)) return e;
      if (dx_sums_done) *dx_sums_done = 1;
    }
  }
  return PV2_OK;
}

}  // namespace pv2

extern "C" {

int pv2_pair_positions(const int32_t* pair_out, const int32_t* pair_in, const int32_t* kstart, int K,
                       int64_t n_pairs_bound, int64_t out_stride, int64_t in_stride,
                       int32_t* pos_out, int32_t* pos_in, pv2_stream_t stream) {
  PV2_REQUIRE(K >= 1 && n_pairs_bound >= 0 && out_stride >= 0 && in_stride >= 0,
              "pv2_pair_positions: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  if (pos_out && out_stride > 0)
    hipLaunchKernelGGL(fill_i32_kernel, dim3(pv2::grid_for(K * out_stride, 256)), dim3(256), 0, s,
                       pos_out, (int64_t)K * out_stride, -1);
  if (pos_in && in_stride > 0)
    hipLaunchKernelGGL(fill_i32_kernel, dim3(pv2::grid_for(K * in_stride, 256)), dim3(256), 0, s,
                       pos_in, (int64_t)K * in_stride, -1);
  if (n_pairs_bound > 0 && (pos_out || pos_in))
    hipLaunchKernelGGL(pair_positions_kernel, dim3(pv2::grid_for(n_pairs_bound, 256)), dim3(256), 0,
                       s, pair_out, pair_in, kstart, K, out_stride, in_stride, pos_out, pos_in);
  return pv2::check_launch("pair_positions");
}

int pv2_spconv_products(const float* in_feat, int c_in, const float* weight, int K, int c_out,
                        int weight_reduction_major, const int32_t* pair_in, const int32_t* kstart,
                        const int32_t* tile_start, int64_t n_tiles, float* prod,
                        pv2_stream_t stream) {
  return pv2::spconv_products(weight_reduction_major != 0, in_feat, c_in, weight, K, c_out, pair_in,
                              kstart, tile_start, n_tiles, prod, (hipStream_t)stream);
}

int pv2_spconv_reduce_rows(const float* prod, const int32_t* pos, int64_t pos_stride, int K, int c,
                           int64_t n_rows, const float* bias, const float* addend, float* out,
                           float* bn_partial, int* bn_blocks, pv2_stream_t stream) {
  return reduce_rows(prod, pos, pos_stride, K, c, n_rows, bias, addend, out, bn_partial, bn_blocks,
                     (hipStream_t)stream);
}

int pv2_convbn_forward(const pv2_conv_geom* g, const float* x, int c_in, const float* weight,
                       int c_out, const float* bn_weight, const float* bn_bias,
                       const float* residual, int relu, float eps, float momentum,
                       float* running_mean, float* running_var, float* prod_ws, float* stats_ws,
                       float* y_conv, float* mean_invstd, float* out, pv2_stream_t stream) {
  PV2_REQUIRE(g != nullptr && g->n_out >= 2, "pv2_convbn_forward: needs at least two output rows");
  hipStream_t s = (hipStream_t)stream;
  if (pv2::use_osm(&g->osm_fwd, g->zero_row, g->K, g->n_out, g->n_in, c_in, c_out)) {
    // one launch: output-stationary conv with the block statistics in its epilogue
    int blocks = 0, rpb = 0;
    if (int e = pv2::spconv_osm(false, x, c_in, weight, g->K, c_out, &g->osm_fwd, g->n_out, g->zero_row,
                                nullptr, y_conv, stats_ws, &blocks, &rpb, s))
      return e;
    return pv2::bn_forward_from_block_stats(y_conv, g->n_out, c_out, stats_ws, blocks, rpb, bn_weight,
                                            bn_bias, residual, relu, eps, momentum, running_mean,
                                            running_var, mean_invstd, out, s);
  }
  if (int e = pv2::spconv_products(false, x, c_in, weight, g->K, c_out, g->pair_in, g->kstart,
                                   g->tile_start, g->n_tiles, prod_ws, s))
    return e;
  int blocks = 0;
  if (int e = reduce_rows(prod_ws, g->pos_out, g->pos_out_stride, g->K, c_out, g->n_out, nullptr,
                          nullptr, y_conv, stats_ws, &blocks, s))
    return e;
  return pv2::bn_forward_from_partials(y_conv, g->n_out, c_out, stats_ws, blocks, bn_weight, bn_bias,
                                       residual, relu, eps, momentum, running_mean, running_var,
                                       mean_invstd, out, s);
}

int pv2_convbn_backward(const pv2_conv_geom* g, const float* grad_out, const float* x, int c_in,
                        const float* weight, int c_out, const float* y_conv,
                        const float* out_or_null, const float* mean_invstd, const float* bn_weight,
                        float* prod_ws, float* stats_ws, float* gsum, float* dy,
                        float* dres_or_null, float* dx_or_null, float* dweight_or_null,
                        float* part_ws, pv2_stream_t stream, pv2_stream_t side_stream) {
  return pv2::convbn_backward(g, grad_out, x, c_in, weight, c_out, y_conv, out_or_null, mean_invstd,
                              bn_weight, prod_ws, stats_ws, gsum, dy, dres_or_null, dx_or_null, 0,
                              dweight_or_null, part_ws, (hipStream_t)stream,
                              side_stream ? (hipStream_t)side_stream : (hipStream_t)stream, 0, nullptr,
                              nullptr);
}

}  // extern "C"
