/*
 * ponderv2_hip.h -- C ABI of libponderv2_hip.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary of the PonderV2 pre-training hot path.  The reference has no C
 * ABI of its own: its native layer is pybind11-on-torch::Tensor (libs/smooth-sampler) plus two
 * out-of-tree pip wheels (spconv 2.x, torch_scatter).  Every entry point below names the
 * reference interface it stands in for (paths relative to the reference checkout).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless it says "host";
 *   - the caller owns all memory (PyTorch's caching allocator on the Python side);
 *   - work is enqueued on `stream` (a hipStream_t passed as void*); nothing synchronises;
 *   - return value: 0 on success, otherwise a hipError_t (launch error) or a negative PV2_E_*
 *     argument error; pv2_last_error() gives a static description for the calling thread;
 *   - float tensors are fp32 unless the function name ends in _f64;
 *   - feature matrices are row-major [rows, channels].
 */
#ifndef PONDERV2_HIP_H
#define PONDERV2_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pv2_stream_t; /* hipStream_t */

#define PV2_OK 0
#define PV2_E_BADARG (-1)
#define PV2_E_UNSUPPORTED (-2)
#define PV2_E_WORKSPACE (-3)

/* element type codes of the entry points that take 16-bit feature matrices (void* arguments) */
#define PV2_F32 0
#define PV2_BF16 1
#define PV2_F16 2

int pv2_abi_version(void);
/* Zero `nbytes` (multiple of 4) with a KERNEL on `stream`.  Scratch and accumulate-into buffers of
 * this library are cleared with a kernel rather than hipMemsetAsync: on ROCm 7.2 a hipMemsetAsync
 * issued from this library while the stream was being captured into a hipGraph was not replayed
 * with the graph (stale bias gradients in the render head until it was replaced). */
int pv2_zero_fill(void* ptr, int64_t nbytes, pv2_stream_t stream);
int pv2_debug_set_os16_variant(int v); /* tuning knob of pv2_spconv16_os_forward (tools/bench_spconv16.py) */
const char* pv2_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Rulebook ("indice pair") generation.
 * Replaces spconv 2.x's hash table + generate_subm_conv_inds / generate_conv_inds, reached in
 * the reference through spconv.SubMConv3d / SparseConv3d / SparseInverseConv3d at
 * ponder/models/sparse_unet/spconv_unet_v1m1_base.py:41,47,58,112,135,171.
 *
 * coords      int32 [n,4] rows (b,x,y,z), 0 <= b < 65536, 0 <= x,y,z < 65504.  Rows with b < 0 are
 *             PADDING (a capacity-sized array whose valid count lives on the device, e.g. the
 *             out_coords of pv2_downsample_unique past *n_out): they enter no hash, have no
 *             neighbours and produce no output voxel - so a chain of levels can be built without
 *             reading any count back to the host.
 * Kernel offset index k = ((ix*K)+iy)*K+iz over the K^3 window, spatial dims in the order they
 * appear in `coords` (first spatial dim slowest), matching a dense conv3d weight [.,kx,ky,kz,.].
 *
 * Canonical order (what "bit-exact rulebook" means in this project, mirrored by oracle/):
 *   pairs are grouped by k ascending and, inside a group, sorted by output row ascending;
 *   strided-conv output voxels are sorted by (b,x,y,z) lexicographically.
 * ------------------------------------------------------------------------------------------ */

/* Build an open-addressing hash  (b,x,y,z) -> row.  table_size must be a power of two and
 * >= 2*n.  Duplicate coordinates keep the smallest row. */
int pv2_hash_build(const int32_t* coords, int64_t n, uint64_t* table_keys, int32_t* table_vals,
                   int64_t table_size, pv2_stream_t stream);

/* Submanifold neighbour table: nbr[k*n + i] = row of voxel coords[i] + (offset k - K/2), or -1.
 * ksize is odd (1,3,5,...).  nbr has ksize^3 * n entries. */
int pv2_subm_neighbor_table(const int32_t* coords, int64_t n, int ksize,
                            const uint64_t* table_keys, const int32_t* table_vals,
                            int64_t table_size, int32_t* nbr, pv2_stream_t stream);

/* Strided (kernel == stride, no padding) sparse conv, step 1: unique output voxels.
 * keys_tmp, keys_sorted: uint64 [n]; out_coords: int32 [n,4] (first *n_out rows valid);
 * n_out: int32 [1] (device).  out_shape: host int32[3], outputs with a coordinate >= out_shape
 * are dropped like spconv does.  workspace: device scratch of pv2_downsample_workspace_bytes(n). */
size_t pv2_downsample_workspace_bytes(int64_t n);
int pv2_downsample_unique(const int32_t* coords, int64_t n, int stride, const int32_t* out_shape,
                          uint64_t* keys_tmp, uint64_t* keys_sorted, int32_t* out_coords,
                          int32_t* n_out, void* workspace, size_t workspace_bytes,
                          pv2_stream_t stream);

/* Step 2: table tbl[k*n_cap + o] = input row that maps to output row o through offset k, or -1
 * (k = ((x%s)*s + y%s)*s + z%s).  n_cap is the row stride of tbl (>= *n_out; normally n). */
int pv2_downsample_table(const int32_t* coords, int64_t n, int stride, const int32_t* out_shape,
                         const uint64_t* keys_sorted, const int32_t* n_out, int32_t* tbl,
                         int64_t n_cap, pv2_stream_t stream);

/* Ordered compaction of a [K, n] table (entries >= 0 are valid) into pair lists.
 * Phase 1 writes kstart[K+1] (exclusive prefix of per-k counts; kstart[K] = total pairs) and
 * block_sums (int32 [K * ceil(n/PV2_SCAN_CHUNK)]).  Phase 2 (after the caller has sized the
 * outputs from kstart[K]) writes pair_other[p] = table value, pair_row[p] = column index,
 * ordered by (k, column).  `n_rows_dev`, when not NULL, is a device int32 holding the number of
 * valid columns (<= n); columns beyond it are ignored. */
#define PV2_SCAN_CHUNK 2048
int pv2_table_count(const int32_t* tbl, int K, int64_t n, const int32_t* n_rows_dev,
                    int32_t* block_sums, int32_t* kstart, pv2_stream_t stream);
int pv2_table_compact(const int32_t* tbl, int K, int64_t n, const int32_t* n_rows_dev,
                      const int32_t* block_sums, int32_t* pair_other, int32_t* pair_row,
                      pv2_stream_t stream);

/* out[s*(K+1) + k] = sum_{j<k} ceil((kstart[j+1]-kstart[j]) / tile_sizes[s]) for up to 4 tile sizes
 * (host array): the tile prefixes (`tile_start`) the conv entry points below take. */
int pv2_tile_prefix(const int32_t* kstart, int K, const int32_t* tile_sizes, int n_sizes,
                    int32_t* out, pv2_stream_t stream);

/* Table helpers for the output-stationary conv below.
 * invert: out[k*n_out_cols + tbl[k*stride_in + j]] = j for valid entries (out is filled with -1
 *   first); n_cols_dev (device int32, may be NULL) caps the columns read.  Every (k, value) pair may
 *   occur at most once (true for the strided-conv tables: a child has one parent per offset).
 * masks: mask[i] = bit k set iff tbl[k*stride_in + i] >= 0 (K <= 63); columns >= *n_cols_dev get
 *   the all-ones mask so they sort last. */
int pv2_table_invert(const int32_t* tbl, int K, int64_t n_cols, int64_t stride_in,
                     const int32_t* n_cols_dev, int32_t* out, int64_t n_out_cols,
                     pv2_stream_t stream);
int pv2_table_masks(const int32_t* tbl, int K, int64_t n_cols, int64_t stride_in,
                    const int32_t* n_cols_dev, int64_t* mask, pv2_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Sparse convolution arithmetic (gather -> MFMA f32 GEMM -> scatter-add, one launch per conv).
 * Replaces spconv 2.x's ops.indice_conv / indice_conv_backward (same call sites as above).
 *
 *   out[pair_out[p], :] += W[k(p)] * in[pair_in[p], :]       for every pair p
 *
 * weight      fp32 [c_out, K, c_in]  (spconv 2.x layout [Cout, kD, kH, kW, Cin], flattened)
 * kstart      int32 [K+1] device prefix of pairs per offset (from pv2_table_count)
 * tile_start  int32 [K+1] device prefix of ceil(count_k / tile_pairs), where
 *             tile_pairs = pv2_spconv_forward_tile(c_in, c_out): 128 for the LDS-staged kernel
 *             (c_in % 32 == 0), PV2_PAIR_TILE for the generic one
 * n_tiles     tile_start[K] (host value)
 * center_tile_lo/hi  tile range [tile_start[kc], tile_start[kc+1]) of an offset kc whose pairs hit
 *             EVERY output row exactly once (the centre tap of a submanifold conv, or the single
 *             offset of a 1x1 conv).  When non-empty (LDS-staged kernel only) that offset runs first
 *             with plain stores, so `out` need NOT be initialised and only the other offsets pay
 *             for atomics.  Pass lo == hi (e.g. 0,0) otherwise; then
 * `out` must be pre-initialised (zeros, or a bias/residual to accumulate onto).
 * ------------------------------------------------------------------------------------------ */
#define PV2_PAIR_TILE 32
int pv2_spconv_forward_tile(int c_in, int c_out);
int pv2_spconv_forward(const float* in_feat, int64_t n_in, int c_in, const float* weight, int K,
                       int c_out, const int32_t* pair_in, const int32_t* pair_out,
                       const int32_t* kstart, const int32_t* tile_start, int tile_pairs,
                       int64_t n_tiles, int64_t center_tile_lo, int64_t center_tile_hi,
                       float* out_feat, int64_t n_out, pv2_stream_t stream);

/* The same scatter-add convolution with the weight tensor given REDUCTION-MAJOR, weight_t[c_in, K,
 * c_out]: what grad-input needs - grad_in = conv(grad_out, W^T) - when W is the forward weight
 * [c_out_fwd = c_in here, K, c_in_fwd = c_out here] as stored, so no transposed copy of the weights
 * is materialised.  LDS-staged kernel only: c_in % 32 == 0, c_out % 4 == 0, tile_pairs = 128;
 * `out` zero-initialised (or pre-loaded). */
int pv2_spconv_forward_wt(const float* in_feat, int64_t n_in, int c_in, const float* weight_t, int K,
                          int c_out, const int32_t* pair_in, const int32_t* pair_out,
                          const int32_t* kstart, const int32_t* tile_start, int tile_pairs,
                          int64_t n_tiles, float* out_feat, int64_t n_out, pv2_stream_t stream);

/* Output-stationary form of the same convolution (same reference call sites): every output row is
 * computed by ONE workgroup from the gather table and written once - no atomics, no zero-fill,
 * bitwise reproducible.
 *   out[o, n] = bias[n] + sum_k sum_c in[nbr[k*nbr_stride + o], c] * weight[n, kw(k), c]
 * with kw(k) = kflip ? K-1-k : k, entries nbr < 0 skipped.  perm (int32 [n_out], may be NULL) is
 * the order in which rows are grouped into 32-row tiles (rows sorted by pv2_table_masks keep the
 * gathered tiles dense); results do not depend on it beyond fp32 summation order per row - which
 * is fixed for a given perm.  bias may be NULL.  Grad-input of a submanifold conv = the same call
 * on grad_out with the transposed weights [c_in, K, c_out] and kflip = 1 (coordinates unique). */
int pv2_spconv_os_forward(const float* in_feat, int64_t n_in, int c_in, const float* weight, int K,
                          int c_out, const int32_t* nbr, int64_t nbr_stride, const int32_t* perm,
                          int kflip, const float* bias, float* out_feat, int64_t n_out,
                          pv2_stream_t stream);

/* grad wrt weight:  dW[n, k, c] += sum_{p in k} dout[pair_out[p], n] * in[pair_in[p], c].
 * dweight must be zero-initialised by the caller.  Here tile_start / n_tiles count chunks of
 * tile_pairs = pv2_spconv_wgrad_tile(c_in, c_out, n_pairs, K) pairs (PV2_WGRAD_TILE or 2048). */
#define PV2_WGRAD_TILE 512
int pv2_spconv_wgrad_tile(int c_in, int c_out, int64_t n_pairs, int K);
int pv2_spconv_backward_weight(const float* in_feat, int64_t n_in, int c_in, const float* dout,
                               int64_t n_out, int c_out, int K, const int32_t* pair_in,
                               const int32_t* pair_out, const int32_t* kstart,
                               const int32_t* tile_start, int tile_pairs, int64_t n_tiles,
                               float* dweight, pv2_stream_t stream);
/* (grad wrt input is pv2_spconv_forward with pair_in/pair_out swapped and weight transposed to
 *  [c_in, K, c_out].) */

/* The same weight gradient in a DETERMINISTIC two-stage form (c_in % 4 == 0, c_out % 4 == 0): every
 * chunk of pairs writes its block of partial sums with plain stores into its own slab of `partial`
 * (pv2_spconv_wgrad_partial_floats(c_in, c_out, n_tiles) floats, uninitialised), then one pass adds
 * the slabs of each offset in slab order and writes EVERY element of dweight (which therefore needs
 * no clearing).  No atomics: bitwise identical results on every run. */
int64_t pv2_spconv_wgrad_partial_floats(int c_in, int c_out, int64_t n_tiles);
int pv2_spconv_backward_weight_det(const float* in_feat, int64_t n_in, int c_in, const float* dout,
                                   int64_t n_out, int c_out, int K, const int32_t* pair_in,
                                   const int32_t* pair_out, const int32_t* kstart,
                                   const int32_t* tile_start, int tile_pairs, int64_t n_tiles,
                                   float* partial, float* dweight, pv2_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Product-row form of the same convolutions (csrc/sparse_conv_pr.hip; same reference call sites:
 * spconv_unet_v1m1_base.py:41,47,58,112,135,171): no atomics, no zero-fill, bitwise reproducible.
 *
 *   stage 1  pv2_spconv_products:   prod[p, :] = W[k(p)] . in[pair_in[p], :]  for every pair p of
 *            the canonical (offset, output row) list - the pair-major MFMA GEMM of
 *            pv2_spconv_forward with one result row per PAIR (plain stores).  c_in % 32 == 0;
 *            tile_start / n_tiles count 128-pair tiles.  weight_reduction_major != 0: the weight is
 *            given as [c_in, K, c_out] (pv2_spconv_forward_wt's layout; grad-input reads the
 *            forward weight in place), c_out % 4 == 0.  prod: n_pairs x c_out floats.
 *   stage 2  pv2_spconv_reduce_rows: out[o, :] = (addend[o, :] + bias) + sum over k ascending of
 *            prod[pos[k*pos_stride + o], :] (entries < 0 skipped); c % 4 == 0, K <= 32.  addend and
 *            bias may be NULL.  bn_partial != NULL: also the BatchNorm forward statistics of `out`
 *            as per-block partial column sums (<= 1024 blocks x 2c floats, the workspace of
 *            pv2_bn_forward; *bn_blocks receives the block count), shifted by row 0.
 *   pv2_pair_positions: the position tables of a rulebook.  pos_out[k*out_stride + pair_out[p]] = p
 *            and pos_in[k*in_stride + pair_in[p]] = p for p < kstart[K] (a device value;
 *            n_pairs_bound >= it sizes the launch), every other entry -1.  Either table may be NULL.
 *            A conv rulebook holds each (offset, row) at most once, so the tables are well defined.
 * Grad-input = stage 1 on grad_out gathered by pair_out with the forward weight reduction-major,
 * stage 2 over pos_in.
 * ------------------------------------------------------------------------------------------ */
int pv2_pair_positions(const int32_t* pair_out, const int32_t* pair_in, const int32_t* kstart, int K,
                       int64_t n_pairs_bound, int64_t out_stride, int64_t in_stride,
                       int32_t* pos_out, int32_t* pos_in, pv2_stream_t stream);
int pv2_spconv_products(const float* in_feat, int c_in, const float* weight, int K, int c_out,
                        int weight_reduction_major, const int32_t* pair_in, const int32_t* kstart,
                        const int32_t* tile_start, int64_t n_tiles, float* prod,
                        pv2_stream_t stream);
int pv2_spconv_reduce_rows(const float* prod, const int32_t* pos, int64_t pos_stride, int K, int c,
                           int64_t n_rows, const float* bias, const float* addend, float* out,
                           float* bn_partial, int* bn_blocks, pv2_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Output-stationary form over MASK-GROUPED rows (csrc/sparse_conv_osm.hip, round 5; same reference
 * call sites: spconv_unet_v1m1_base.py:47-83,112-121,135-146,171-181): ONE launch per conv and
 * direction - no product rows, no row reduce, no atomics, no zero-fill; every output element is
 * written once, offsets summed in ascending table order inside the MFMA accumulators (bitwise
 * reproducible).  The rows of a 32-row tile are chosen to share their kernel offsets:
 *
 *   pv2_osm_plan: from a gather table tbl[k*stride + o] (input row feeding output row o under
 *       offset k, or -1; K <= 31; rows >= *n_cols_dev - when given - are padding) build
 *         perm  [n_cols]      rows sorted stably by the bit mask of their present offsets
 *                             (padding rows last),
 *         tblp  [K, n_pad]    tblp[k*n_pad + r] = tbl[k*stride + perm[r]] (-1 past the valid rows),
 *         tmask [n_pad / 32]  OR of the masks of sorted rows 32 t .. 32 t + 31.
 *       n_pad: a multiple of 256, >= n_cols.  workspace: pv2_osm_plan_workspace_bytes(n_cols).
 *   pv2_spconv_osm: out[perm[r], :] = (addend[perm[r], :]) + sum_k W[kw(k)] . in[tblp[k][r], :] for
 *       r < n_out, kw(k) = plan->kflip ? K-1-k : k.  weight [c_out, K, c_in], or - with
 *       weight_reduction_major != 0 - [c_in, K, c_out] (grad-input reads the forward weight in
 *       place).  c_in % 32 == 0, c_out % 4 == 0.  addend may be NULL or alias out.  zero_row: at least
 *       c_in zero floats.  bn_partial != NULL: the BatchNorm statistics of `out` per block of
 *       *bn_rows_per_block sorted rows - partial[b][0..c) = column sums, [c..2c) = sums of squares
 *       about the BLOCK mean (combined by pv2_bn_forward's second stage in double precision);
 *       *bn_blocks <= PV2_BN_MAX_PARTIAL_BLOCKS.
 *   Arithmetic: fp32 products as six bf16 MFMAs over three exact bf16 pieces per operand
 *   (csrc/mfma_split.h), fp32 accumulation.  Rows without a neighbour under an offset contribute
 *   0 * w: with a non-finite WEIGHT the whole 32-row tile turns NaN (the pair-major forms confine it
 *   to the rows that have the neighbour); non-finite FEATURES stay in their own rows.
 * ------------------------------------------------------------------------------------------ */
#define PV2_BN_MAX_PARTIAL_BLOCKS 8192
typedef struct pv2_osm_plan_t {
  const int32_t* tblp;
  const int32_t* perm;
  const uint32_t* tmask;
  int64_t n_pad;
  int32_t kflip;
  int32_t reserved;
} pv2_osm_plan_t;
/* tuning / test knob: mode 0 / 1 / 2 = PV2_CONV_OSM off / on / auto for pv2_convbn_*; nb (1..4, 0 = automatic)
 * and wr (2 | 4, 0 = automatic) force the tile of pv2_spconv_osm; min_wgs the workgroup target of the
 * automatic choice.  Negative (min_wgs: <= 0) arguments leave a setting as it is. */
int pv2_debug_set_osm(int mode, int nb, int wr, int min_wgs);
size_t pv2_osm_plan_workspace_bytes(int64_t n_cols);
int pv2_osm_plan(const int32_t* tbl, int K, int64_t n_cols, int64_t stride, const int32_t* n_cols_dev,
                 int32_t* perm, int32_t* tblp, uint32_t* tmask, int64_t n_pad, void* workspace,
                 size_t workspace_bytes, pv2_stream_t stream);
int pv2_spconv_osm(const float* in_feat, int c_in, const float* weight, int K, int c_out,
                   int weight_reduction_major, const pv2_osm_plan_t* plan, int64_t n_out,
                   const float* zero_row, const float* addend, float* out, float* bn_partial,
                   int* bn_blocks, int* bn_rows_per_block, pv2_stream_t stream);

/* One conv -> BatchNorm1d -> (+ shortcut) -> ReLU unit of the backbone as ONE call per direction
 * (BasicBlock.forward, spconv_unet_v1m1_base.py:70-83; the conv + norm_fn + ReLU stages :108,
 * 120-121,143-145): the product-row conv, the statistics in its reduce epilogue, combine + apply.
 * The geometry of the rulebook travels as one struct (device pointers + host sizes):
 *   tile_start / n_tiles       prefix of 128-pair tiles per offset (pv2_tile_prefix)
 *   tile_start_w / tile_pairs_w / n_tiles_w   the weight gradient's chunking (pv2_spconv_wgrad_tile)
 *   pos_out / pos_in           position tables (pv2_pair_positions) with their row strides
 * forward : y_conv = conv(x) [n_out, c_out] (kept for the backward), mean_invstd [2*c_out],
 *           out = [relu]( bn(y_conv) [+ residual] ); running statistics updated as pv2_bn_forward.
 *           prod_ws >= n_pairs * c_out floats, stats_ws = pv2_bn_workspace_floats(c_out).
 * backward: gsum [2*c_out] = (d bias, d weight) of the BatchNorm, dy [n_out, c_out] = gradient of
 *           y_conv (scratch the caller keeps alive until the side stream has joined), dres = masked
 *           grad_out (NULL without a shortcut), dx [n_in, c_in] (NULL: not needed), dweight
 *           [c_out, K, c_in] by the deterministic two-stage weight gradient on `side_stream` (NULL
 *           handle: on `stream`) behind an event; part_ws >= pv2_spconv_wgrad_partial_floats(...).
 *           prod_ws >= n_pairs * c_in floats.  out_or_null: the unit's output (the ReLU mask).
 * c_in % 32 == 0 and c_out % 32 == 0 (every unit of SpUNet but the 6-channel stem). */
typedef struct pv2_conv_geom {
  int32_t K;
  int32_t tile_pairs_w;
  int64_t n_in, n_out, n_tiles, n_tiles_w, pos_out_stride, pos_in_stride;
  const int32_t* pair_in;
  const int32_t* pair_out;
  const int32_t* kstart;
  const int32_t* tile_start;
  const int32_t* tile_start_w;
  const int32_t* pos_out;
  const int32_t* pos_in;
  /* round 5: the mask-grouped output-stationary route (pv2_spconv_osm below).  osm_fwd walks the OUTPUT
   * rows of the conv (forward pass), osm_bwd its INPUT rows (grad-input); tblp == NULL: not planned, the
   * unit takes the product-row route.  zero_row: >= 4096 zero floats (what absent neighbours read). */
  pv2_osm_plan_t osm_fwd, osm_bwd;
  const float* zero_row;
} pv2_conv_geom;
int pv2_convbn_forward(const pv2_conv_geom* g, const float* x, int c_in, const float* weight,
                       int c_out, const float* bn_weight, const float* bn_bias,
                       const float* residual, int relu, float eps, float momentum,
                       float* running_mean, float* running_var, float* prod_ws, float* stats_ws,
                       float* y_conv, float* mean_invstd, float* out, pv2_stream_t stream);
int pv2_convbn_backward(const pv2_conv_geom* g, const float* grad_out, const float* x, int c_in,
                        const float* weight, int c_out, const float* y_conv,
                        const float* out_or_null, const float* mean_invstd, const float* bn_weight,
                        float* prod_ws, float* stats_ws, float* gsum, float* dy,
                        float* dres_or_null, float* dx_or_null, float* dweight_or_null,
                        float* part_ws, pv2_stream_t stream, pv2_stream_t side_stream);

/* ------------------------------------------------------------------------------------------
 * Dense fp32 MFMA GEMMs for the MLP heads of the render field (SDF / RGB / semantic decoders).
 * Replaces the rocBLAS/hipBLASLt calls behind nn.Linear and its autograd at
 * ponder/models/ponder/render_utils/decoders.py:21-35,57-76,91-109 (M = rays x samples rows).
 *   pv2_gemm_nt:  y[m, n] = sum_k x[m, k] * w[n, k] (+ bias[n]);  k % 8 == 0; bias may be NULL.
 *   pv2_gemm_tn:  c[i, j] += sum_m a[m, i] * b[m, j];  k1 % 4 == 0, k2 % 4 == 0; c ZERO-initialised.
 * ------------------------------------------------------------------------------------------ */
int pv2_gemm_nt(const float* x, int64_t m, int k, const float* w, int n, const float* bias,
                float* y, pv2_stream_t stream);
int pv2_gemm_tn(const float* a, const float* b, int64_t m, int k1, int k2, float* c,
                pv2_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The same sparse convolutions on 16-bit features (bf16 / fp16 storage, fp32 accumulation on
 * v_mfma_f32_32x32x16_{bf16,f16}): what spconv runs under the reference's shipped training mode
 * (enable_amp = True, configs/scannet/pretrain-ponder-spunet-v1m1-0-base.py:12; the autocast region
 * of ponder/engines/train.py:183-196).  Same call sites as above.
 *   pv2_spconv16_pack_weights: the fp32 master weight [c_out, K, c_in] -> two 16-bit copies in
 *       MFMA-fragment order (once per optimiser step): packed_fwd for the forward pass
 *       (pv2_spconv16_packed_elems(c_out, K, c_in) elements) and packed_bwd, the transposed
 *       operand of the grad-input pass (pv2_spconv16_packed_elems(c_in, K, c_out) elements).
 *   pv2_spconv16_os_forward: the output-stationary conv of pv2_spconv_os_forward (gather table,
 *       perm, kflip and bias as there; no atomics, every element written once, bitwise
 *       reproducible) on 16-bit in_feat [n_in, c_in] -> out_feat [n_out, c_out] with a PACKED
 *       weight.  Grad-input = the same call on grad_out with packed_bwd, channel counts swapped
 *       and the transposed gather table.  c_in % 8 == 0, c_out % 8 == 0.
 *   pv2_spconv16_backward_weight: dweight (fp32 [c_out, K, c_in], ZERO-initialised by the caller)
 *       += the pair-major reduction over 16-bit in_feat / grad_out; tile_start / n_tiles count
 *       chunks of tile_pairs pairs (a multiple of 64, at most 512: PV2_WGRAD_TILE).
 * dtype: PV2_BF16 or PV2_F16.
 * ------------------------------------------------------------------------------------------ */
int64_t pv2_spconv16_packed_elems(int rows, int K, int reduction);
int pv2_spconv16_pack_weights(const float* weight, int c_out, int K, int c_in, int dtype,
                              void* packed_fwd, void* packed_bwd, pv2_stream_t stream);
int pv2_spconv16_os_forward(const void* in_feat, int64_t n_in, int c_in, const void* packed_weight,
                            int K, int c_out, int dtype, const int32_t* nbr, int64_t nbr_stride,
                            const int32_t* perm, int kflip, const float* bias, void* out_feat,
                            int64_t n_out, pv2_stream_t stream);
int pv2_spconv16_backward_weight(const void* in_feat, int64_t n_in, int c_in, const void* grad_out,
                                 int64_t n_out, int c_out, int dtype, int K, const int32_t* pair_in,
                                 const int32_t* pair_out, const int32_t* kstart,
                                 const int32_t* tile_start, int tile_pairs, int64_t n_tiles,
                                 float* grad_weight, pv2_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The sparse U-Net as ONE call per direction (csrc/spunet_exec.hip): SpUNetBase.forward
 * (ponder/models/sparse_unet/spconv_unet_v1m1_base.py:242-278 - conv_input, 4 x (down + residual
 * blocks), 4 x (up + skip concat + residual blocks)) and its autograd graph as a flat array of
 * records walked natively, instead of a Python call + autograd node per module.  The caller owns
 * every buffer (arenas carved by the host code) and fills the pointers; sizes are host values.
 *   PV2_UNET_CONV_BN  conv -> BatchNorm1d -> (+ residual) -> ReLU on the product-row path: exactly
 *                     pv2_convbn_forward / pv2_convbn_backward with this record's fields.
 *   PV2_UNET_STEM     the 125-offset stem: pv2_spconv_os_forward over (nbr, nbr_stride, kflip) +
 *                     pv2_bn_forward; backward: pv2_bn_backward + the deterministic weight gradient
 *                     (geom's pair lists; no grad-input).
 *   PV2_UNET_CONCAT   out[n_out, c_in + c_out] = [x | residual] (the decoder's skip concat);
 *                     backward: dx = grad_out[:, :c_in], dres (+)= grad_out[:, c_in:].
 * forward fields : x, residual (or NULL), weight, bn_weight, bn_bias, running_mean / _var, eps,
 *                  momentum, relu -> y_conv, mean_invstd, out.
 * backward fields: grad_out (complete when the record is reached: records run last to first) ->
 *                  gsum, dy (scratch, kept until the side stream has joined), dres (or NULL), dx (or
 *                  NULL; dx_accumulate != 0: ADD to what dx holds - an earlier record wrote this
 *                  activation's gradient first; for CONCAT the flag applies to dres), dweight.
 *                  STEM with dx != NULL (the input features carry a gradient: a learnable mask token
 *                  was written into them): weight_t = the weight transposed to [c_in, K, c_out].
 * Workspaces as for pv2_convbn_*: prod_ws >= max pairs x channels floats, stats_ws =
 * pv2_bn_workspace_floats(max channels), part_ws the weight gradient's partial sums (side stream).
 * ------------------------------------------------------------------------------------------ */
#define PV2_UNET_CONV_BN 0
#define PV2_UNET_STEM 1
#define PV2_UNET_CONCAT 2
#define PV2_UNET_CONV_BN16 3   /* a conv + BatchNorm unit on 16-bit activations (dtype; sparse_conv16.hip) */
typedef struct pv2_unet_op {
  int32_t kind, c_in, c_out, relu;
  int32_t K, kflip, dx_accumulate;
  int32_t dx_producer;     /* CONV_BN: 1 + index of the unit whose output is x when this unit's grad-input is the
                            * LAST gradient x receives in backward order (0: not so, or x is no unit's output):
                            * that unit's BatchNorm backward sums ride in this unit's row reduce */
  int64_t n_in, n_out, nbr_stride;
  const pv2_conv_geom* geom;
  const int32_t* nbr;
  const float* x;
  const float* residual;
  const float* weight;
  const float* bn_weight;
  const float* bn_bias;
  float* running_mean;
  float* running_var;
  float* y_conv;
  float* mean_invstd;
  float* out;
  const float* grad_out;
  float* dy;
  float* gsum;
  float* dres;
  float* dx;
  float* dweight;
  const float* weight_t;   /* STEM only, with dx: the weight as [c_in, K, c_out] (grad-input pass) */
  float eps, momentum;
  /* The reduced-precision training mode (enable_amp, configs/scannet/pretrain-ponder-spunet-v1m1-0-base.py:12;
   * csrc/sparse_conv16.hip): dtype = PV2_BF16 / PV2_F16 is the element type of x, residual, y_conv, out and
   * of grad_out, dy, dres, dx for CONV_BN16 and CONCAT (widths then count 16-bit elements, multiples of 8),
   * of out / residual / grad_out / dres only for the STEM (its conv stays fp32: the fp32 -> 16-bit edge is
   * its BatchNorm).  Statistics, parameters and every parameter gradient stay fp32.  CONV_BN16 reads the
   * packed weights of pv2_spconv16_pack_weights and walks gather tables: nbr / nbr_stride / perm / kflip
   * over its output rows, the *_t set over its input rows (grad-input); tile_start16 / n_tiles16: chunks of
   * 512 pairs (the weight gradient, which ADDS into dweight - the executor clears it first).  dx_tmp: scratch
   * [n_in, c_in] for a grad-input that is added to an existing gradient (dx_accumulate). */
  int32_t dtype, kflip_t;
  int64_t nbr_t_stride, n_tiles16;
  const void* packed_fwd;
  const void* packed_bwd;
  const int32_t* nbr_t;
  const int32_t* perm;
  const int32_t* perm_t;
  const int32_t* tile_start16;
  void* dx_tmp;
} pv2_unet_op;
int pv2_unet_forward(const pv2_unet_op* ops, int n_ops, float* prod_ws, float* stats_ws,
                     pv2_stream_t stream);
int pv2_unet_backward(const pv2_unet_op* ops, int n_ops, float* prod_ws, float* stats_ws,
                      float* part_ws, pv2_stream_t stream, pv2_stream_t side_stream);
/* The same with CHECKPOINTS for a gradient reduction that overlaps the backward pass (what the
 * reference gets from DistributedDataParallel's bucketed all-reduce, ponder/engines/defaults.py:22-43):
 * the executor walks the units last to first, so the parameter gradients of units >= ckpt_unit[j] are
 * complete once unit ckpt_unit[j] has been processed.  At that point ckpt_event_main[j] (a hipEvent_t,
 * host array of handles) is recorded on `stream` (BatchNorm gradients) and ckpt_event_side[j] on the side
 * stream (weight gradients; on `stream` when side_stream is NULL).  ckpt_unit: host array, strictly
 * descending.  n_ckpt == 0: exactly pv2_unet_backward. */
int pv2_unet_backward_ev(const pv2_unet_op* ops, int n_ops, float* prod_ws, float* stats_ws,
                         float* part_ws, pv2_stream_t stream, pv2_stream_t side_stream, int n_ckpt,
                         const int32_t* ckpt_unit, void* const* ckpt_event_main,
                         void* const* ckpt_event_side);

/* ------------------------------------------------------------------------------------------
 * 2 x 2 x 2 max-pooling of a channels-last dense grid x[B, Z, Y, X, C] -> y[B, Z/2, Y/2, X/2, C]
 * (floor sizes, no padding): nn.MaxPool3d(kernel_size=2) of the projection network's encoders
 * (ponder/models/ponder/unet3d.py:326-330) without ATen's transposes through NCDHW.  idx: one
 * uint32 per 4 output channels = four 8-bit window positions (z*4 + y*2 + x) of the maxima (first
 * maximum wins, as max_pool3d_with_indices).  backward writes EVERY element of grad_x (zeros
 * outside the maxima): nothing to clear, no atomics.  C % 4 == 0.
 * ------------------------------------------------------------------------------------------ */
int pv2_maxpool3d_cl_forward(const float* x, int B, int Z, int Y, int X, int C, float* y,
                             uint32_t* idx, pv2_stream_t stream);
int pv2_maxpool3d_cl_backward(const float* grad_y, const uint32_t* idx, int B, int Z, int Y, int X,
                              int C, float* grad_x, pv2_stream_t stream);
/* The same with the pooled tensor's OTHER gradient added in: grad_x = addend + unpool(grad_y).  In the
 * U-Net an encoder level's output feeds both the next level's pooling and a decoder level's skip sum
 * (unet3d.py:401-411); autograd would add the two gradients in a separate pass.  addend [as grad_x] or NULL.
 * relu_mask_src [as grad_x] or NULL: the pooled tensor itself when it is a ReLU's output - the sum is
 * zeroed where it was <= 0 (the ReLU backward of the level below, done here instead of by its consumers). */
int pv2_maxpool3d_cl_backward_add(const float* grad_y, const uint32_t* idx, const float* addend,
                                  const float* relu_mask_src, int B, int Z, int Y, int X, int C,
                                  float* grad_x, pv2_stream_t stream);

/* Batched inverse of `batch` row-major n x n matrices, n <= 4 (one launch; Gauss-Jordan with partial
 * pivoting in double precision, result rounded to fp32): the camera / unit-cube transforms of the
 * ray set-up, torch.linalg.inv at ponder/models/ponder/ponder_indoor_base.py:380-470.  Singular
 * input gives inf / nan entries, no error. */
int pv2_small_inverse(const float* a, int64_t batch, int n, float* out, pv2_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Training-mode BatchNorm1d over the active-voxel feature matrix x[n, c], fused with the optional
 * residual add and ReLU that follow it in SpUNet's blocks; and a column sum.
 * Replaces the ATen batch_norm / add / relu kernels behind BasicBlock.forward
 * (ponder/models/sparse_unet/spconv_unet_v1m1_base.py:70-83) and the BatchNorm1d + ReLU entries of
 * its SparseSequential stages (:108,120-121,143-145,178-180).
 *   forward : y = [relu]( (x - mean) * invstd * weight + bias [+ residual] ), biased batch
 *             variance; running_mean / running_var (may be NULL) updated with `momentum` and the
 *             unbiased variance; mean_invstd[2c] = {mean, invstd} is kept for backward.
 *   backward: g = dy * (y > 0) when y is given (fused ReLU), else dy;
 *             dx = weight*invstd*(g - mean(g) - xhat*mean(g*xhat)); dresidual = g (optional);
 *             gsum[0..c) = sum g (= dbias), gsum[c..2c) = sum g*xhat (= dweight).
 * workspace: device scratch of pv2_bn_workspace_floats(c) floats, contents irrelevant on entry (the
 * statistics kernel writes per-block partial sums into it, the apply kernel adds them in a fixed
 * order: no atomics, bitwise reproducible); one buffer serves every layer issued on one stream.
 * c <= 1024.  weight / bias may be NULL (affine=False).
 * ------------------------------------------------------------------------------------------ */
int64_t pv2_bn_workspace_floats(int c);
int pv2_bn_forward(const float* x, int64_t n, int c, const float* weight, const float* bias,
                   const float* residual, int relu, float eps, float momentum,
                   float* running_mean, float* running_var, float* workspace,
                   float* mean_invstd, float* y, pv2_stream_t stream);
int pv2_bn_backward(const float* dy, const float* x, const float* y_or_null,
                    const float* mean_invstd, const float* weight, int64_t n, int c,
                    float* workspace, float* gsum, float* dx, float* dresidual_or_null,
                    pv2_stream_t stream);
/* The same two operations on mixed element types (the reduced-precision training mode,
 * configs/scannet/pretrain-ponder-spunet-v1m1-0-base.py:12 enable_amp): x / dx are `x_dtype`,
 * y / residual / dy / dresidual are `y_dtype`; statistics, parameters and their gradients stay
 * fp32.  Supported (x, y) pairs: (F32, F32), (F32, BF16), (F32, F16), (BF16, BF16), (F16, F16). */
int pv2_bn_forward_mixed(const void* x, int x_dtype, int64_t n, int c, const float* weight,
                         const float* bias, const void* residual, int relu, float eps,
                         float momentum, float* running_mean, float* running_var,
                         float* workspace, float* mean_invstd, void* y, int y_dtype,
                         pv2_stream_t stream);
int pv2_bn_backward_mixed(const void* dy, const void* x, int x_dtype, const void* y_or_null,
                          int y_dtype, const float* mean_invstd, const float* weight, int64_t n,
                          int c, float* workspace, float* gsum, void* dx,
                          void* dresidual_or_null, pv2_stream_t stream);
/* out[c] = sum_r x[r, c] */
/* Training-mode statistics of the rows of x [n, c] without applying them: mean_invstd [2c], the
 * running statistics (as pv2_bn_forward), and affine [2c] = (weight * invstd, bias - mean * weight *
 * invstd) - BatchNorm as one multiply-add, folded into the load path of the convolution that follows
 * it (nn.BatchNorm3d in SingleConv order "bcr", unet3d.py:125-156 -> pv2_dconv3_forward's in_scale /
 * in_shift).  workspace: pv2_bn_workspace_floats(c). */
int pv2_bn_statistics(const float* x, int64_t n, int c, const float* weight, const float* bias,
                      float eps, float momentum, float* running_mean, float* running_var,
                      float* workspace, float* mean_invstd, float* affine, pv2_stream_t stream);
/* The same for a matrix followed by `extra_zero_rows` all-zero rows that are not stored (the empty
 * cells of the dense grid whose occupied cells are the rows of x; negative when x carries zero rows
 * the matrix does not have - capacity-sized cell arrays), and the backward of that
 * normalisation: total_parts [n_total_parts][c], added in order, is the column sum of the gradient over
 * ALL rows, stored or not.  gsum [2c] = (d bias, d weight).  See pv2_cells_* below. */
int pv2_bn_statistics_padded(const float* x, int64_t n_rows, int64_t extra_zero_rows, int c,
                             const float* weight, const float* bias, float eps, float momentum,
                             float* running_mean, float* running_var, float* workspace,
                             float* mean_invstd, float* affine, pv2_stream_t stream);
int pv2_bn_backward_padded(const float* dy, const float* x, int64_t n_rows, int64_t extra_zero_rows,
                           int c, const float* mean_invstd, const float* weight,
                           const float* total_parts, int n_total_parts, float* workspace,
                           float* gsum, float* dx, pv2_stream_t stream);
int pv2_col_sum(const float* x, int64_t n, int c, float* out, pv2_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * First level of the projection network from the occupied cells (csrc/cells_level.hip).  The
 * reference scatters the backbone features into a dense grid (ponder_indoor_base.py:177-342
 * to_dense) and runs UNet3D's first BatchNorm3d -> Conv3d(3x3x3) -> ReLU on it (unet3d.py:292-318);
 * here the normalised grid is  y0 + [occupied] * x * scale  and the convolution is a constant part
 * (by border class) plus a sparse convolution of the cell rows.
 *   pv2_cells_tap_table      table [27][cap] int32: the output row cell i (dense row lin[i], -1 =
 *                            padding) feeds through tap k, or -1
 *   pv2_cells_fold_weights   conv weight (any strides, in elements) -> w_okc / ws_okc [c_out][27][c_in]
 *                            (plain / times scale) and u [27][c_out] = sum_c W[o,c,tap] * y0[c];
 *                            affine = [scale | y0] as pv2_bn_statistics_padded writes it
 *   pv2_cells_expand         out (B, Z, Y, X, c_out) = bias + sum of u over the taps inside the grid
 *   pv2_cells_backward_table g (B, Z, Y, X, c_out) -> gu [27][c_out] (gradient of u) and
 *                            gy0_parts [27][c_in] (tap-wise shares of d y0); workspace:
 *                            pv2_cells_backward_workspace_floats
 *   pv2_cells_dw_finish      dW[o,c,tap] = dws_okc[o,tap,c] * scale[c] + gu[tap][o] * y0[c] */
int pv2_cells_tap_table(const int64_t* lin, int64_t cap, int z, int y, int x, int32_t* table,
                        pv2_stream_t stream);
int pv2_cells_fold_weights(const float* weight, int64_t s_out, int64_t s_in, int64_t s_z, int64_t s_y,
                           int64_t s_x, int c_out, int c_in, const float* affine, float* w_okc,
                           float* ws_okc, float* u, pv2_stream_t stream);
int pv2_cells_expand(const float* u, const float* bias_or_null, int b, int z, int y, int x, int c_out,
                     float* out, pv2_stream_t stream);
int64_t pv2_cells_backward_workspace_floats(int b, int z, int y, int c_out);
int pv2_cells_backward_table(const float* g, int b, int z, int y, int x, int c_out,
                             const float* weight, int64_t s_out, int64_t s_in, int64_t s_z,
                             int64_t s_y, int64_t s_x, int c_in, float* workspace, float* gu,
                             float* gy0_parts, pv2_stream_t stream);
int pv2_cells_dw_finish(const float* dws_okc, const float* gu, const float* affine, int c_out, int c_in,
                        float* dweight, int64_t s_out, int64_t s_in, int64_t s_z, int64_t s_y,
                        int64_t s_x, pv2_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Alpha compositing along rays ("ray march" of SURVEY.md section 8b).  Replaces the cumprod /
 * elementwise / reduction chain of
 *   ponder/models/ponder/render_utils/rays.py:83-105   get_weights_and_transmittance_from_alphas
 *   ponder/models/ponder/render_utils/renderers.py:5-75  RGB / Depth / Normal / Semantic renderers
 * All arrays fp32, row-major, rays outermost: alpha / weights [n_rays, n_samples],
 * transmittance [n_rays, n_samples + 1] (optional), values [n_rays, n_samples, n_features].
 *   weights  : T_s = prod_{j<s}(1 - alpha_j + 1e-7),  w_s = alpha_s * T_s   (n_samples <= 256)
 *              backward: grad_alpha from grad_weights (the transmittance output carries no gradient)
 *   accumulate: out[r, f] = sum_s w[r, s] * values[r, s, f]                  (n_features <= 512)
 *              backward: grad_weights[r, s] = sum_f values * grad_out, grad_values = w * grad_out;
 *              either output pointer may be NULL
 * ------------------------------------------------------------------------------------------ */
int pv2_raymarch_weights_forward(const float* alpha, int64_t n_rays, int n_samples, float* weights,
                                 float* transmittance_or_null, pv2_stream_t stream);
int pv2_raymarch_weights_backward(const float* alpha, const float* grad_weights, int64_t n_rays,
                                  int n_samples, float* grad_alpha, pv2_stream_t stream);
int pv2_raymarch_accumulate_forward(const float* weights, const float* values, int64_t n_rays,
                                    int n_samples, int n_features, float* out, pv2_stream_t stream);
int pv2_raymarch_accumulate_backward(const float* weights, const float* values, const float* grad_out,
                                     int64_t n_rays, int n_samples, int n_features,
                                     float* grad_weights_or_null, float* grad_values_or_null,
                                     pv2_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused ray march of the NeuS render head (SURVEY.md section 8b `raymarch_{fwd,bwd}`): sampling,
 * trilinear feature lookup, SDF / colour MLPs on f32 MFMA, grad sdf, NeuS alpha; together with the
 * compositing entry points above this is the whole per-sample part of
 *   ponder/models/ponder/render_utils/ray_samplers.py:55-107,227-322,355-463   (samplers)
 *   ponder/models/ponder/render_utils/rays.py:118-153                          (merge)
 *   ponder/models/ponder/render_utils/fields/sdf_field.py:122-146,148-197,211-284
 *   ponder/models/ponder/render_utils/decoders.py:6-76
 * for the shipped head shape (configs/scannet/pretrain-ponder-spunet-v1m1-0-base.py:33-57): volume
 * [B, Z, Y, X, 128] channels-last fp32, first 64 channels -> SDF MLP (hidden 128, one block,
 * Softplus(beta=100)) -> 1 + 64, colour head 134 -> 3 with no hidden block, points_factor 0, zeros
 * padding, align_corners.  pv2_neus_head_dims reports the compiled-in widths.  Layers without an
 * activation between them are passed COLLAPSED (row-major fp32):
 *   mw [256,64] = [W0 Wc0 ; Wc1], c0 [128] = W0 bc0 + b0, bc1 [128], w1 [65,128], b1 [65],
 *   m_t [64,128] = (mw[:128])^T, q0 [64] = mw[128:]^T w1[0], a_rgb [3,134] = Wr1 Wrc,
 *   b_rgb [3] = Wr1 brc + br1, inv_s [1] (device scalar), w1g_t [128,64] = (w1[1:])^T,
 *   wc1_t [64,128] = (mw[128:])^T.
 * Rays are scene-major with n_rays / vol_b rays per scene; sample n = ray * n_samples + k.
 *
 * coarse_sample: per ray, n_coarse stratified bins (lin_bins [n_coarse+1] = linspace(0,1); t_rand
 *   [n_rays, 1 | n_coarse+1] or NULL), SDF of the UN-normalised start positions, fixed-inv_s
 *   weights, n_importance inverse-CDF samples (lin_u [n_importance+1] = linspace(0, 1-1/nb, nb);
 *   u_rand [n_rays, 1 | nb] or NULL), sorted merge.  Writes bins [n_rays, S+1] (spacing), starts and
 *   deltas [n_rays, S] (ray distance), S = n_coarse + n_importance.  dbg_* may be NULL.
 * field_forward: per sample sdf, alpha and the value row
 *   values[n, 140] = f'(64) geo(64) grad(3) normal(3) rgb(3) t 1 0
 *   (composite it with pv2_raymarch_weights_forward + pv2_raymarch_accumulate_forward), and the
 *   activations the backward needs: save_f [N,64], save_h0 [N,128], save_a1 [N,128], save_q [N,64].
 * field_backward: given g_alpha [N] (pv2_raymarch_weights_backward of the accumulate backward's
 *   grad_weights), weights [N], g_comp [n_rays,140] (gradient of the composited value row), optional
 *   g_sdf [N] / g_grad [N,3]: writes gfeat [N,128], gvec [N,4] (total d loss / d grad sdf), the
 *   weight-gradient GEMM operands gz [N,256] = [gh0 | ga1], tmat [N,128], gq [N,64], gh [N,68] =
 *   [gsdf, ggeo, 0..], gy [N,4], the column sums `sums` (layout in raymarch_fused.hip; length from
 *   pv2_neus_head_dims) and, when grad_volume != NULL (ZERO-initialised or holding other
 *   contributions), scatter-adds the volume gradient including the second-order term.
 * ------------------------------------------------------------------------------------------ */
int pv2_neus_head_dims(int* hidden, int* f_sdf, int* f_rest, int* geo, int* value_row, int* sums_len);
int pv2_neus_coarse_sample(const float* volume, int vol_b, int vol_z, int vol_y, int vol_x,
                           int vol_c, const float* origins, const float* dirs, const float* nears,
                           const float* fars, int64_t n_rays, int n_coarse, int n_importance,
                           const float* lin_bins, const float* t_rand, int t_rand_cols,
                           const float* lin_u, const float* u_rand, int u_rand_cols,
                           const float* mw, const float* c0, const float* bc1, const float* w1,
                           const float* b1, float base_inv_s, float* bins_out, float* starts_out,
                           float* deltas_out, int32_t* dbg_idx, float* dbg_sdf, float* dbg_w,
                           pv2_stream_t stream);
int pv2_neus_field_forward(const float* volume, int vol_b, int vol_z, int vol_y, int vol_x,
                           int vol_c, const float* origins, const float* dirs, const float* starts,
                           const float* deltas, int64_t n_rays, int n_samples, const float* mw,
                           const float* c0, const float* bc1, const float* w1, const float* b1,
                           const float* m_t, const float* q0, const float* a_rgb,
                           const float* b_rgb, const float* inv_s, int norm_pts, float norm_div,
                           float* sdf, float* alpha, float* values, float* save_f, float* save_h0,
                           float* save_a1, float* save_q, pv2_stream_t stream);
int pv2_neus_field_backward(const float* volume, int vol_b, int vol_z, int vol_y, int vol_x,
                            int vol_c, const float* origins, const float* dirs, const float* starts,
                            const float* deltas, int64_t n_rays, int n_samples, const float* mw,
                            const float* w1, const float* m_t, const float* w1g_t,
                            const float* wc1_t, const float* a_rgb, const float* inv_s,
                            int norm_pts, float norm_div, const float* sdf, const float* values,
                            const float* save_h0, const float* save_q, const float* weights,
                            const float* g_alpha, const float* g_sdf, const float* g_grad,
                            const float* g_comp, float* gfeat, float* gvec, float* gz, float* tmat,
                            float* gq, float* gh, float* gy, float* sums, float* grad_volume,
                            pv2_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Ray set-up of the indoor model (csrc/ray_setup.hip): PonderIndoor.to_unit_cube + ray_sample +
 * get_mask_at_box, ponder/models/ponder/ponder_indoor_base.py:344-497 of the reference - fp32 in the
 * reference's operation order (no contraction), the slab test in double.
 *   pv2_unit_cube: coords [N,3] of n_scenes scenes (offsets: int64 cumulative counts), extrinsic
 *     [B,V,4,4] world -> camera (its [3,3] entry is taken as 1), kmat [B,V,3,3], depth_scale [B]
 *     -> scene [B, scene_floats]: scale | t(3) | extent (pc_scale) | new depth scale | bbox lo(3), hi(3);
 *        extrinsic_out [B,V,4,4] = extrinsic . S^-1;  view [B*V, view_floats]: camera pose rows 0..2
 *        ([R | o]) | K^-1 (9) | image-plane normal (3);  coords_out [N,3] the normalised cloud in
 *        metres of the scaled scene.  bounds_ws: 6 B floats of scratch.  Record sizes:
 *        pv2_ray_setup_record_sizes.
 *   pv2_ray_gen: pixels [B,V,n] int64 flat indices y * width + x into colors [B,V,H,W,3], depths
 *     [B,V,H,W], semantic [B,V,H,W] int64 (or NULL) -> ray_o, ray_d [B, V*n, 3], rgb [B*V*n, 3], depth
 *     [B*V*n] (distance along the ray; -0.001 and zero colour for rays that miss the padded cube
 *     bounds_lo / bounds_hi, HOST doubles), semantic_row [B*V*n] = class + 1, 0 for class <= 0 / miss.
 * ------------------------------------------------------------------------------------------ */
int pv2_ray_setup_record_sizes(int* scene_floats, int* view_floats);
int pv2_unit_cube(const float* coords, const int64_t* offsets, int n_scenes, int64_t n_points,
                  int n_views, float z_level, const float* depth_scale, const float* extrinsic,
                  const float* kmat, float* bounds_ws, float* scene, float* extrinsic_out, float* view,
                  float* coords_out, pv2_stream_t stream);
int pv2_ray_gen(const int64_t* pixels, int n_scenes, int n_views, int n_rays, int height, int width,
                const float* view, const float* scene, const float* colors, const float* depths,
                const int64_t* semantic, const double* bounds_lo, const double* bounds_hi, float* ray_o,
                float* ray_d, float* rgb, float* depth, int64_t* semantic_row, pv2_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Per-ray epilogue of the composited rows and the semantic loss term (csrc/ray_epilogue.hip): the tail
 * of SurfaceModel.get_outputs - RGBRenderer / DepthRenderer / SemanticRenderer,
 * ponder/models/ponder/render_utils/renderers.py:5-75 - and the semantic term of SurfaceModel.get_loss
 * (base_surface_model.py:102-211), around three library matrix products.
 *   comp [R, nv] = f'(n_f2) geo(n_geo) grad(3) normal(3) rgb(3) t sum_w pad..; starts [R, S], rays
 *   scene-major with R / n_scenes rays per scene; bg = RGBRenderer.background_color.
 *   pv2_ray_rows_forward: lohi [n_scenes, 2] = min / max sample distance per scene;
 *     xbar [R, n_f2 + n_geo + 4] = [grad | f' | geo | sum_w] (the semantic head's input, biases ride on
 *     the last column); rgb [R,3] = (rgb + bg) - sum_w bg; depth [R] = clamp(t / (sum_w + 1e-10), lo, hi).
 *   pv2_ray_rows_backward: d_comp [R, nv] (every element written) from d_xbar, g_rgb, g_depth (any NULL).
 *   pv2_semantic_ce_forward: raw [R, R] = sem . gt^T, sem / gt [R, c_sem], depth_gt [R]:
 *     l_ij = raw_ij / max(|sem_i|, 1e-12) / temperature; row i counts when depth_gt_i > 0 and gt_i != 0;
 *     info [R, pv2_ray_loss_info_floats()] = {|sem_i|, scale_i, logsumexp_i, ok_i, loss_i, ..}.
 *   pv2_ray_loss_finalize: surface_terms [6] = pv2_surface_loss_forward's out; info may be NULL (no
 *     semantic term) -> out [9] = depth, rgb, psnr, semantic = w_sem sum loss_i / max(count, 1),
 *     free_space, sdf, eikonal, TOTAL (the loss terms added in that order), count.
 *   pv2_semantic_ce_backward: g_total [1] (device) -> d_raw [R, R], d_sem_norm [R, c_sem] = the gradient
 *     of sem through its norm; the caller adds d_raw . gt.
 * ------------------------------------------------------------------------------------------ */
int pv2_ray_loss_info_floats(void);
int pv2_ray_rows_forward(const float* comp, int nv, int n_f2, int n_geo, const float* starts,
                         int64_t n_rays, int n_samples, int n_scenes, float bg0, float bg1, float bg2,
                         float* lohi, float* xbar, float* rgb, float* depth, pv2_stream_t stream);
int pv2_ray_rows_backward(const float* comp, int nv, int n_f2, int n_geo, int64_t n_rays, int n_scenes,
                          float bg0, float bg1, float bg2, const float* lohi, const float* d_xbar,
                          const float* g_rgb, const float* g_depth, float* d_comp, pv2_stream_t stream);
int pv2_semantic_ce_forward(const float* raw, const float* sem, const float* gt, const float* depth_gt,
                            int64_t n_rays, int c_sem, float temperature, float* info,
                            pv2_stream_t stream);
int pv2_ray_loss_finalize(const float* info, int64_t n_rays, float w_sem, const float* surface_terms,
                          float* out, pv2_stream_t stream);
int pv2_semantic_ce_backward(const float* raw, const float* sem, const float* info, const float* g_total,
                             const float* out, int64_t n_rays, int c_sem, float w_sem, float* d_raw,
                             float* d_sem_norm, pv2_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Per-ray / per-sample loss terms of the surface-rendering models (csrc/surface_loss.hip):
 * SurfaceModel.get_loss, ponder/models/ponder/render_utils/models/base_surface_model.py:102-211 -
 * depth, colour (+ psnr), free-space, SDF and eikonal terms; the semantic term is not covered.
 *   depth, depth_gt [R]; rgb, rgb_gt [R,3] or both NULL; sdf, z [R,S]; grad [R,S,3] or NULL.
 *   weights [5] (device): depth, rgb, free_space, sdf, eikonal.  trunc = sensor_depth_truncation.
 *   forward: out [6] = depth_loss, rgb_loss, psnr, free_space_loss, sdf_loss, eikonal_loss; sums [9] is
 *     kept for the backward; workspace: pv2_surface_loss_workspace_floats() floats.
 *   backward: upstream = 6 device pointers to the scalar gradients of `out` (NULL = none) ->
 *     g_depth [R], g_rgb [R,3] (or NULL), g_sdf [R,S], g_grad [R,S,3] (or NULL), every element written.
 * Reproducible: per-workgroup partial sums added in workgroup order.
 * ------------------------------------------------------------------------------------------ */
int pv2_surface_loss_workspace_floats(void);
int pv2_surface_loss_forward(const float* depth, const float* depth_gt, const float* rgb,
                             const float* rgb_gt, const float* sdf, const float* z, const float* grad,
                             int64_t n_rays, int n_samples, float trunc, const float* weights,
                             float* workspace, float* out, float* sums, pv2_stream_t stream);
int pv2_surface_loss_backward(const float* depth, const float* depth_gt, const float* rgb,
                              const float* rgb_gt, const float* sdf, const float* z, const float* grad,
                              int64_t n_rays, int n_samples, float trunc, const float* weights,
                              const float* sums, const float* const* upstream, float* g_depth,
                              float* g_rgb, float* g_sdf, float* g_grad, pv2_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The NeuS head for NARROW SDF decoders (csrc/raymarch_narrow.hip): the head of the reference's
 * nuScenes configuration (configs/nuscenes/pretrain-ponder-spunet-v1m1-0-base.py: SDFField with
 * sdf_decoder = dict(in_dim=32, out_dim=16+1, hidden_size=16, n_blocks=5), no colour / semantic
 * decoder, depth loss).  Replaces, per call, what the reference runs as ~500 small autograd ops:
 * ponder/models/ponder/render_utils/decoders.py:6-36 (SDFDecoder: x = fc_p(p) * points_factor,
 * x = lin_l(x + fc_c[l](feat)), Softplus(beta=100)), fields/sdf_field.py:185-197, 211-284, 122-146
 * (get_sdf, grad sdf by autograd, NeuS alphas), rays.py:83-105, renderers.py:33-45 (weights,
 * expected depth) and ray_samplers.py:355-463 (coarse pass + importance sampling).
 *   theta [theta_len]: Wp [H,3] bp [H] | l = 0..L-1: Wc_l [H,C] bc_l [H] | l = 0..L-2: W_l [H,H]
 *     b_l [H] | w_last [H] b_last  (fc_p, fc_c[l], lin_l, row 0 of the last lin; C, H, L and
 *     theta_len from pv2_narrow_head_dims: 32, 16, 6, 4609).  points_factor multiplies fc_p(p).
 *   volume: channels-last [B, Z, Y, X, C]; rays are scene-major, n_rays / B per scene; zeros
 *     padding, align_corners, points NOT normalised (SDFField.norm_pts = False).
 *   coarse_sample: as pv2_neus_coarse_sample (same outputs, same random-number inputs).
 *   field_forward: sdf [N], grad [N,3] (N = n_rays * n_samples, sample n = ray * n_samples + k),
 *     weights [N], trans [N] (transmittance in front of sample k), comp [n_rays, 2] =
 *     [sum_k w_k starts_k, sum_k w_k]; inv_s: one device float.
 *   field_backward: upstream g_comp [n_rays,2] and optional (NULL = zero) g_weights [N], g_sdf [N],
 *     g_grad [N,3].  Writes work [N,4] (scratch), ADDS the volume gradient into g_volume (caller
 *     zero-initialises; NULL skips it), writes g_theta_slabs [pv2_narrow_backward_slabs(..),
 *     theta_len] - the caller sums the slabs in order - and g_inv_s_part [n_rays] (sum = d/d inv_s).
 *     Second-order terms through grad sdf are included (hand-derived; oracle/narrow_head.py).
 * ------------------------------------------------------------------------------------------ */
int pv2_narrow_head_dims(int* channels, int* hidden, int* layers, int* theta_len);
int pv2_narrow_coarse_sample(const float* volume, int vol_b, int vol_z, int vol_y, int vol_x, int vol_c,
                             const float* origins, const float* dirs, const float* nears,
                             const float* fars, int64_t n_rays, int n_coarse, int n_importance,
                             const float* lin_bins, const float* t_rand, int t_rand_cols,
                             const float* lin_u, const float* u_rand, int u_rand_cols,
                             const float* theta, float points_factor, float base_inv_s, float* bins_out,
                             float* starts_out, float* deltas_out, int32_t* dbg_idx, float* dbg_sdf,
                             float* dbg_w, pv2_stream_t stream);
int pv2_narrow_field_forward(const float* volume, int vol_b, int vol_z, int vol_y, int vol_x, int vol_c,
                             const float* origins, const float* dirs, const float* starts,
                             const float* deltas, int64_t n_rays, int n_samples, const float* theta,
                             float points_factor, const float* inv_s, float* sdf, float* grad,
                             float* weights, float* trans, float* comp, pv2_stream_t stream);
int64_t pv2_narrow_backward_slabs(int64_t n_rays, int n_samples);
int pv2_narrow_field_backward(const float* volume, int vol_b, int vol_z, int vol_y, int vol_x, int vol_c,
                              const float* origins, const float* dirs, const float* starts,
                              const float* deltas, int64_t n_rays, int n_samples, const float* theta,
                              float points_factor, const float* inv_s, const float* sdf,
                              const float* grad, const float* weights, const float* trans,
                              const float* g_comp, const float* g_weights, const float* g_sdf,
                              const float* g_grad, float* work, float* g_volume, float* g_theta_slabs,
                              float* g_inv_s_part, pv2_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The same head with the projection network's final 1x1x1 convolution FOLDED into it.
 * ponder/models/ponder/unet3d.py:637-641,703 (``final_conv = nn.Conv3d(f_maps[0], out_channels,
 * 1)``) writes V = Wf X + bf on every grid cell and ponder/models/ponder/render_utils/fields/
 * sdf_field.py:122-146 then samples V trilinearly.  Sampling is linear in the volume, so
 *     f(p) = Wf xt(p) + bf s(p),   xt = sum_c w_c X_c,   s = sum_c w_c   (in-bounds corners c)
 * and the 128-channel volume (537 MB at the ScanNet grid, as much again for its gradient) need not
 * exist.  volume: the pre-convolution activations X, channels-last [B, Z, Y, X, C] with C and the
 * row width reported by pv2_neus_fold_dims (32 and 40).
 *   fold_gather: per sample (main pass: sample n = ray * n_samples + k at distance starts[n])
 *     gval [N, 40] = [xt(32), s, 0 x 7] and gder [N, 3, 40] = the same of d/dp_a.  Two tall GEMMs
 *     (pv2_gemm_nt) with [Wf | bf | 0] give frows [N, 128] = f and, with its first 64 rows,
 *     jrows [N, 3, 64] = d f_sdf / d p.
 *   field_forward_rows / field_backward_rows: pv2_neus_field_forward / _backward reading those
 *     rows instead of gathering from a volume (no grad_volume: see fold_scatter).
 *   fold_scatter: grad_volume[corner c] += w_c gx + D_c qx with gx = gfeat . Wf [N, 32],
 *     qx = save_q . Wf_sdf [N, 32], D_c = sum_a gvec_a dw_c/dp_a; grad_volume ZERO-initialised.
 *   coarse_sample_folded: pv2_neus_coarse_sample on X with wfs [64, 40] = [Wf_sdf | bf_sdf | 0].
 * ------------------------------------------------------------------------------------------ */
int pv2_neus_fold_dims(int* channels, int* row_width);
int pv2_neus_fold_gather(const float* volume, int vol_b, int vol_z, int vol_y, int vol_x, int vol_c,
                         const float* origins, const float* dirs, const float* starts,
                         int64_t n_rays, int n_samples, int norm_pts, float norm_div, float* gval,
                         float* gder, pv2_stream_t stream);
int pv2_neus_fold_scatter(int vol_b, int vol_z, int vol_y, int vol_x, int vol_c,
                          const float* origins, const float* dirs, const float* starts,
                          int64_t n_rays, int n_samples, int norm_pts, float norm_div,
                          const float* gx, const float* gvec, const float* qx, float* grad_volume,
                          pv2_stream_t stream);
int pv2_neus_coarse_sample_folded(const float* volume, int vol_b, int vol_z, int vol_y, int vol_x,
                                  int vol_c, const float* wfs, const float* origins,
                                  const float* dirs, const float* nears, const float* fars,
                                  int64_t n_rays, int n_coarse, int n_importance,
                                  const float* lin_bins, const float* t_rand, int t_rand_cols,
                                  const float* lin_u, const float* u_rand, int u_rand_cols,
                                  const float* mw, const float* c0, const float* bc1, const float* w1,
                                  const float* b1, float base_inv_s, float* bins_out,
                                  float* starts_out, float* deltas_out, int32_t* dbg_idx,
                                  float* dbg_sdf, float* dbg_w, pv2_stream_t stream);
int pv2_neus_field_forward_rows(const float* frows, const float* jrows, const float* origins,
                                const float* dirs, const float* starts, const float* deltas,
                                int64_t n_rays, int n_samples, const float* mw, const float* c0,
                                const float* bc1, const float* w1, const float* b1, const float* m_t,
                                const float* q0, const float* a_rgb, const float* b_rgb,
                                const float* inv_s, int norm_pts, float norm_div, float* sdf,
                                float* alpha, float* values, float* save_f, float* save_h0,
                                float* save_a1, float* save_q, pv2_stream_t stream);
int pv2_neus_field_backward_rows(const float* jrows, const float* origins, const float* dirs,
                                 const float* starts, const float* deltas, int64_t n_rays,
                                 int n_samples, const float* mw, const float* w1, const float* m_t,
                                 const float* w1g_t, const float* wc1_t, const float* a_rgb,
                                 const float* inv_s, int norm_pts, float norm_div, const float* sdf,
                                 const float* values, const float* save_h0, const float* weights,
                                 const float* g_alpha, const float* g_sdf, const float* g_grad,
                                 const float* g_comp, float* gfeat, float* gvec, float* gz,
                                 float* tmat, float* gq, float* gh, float* gy, float* sums,
                                 pv2_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * Dense-grid scatter (to_dense).  Replaces torch_scatter.scatter(src, index, dim=0,
 * reduce="mean"|"sum", out=...) at ponder/models/ponder/ponder_indoor_base.py:214 and
 * ponder_outdoor_base.py:204.
 * ------------------------------------------------------------------------------------------ */
/* out[index[i], :] += src[i, :];  count[index[i]] += 1.   index: int64 [m], values in [0, g). */
/* out[i, :] = table[index[i], :], c % 4 == 0: the expansion of a small table of rows (the 27 border
 * classes of the dense grid's first convolution, models/ponder/sparse_input.py) into a full matrix. */
int pv2_gather_rows(const float* table, const int32_t* index, int64_t n, int c, float* out,
                    pv2_stream_t stream);
int pv2_scatter_add(const float* src, const int64_t* index, int64_t m, int c, float* out,
                    float* count, int64_t g, pv2_stream_t stream);
/* out[r, :] /= max(count[r], 1) */
int pv2_scatter_mean_finish(float* out, const float* count, int64_t g, int c, pv2_stream_t stream);
/* backward: dsrc[i, :] = dout[index[i], :] / max(count[index[i]], 1)   (count may be NULL: sum) */
int pv2_scatter_backward(const float* dout, const int64_t* index, const float* count, int64_t m,
                         int c, float* dsrc, int64_t g, pv2_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Trilinear ("smooth") grid sampler, forward / backward / backward-of-backward.
 * Replaces smooth_sampler._C.forward / backward / backward_backward
 * (libs/smooth-sampler/smooth_sampler/csrc/smooth_sampler.cpp:36-98, kernels
 * smooth_sampler_kernel.cu:39-153, 155-356, 358-619).
 *
 * Shapes follow torch grid_sample for 5-D input: input (N,C,D,H,W), grid (N,Do,Ho,Wo,3) with
 * grid[...,0] -> W axis.  The grid and every 3-vector tensor are dense [P,3] with
 * P = N*Do*Ho*Wo points, `points_per_n` = Do*Ho*Wo.  Volume-shaped tensors are addressed with
 * explicit ELEMENT strides so that both NCDHW and channels-last (NDHWC) storage work; tensors
 * shaped like the output (N,C,Do,Ho,Wo) are addressed as  n*o_sn + c*o_sc + q*o_sp  with q the
 * point index inside batch item n.
 *
 * padding_mode: 0 zeros, 1 border, 2 reflection (ATen GridSamplerPadding).
 * ------------------------------------------------------------------------------------------ */
typedef struct pv2_volume_desc {
  int64_t n, c, d, h, w;       /* sizes */
  int64_t sn, sc, sd, sh, sw;  /* element strides */
} pv2_volume_desc;

typedef struct pv2_points_desc {
  int64_t n_points;     /* P = N * points_per_n */
  int64_t points_per_n; /* Do*Ho*Wo */
  int64_t o_sn, o_sc, o_sp;
} pv2_points_desc;

int pv2_trilinear_forward_f32(const float* input, const pv2_volume_desc* vol, const float* grid,
                              const pv2_points_desc* pts, float* output, int padding_mode,
                              int align_corners, int apply_smoothstep, pv2_stream_t stream);
int pv2_trilinear_forward_f64(const double* input, const pv2_volume_desc* vol, const double* grid,
                              const pv2_points_desc* pts, double* output, int padding_mode,
                              int align_corners, int apply_smoothstep, pv2_stream_t stream);

/* grad_input (may be NULL to skip; otherwise ZERO-initialised, same strides as `vol`) and
 * grad_grid [P,3] given grad_output (output-shaped, strides from `pts`). */
int pv2_trilinear_backward_f32(const float* grad_output, const float* input,
                               const pv2_volume_desc* vol, const float* grid,
                               const pv2_points_desc* pts, float* grad_input, float* grad_grid,
                               int padding_mode, int align_corners, int apply_smoothstep,
                               pv2_stream_t stream);
int pv2_trilinear_backward_f64(const double* grad_output, const double* input,
                               const pv2_volume_desc* vol, const double* grid,
                               const pv2_points_desc* pts, double* grad_input, double* grad_grid,
                               int padding_mode, int align_corners, int apply_smoothstep,
                               pv2_stream_t stream);

/* Backward of backward.  Inputs: g_ginput (grad wrt the grad_input output; may be NULL = zeros;
 * strides of `vol`), g_ggrid [P,3] (grad wrt the grad_grid output), and the saved input / grid /
 * grad_output.  Outputs: grad_input2 (may be NULL; ZERO-initialised), grad_grid2 [P,3],
 * grad_grad_output (output-shaped, fully written). */
int pv2_trilinear_backward_backward_f32(const float* g_ginput, const float* g_ggrid,
                                        const float* input, const pv2_volume_desc* vol,
                                        const float* grid, const float* grad_output,
                                        const pv2_points_desc* pts, float* grad_input2,
                                        float* grad_grid2, float* grad_grad_output,
                                        int padding_mode, int align_corners, int apply_smoothstep,
                                        pv2_stream_t stream);
int pv2_trilinear_backward_backward_f64(const double* g_ginput, const double* g_ggrid,
                                        const double* input, const pv2_volume_desc* vol,
                                        const double* grid, const double* grad_output,
                                        const pv2_points_desc* pts, double* grad_input2,
                                        double* grad_grid2, double* grad_grad_output,
                                        int padding_mode, int align_corners, int apply_smoothstep,
                                        pv2_stream_t stream);

/* The sampler on 16-bit tensors (dtype: 1 = bfloat16, 2 = float16).  Replaces the half branch of the
 * reference's AT_DISPATCH_FLOATING_TYPES_AND_HALF (libs/smooth-sampler/smooth_sampler/csrc/
 * smooth_sampler_kernel.cu:630,670,726), which its shipped enable_amp=True configs reach.  input,
 * grid, grad_output, g_ginput, g_ggrid, output, grad_grid, grad_grid2, grad_grad_output are
 * 16-bit with the strides of the descriptors; arithmetic is fp32; the atomically accumulated
 * volume gradients grad_input_f32 / grad_input2_f32 (NULL to skip; ZERO-initialised, strides of
 * `vol`) are FP32 buffers the caller narrows once - the reference accumulates them in half. */
int pv2_trilinear_forward_16(const void* input, int dtype, const pv2_volume_desc* vol,
                             const void* grid, const pv2_points_desc* pts, void* output,
                             int padding_mode, int align_corners, int apply_smoothstep,
                             pv2_stream_t stream);
int pv2_trilinear_backward_16(const void* grad_output, const void* input, int dtype,
                              const pv2_volume_desc* vol, const void* grid,
                              const pv2_points_desc* pts, float* grad_input_f32, void* grad_grid,
                              int padding_mode, int align_corners, int apply_smoothstep,
                              pv2_stream_t stream);
int pv2_trilinear_backward_backward_16(const void* g_ginput, const void* g_ggrid, const void* input,
                                       int dtype, const pv2_volume_desc* vol, const void* grid,
                                       const void* grad_output, const pv2_points_desc* pts,
                                       float* grad_input2_f32, void* grad_grid2,
                                       void* grad_grad_output, int padding_mode, int align_corners,
                                       int apply_smoothstep, pv2_stream_t stream);

/* ---- Dense 3x3x3 convolutions of the projection network (csrc/dense_conv.hip) -------------------
 * Replace the cuDNN / MIOpen convolutions behind nn.Conv3d(k3, p1) and nn.ConvTranspose3d(k3, s2, p1)
 * of the reference's UNet3D-v1m2 (ponder/models/ponder/unet3d.py:45-156 SingleConv "bcr", :292-356
 * Encoder, :359-493 Decoder / Upsampling) - forward, grad-input, grad-weight.  Grids are fp32
 * (b, z, y, x, c): the channels_last_3d storage of a (b, c, z, y, x) tensor.
 *
 * pv2_dconv3_pack_weights: the weight in MFMA fragment order, once per optimiser step, for the
 * pv2_dconv3_forward mode it will be used with (mode 0 runs on the bf16 matrix cores: its weights are
 * stored as three bf16 pieces per value, csrc/mfma_split.h; pv2_dconv3_packed_floats sizes the buffer).
 *   packed[t][ck][nb][s][lane][q] = w[out*s_out + red*s_red + kz*s_z + ky*s_y + kx*s_x] with out = nb*32
 *   + (lane & 31), red = ck*16 + 8s + 4(lane >> 5) + q, (kz,ky,kx) = tap (flip ? 26 - t : t).
 *   n_out % 32 == 0 (the output channels of the launch that will use it), n_red % 16 == 0.
 *     conv forward:        out = dim 0 of the Conv3d weight [C_out, C_in, 3,3,3], red = dim 1, flip 0
 *     conv grad-input:     out = dim 1, red = dim 0, flip 1
 *     transposed forward:  ConvTranspose3d weight [C_in, C_out, 3,3,3]: out = dim 1, red = dim 0, flip 0
 *     transposed grad-in:  out = dim 0, red = dim 1, flip 0   (mode 2 below)
 * pv2_dconv3_forward: out = [relu]( conv(mask(x * in_scale + in_shift)) + bias + addend )
 *   mode 0: conv k3 s1 p1, out grid = in grid;  mode 1: transposed conv k3 s2 p1, out = 2 x in per axis
 *   (bias / addend only);  mode 2: strided conv k3 s2 p1, out = in / 2 (even sizes): mode 1's grad-input.
 *   in_scale / in_shift [c_in] or NULL: the BatchNorm3d in front of the conv (zero padding AFTER it);
 *   in_mask_src [as x] or NULL: x is zeroed where in_mask_src <= 0 (ReLU backward); bias [c_out] or
 *   NULL; addend [as out] or NULL (the decoder's skip features, unet3d.py:401-411); out_mask_src [as out]
 *   or NULL: the result is zeroed where out_mask_src <= 0 (modes 0 / 2: a gradient about to pass a ReLU).
 * pv2_dconv3_backward_weight: dw[n*s_n + c*s_c + kz*s_z + ky*s_y + kx*s_x] = sum over cells of
 *   mask(gy)[cell_g][n] * (x * in_scale + in_shift)[cell_x][c]; mode 0 (conv k3 s1 p1: x the conv
 *   input [b,z,y,xx,c_x], gy [b,z,y,xx,c_g]) or mode 1 (transposed: x the coarse input, gy
 *   [b,2z,2y,2xx,c_g]).  Partial slabs in partial_ws (pv2_dconv3_wgrad_partial_floats), summed in a fixed
 *   order: bitwise repeatable.  c_x % 32 == 0, c_g % 32 == 0. */
int64_t pv2_dconv3_packed_floats(int c_out, int c_in, int mode);
int pv2_dconv3_pack_weights(const float* w, int n_out, int n_red, int64_t s_out, int64_t s_red,
                            int64_t s_z, int64_t s_y, int64_t s_x, int flip, int mode, float* packed,
                            pv2_stream_t stream);
/* Reduced-precision products for the dense convolutions (process-wide switch, 0 restores fp32): every
 * product of pv2_dconv3_forward / _backward_weight on the LEADING bf16 piece of each operand only - what
 * the reference's enable_amp = True buys from the library's 16-bit convolutions (ponder/engines/train.py:
 * 183-196, unet3d.py:45-156): operands cut to bf16 by truncation, fp32 sums, fp32 results. */
int pv2_dconv3_set_one_term(int on);
int pv2_dconv3_forward(const float* x, int b, int z, int y, int xx, int c_in, const float* packed_w,
                       int c_out, int mode, const float* in_scale, const float* in_shift,
                       const float* in_mask_src, const float* bias, const float* addend, int relu,
                       const float* out_mask_src, float* out, pv2_stream_t stream);
int64_t pv2_dconv3_wgrad_partial_floats(int b, int z, int y, int xx, int c_x, int c_g, int mode);
int pv2_dconv3_backward_weight(const float* x, int b, int z, int y, int xx, int c_x,
                               const float* in_scale, const float* in_shift, const float* gy,
                               int c_g, const float* gy_mask_src, int mode, float* partial_ws,
                               float* dw, int64_t s_n, int64_t s_c, int64_t s_z, int64_t s_y,
                               int64_t s_x, pv2_stream_t stream);

/* ---- Device GridSample (csrc/voxelize.hip) --------------------------------------------------------
 * Train-mode GridSample of the reference's input pipeline (ponder/datasets/transform.py:1103-1145,
 * hashes :1180-1213) on raw points already in HBM: one representative per occupied voxel, voxels in
 * ascending UNSIGNED key order, de-duplicated on the reference's own 64-bit key (fnv-1a, or ravel when
 * ravel_hash != 0).  Two calls around ONE device -> host read of the voxel count:
 *   stage 1  coord [n,3] (float32, or float64 when coord_is_f64) -> grid [n,3] = floor(coord /
 *            grid_size) - min, slot_of [n], the hash table (table_keys [table_size] pre-filled with
 *            0xff bytes, table_count [table_size] zeroed; table_size a power of two >= 2 n), the
 *            unsorted unique (key, slot) list and *n_vox (zeroed by the caller); minmax [6] scratch.
 *   stage 2  sorts the list (rocPRIM, workspace of pv2_voxelize_workspace_bytes(n_vox)), builds each
 *            voxel's member list and writes idx_unique [n_vox] = the ((draws mod max_count) mod
 *            count)-th member of voxel v in point-index order, grid_coord [n_vox,3] (both int64).
 *            cursor [n_vox] and max_count [1] zeroed by the caller; the rest is scratch. */
int pv2_voxelize_stage1(const void* coord, int coord_is_f64, int64_t n, double grid_size, int ravel_hash,
                        int64_t* minmax, uint64_t* table_keys, int32_t* table_count, int64_t table_size,
                        int32_t* grid, int32_t* slot_of, uint64_t* uniq_keys, int32_t* uniq_slot,
                        int32_t* n_vox, pv2_stream_t stream);
size_t pv2_voxelize_workspace_bytes(int64_t n_vox);
int pv2_voxelize_stage2(int64_t n, int64_t n_vox, const uint64_t* uniq_keys, const int32_t* uniq_slot,
                        const int32_t* table_count, const int32_t* slot_of, const int32_t* grid,
                        const int64_t* draws, uint64_t* sorted_keys, int32_t* sorted_slot,
                        int32_t* rank_of_slot, int32_t* count, int32_t* start, int32_t* cursor,
                        int32_t* members, int32_t* max_count, void* workspace, size_t workspace_bytes,
                        int64_t* idx_unique, int64_t* grid_coord, pv2_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PONDERV2_HIP_H */
