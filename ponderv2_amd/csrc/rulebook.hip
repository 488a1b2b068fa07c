// Rulebook (indice-pair) generation for submanifold / strided sparse 3-D convolutions on gfx950.
//
// Stands in for spconv 2.x's hash table + generate_subm_conv_inds / generate_conv_inds that the
// reference reaches through ponder/models/sparse_unet/spconv_unet_v1m1_base.py:112 (stem k5),
// :47/:58 (k3 blocks), :135 (k2 s2 down) and :171 (inverse).  The design is ours:
//   * a 64-bit packed (b,x,y,z) key in an open-addressing table (one atomicCAS per voxel);
//   * a dense [K^3, N] neighbour table written with the voxel index on the fast axis, so every
//     table write and the later ordered compaction are fully coalesced;
//   * ordered compaction by a 3-pass scan whose blocks never straddle a kernel offset, which
//     yields the canonical (offset, output row) pair order with no sort;
//   * strided convs get their unique output set from a device radix sort of the packed keys
//     (rocPRIM), so outputs come out in (b,x,y,z) order - again canonical, no tie-breaking.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

#include "common.h"

namespace {

constexpr uint64_t kEmptyKey = ~0ull;
constexpr int kBias = 16;  // room for negative neighbour coordinates

__device__ __forceinline__ uint64_t pack_key(int b, int x, int y, int z) {
  return ((uint64_t)(uint32_t)b << 48) | ((uint64_t)(uint32_t)(x + kBias) << 32) |
         ((uint64_t)(uint32_t)(y + kBias) << 16) | (uint64_t)(uint32_t)(z + kBias);
}

__device__ __forceinline__ uint64_t mix64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}

__global__ void hash_init_kernel(uint64_t* keys, int32_t* vals, int64_t size) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < size; i += stride) {
    keys[i] = kEmptyKey;
    vals[i] = 0x7fffffff;
  }
}

__global__ void hash_insert_kernel(const int4* __restrict__ coords, int64_t n,
                                   unsigned long long* keys, int32_t* vals, uint64_t mask) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int4 c = coords[i];
    if (c.x < 0) continue;  // padding row of a capacity-sized coordinate array
    uint64_t key = pack_key(c.x, c.y, c.z, c.w);
    uint64_t slot = mix64(key) & mask;
    for (;;) {
      unsigned long long prev = atomicCAS(&keys[slot], (unsigned long long)kEmptyKey,
                                          (unsigned long long)key);
      if (prev == kEmptyKey || prev == key) {
        atomicMin(&vals[slot], (int32_t)i);
        break;
      }
      slot = (slot + 1) & mask;
    }
  }
}

__device__ __forceinline__ int32_t hash_lookup(const uint64_t* __restrict__ keys,
                                               const int32_t* __restrict__ vals, uint64_t mask,
                                               uint64_t key) {
  uint64_t slot = mix64(key) & mask;
  for (;;) {
    uint64_t cur = keys[slot];
    if (cur == key) return vals[slot];
    if (cur == kEmptyKey) return -1;
    slot = (slot + 1) & mask;
  }
}

// One thread per (voxel, dx, dy) column of the window, looping over dz: the hash probes of a voxel
// are independent, so spreading them over ksize^2 threads (25 for the 5x5x5 stem) hides their
// latency; nbr[k*n + i] is still written coalesced along i (consecutive threads = consecutive i).
__global__ void subm_table_kernel(const int4* __restrict__ coords, int64_t n, int ksize,
                                  const uint64_t* __restrict__ keys,
                                  const int32_t* __restrict__ vals, uint64_t mask,
                                  int32_t* __restrict__ nbr) {
  const int r = ksize / 2;
  const int64_t total = n * ksize * ksize;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t i = e % n;
    const int col = (int)(e / n);
    const int dx = col / ksize - r, dy = col % ksize - r;
    const int4 c = coords[i];
    const int x = c.y + dx, y = c.z + dy;
    int k = col * ksize;
    for (int dz = -r; dz <= r; ++dz, ++k) {
      const int z = c.w + dz;
      int32_t j = -1;
      if (c.x < 0) {
        // padding row: no neighbours, not even itself
      } else if (dx == 0 && dy == 0 && dz == 0) {
        j = (int32_t)i;  // centre tap: the voxel itself (keeps duplicates self-consistent)
      } else if (x >= 0 && y >= 0 && z >= 0) {
        j = hash_lookup(keys, vals, mask, pack_key(c.x, x, y, z));
      }
      nbr[(int64_t)k * n + i] = j;
    }
  }
}

// ---------------------------------------------------------------- ordered compaction
// grid = (nchunks, K); block = 256 threads; a block owns PV2_SCAN_CHUNK consecutive columns.
__global__ void table_count_kernel(const int32_t* __restrict__ tbl, int64_t n,
                                   const int32_t* __restrict__ n_rows_dev,
                                   int32_t* __restrict__ block_sums) {
  __shared__ int wsum[4];
  const int64_t nrows = n_rows_dev ? (int64_t)*n_rows_dev : n;
  const int k = blockIdx.y;
  const int64_t base = (int64_t)blockIdx.x * PV2_SCAN_CHUNK;
  int cnt = 0;
  for (int t = threadIdx.x; t < PV2_SCAN_CHUNK; t += 256) {
    int64_t col = base + t;
    if (col < nrows && tbl[(int64_t)k * n + col] >= 0) ++cnt;
  }
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0)
    block_sums[(int64_t)k * gridDim.x + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// Single block: in-place exclusive scan of block_sums[K*nchunks]; kstart[k] = prefix at k*nchunks.
__global__ void table_scan_kernel(int32_t* __restrict__ block_sums, int K, int nchunks,
                                  int32_t* __restrict__ kstart) {
  __shared__ int wtot[16];
  __shared__ int carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t total = (int64_t)K * nchunks;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < total; base += 1024) {
    int64_t idx = base + tid;
    int v = idx < total ? block_sums[idx] : 0;
    int incl = v;
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wtot[w];
    int carry = carry_s;
    int excl = carry + woff + incl - v;
    if (idx < total) {
      block_sums[idx] = excl;
      if (idx % nchunks == 0) kstart[idx / nchunks] = excl;
    }
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (tid == 0) kstart[K] = carry_s;
}

// out[s][k] = sum_{j<k} ceil((kstart[j+1]-kstart[j]) / tile[s]): the workgroup-tile prefixes the
// conv kernels search, for up to 4 tile sizes at once (one thread each; K is tiny).
__global__ void tile_prefix_kernel(const int32_t* __restrict__ kstart, int K, int4 tiles, int n,
                                   int32_t* __restrict__ out) {
  const int s = threadIdx.x;
  if (s >= n) return;
  const int tile = s == 0 ? tiles.x : s == 1 ? tiles.y : s == 2 ? tiles.z : tiles.w;
  int acc = 0;
  int32_t* o = out + (int64_t)s * (K + 1);
  for (int k = 0; k < K; ++k) {
    o[k] = acc;
    acc += (kstart[k + 1] - kstart[k] + tile - 1) / tile;
  }
  o[K] = acc;
}

__global__ void table_compact_kernel(const int32_t* __restrict__ tbl, int64_t n,
                                     const int32_t* __restrict__ n_rows_dev,
                                     const int32_t* __restrict__ block_excl,
                                     int32_t* __restrict__ pair_other,
                                     int32_t* __restrict__ pair_row) {
  __shared__ int wcnt[4];
  const int64_t nrows = n_rows_dev ? (int64_t)*n_rows_dev : n;
  const int k = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * PV2_SCAN_CHUNK;
  int running = block_excl[(int64_t)k * gridDim.x + blockIdx.x];
  for (int sub = 0; sub < PV2_SCAN_CHUNK; sub += 256) {
    int64_t col = base + sub + threadIdx.x;
    int32_t v = -1;
    if (col < nrows) v = tbl[(int64_t)k * n + col];
    const bool valid = v >= 0;
    unsigned long long bal = __ballot(valid);
    int prefix = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wcnt[wave] = __popcll(bal);
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wcnt[w];
    int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    if (valid) {
      int pos = running + woff + prefix;
      pair_other[pos] = v;
      pair_row[pos] = (int32_t)col;
    }
    running += tot;
    __syncthreads();
  }
}

// ---------------------------------------------------------------- strided conv (kernel == stride)
__global__ void down_keys_kernel(const int4* __restrict__ coords, int64_t n, int s, int ox, int oy,
                                 int oz, uint64_t* __restrict__ keys) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int4 c = coords[i];
    int x = c.y / s, y = c.z / s, z = c.w / s;
    keys[i] = (c.x >= 0 && x < ox && y < oy && z < oz) ? pack_key(c.x, x, y, z) : kEmptyKey;
  }
}

// After rocprim::unique: drop the sentinel (it sorts last) and decode coordinates.
__global__ void down_fix_count_kernel(const uint64_t* __restrict__ uniq, int32_t* n_out) {
  int c = *n_out;
  if (c > 0 && uniq[c - 1] == kEmptyKey) *n_out = c - 1;
}

__global__ void down_decode_kernel(const uint64_t* __restrict__ uniq,
                                   const int32_t* __restrict__ n_out, int64_t n,
                                   int4* __restrict__ out_coords) {
  const int64_t cnt = *n_out;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int4 c = make_int4(-1, -1, -1, -1);
    if (i < cnt) {
      uint64_t key = uniq[i];
      c.x = (int)(key >> 48);
      c.y = (int)((key >> 32) & 0xffff) - kBias;
      c.z = (int)((key >> 16) & 0xffff) - kBias;
      c.w = (int)(key & 0xffff) - kBias;
    }
    out_coords[i] = c;
  }
}

__global__ void fill_i32_kernel(int32_t* p, int64_t n, int32_t v) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

__global__ void down_table_kernel(const int4* __restrict__ coords, int64_t n, int s, int ox,
                                  int oy, int oz, const uint64_t* __restrict__ uniq,
                                  const int32_t* __restrict__ n_out, int32_t* __restrict__ tbl,
                                  int64_t n_cap) {
  const int64_t cnt = *n_out;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int4 c = coords[i];
    int x = c.y / s, y = c.z / s, z = c.w / s;
    if (c.x < 0 || !(x < ox && y < oy && z < oz)) continue;
    uint64_t key = pack_key(c.x, x, y, z);
    int64_t lo = 0, hi = cnt;  // lower_bound
    while (lo < hi) {
      int64_t mid = (lo + hi) >> 1;
      if (uniq[mid] < key) lo = mid + 1; else hi = mid;
    }
    int k = ((c.y - x * s) * s + (c.z - y * s)) * s + (c.w - z * s);
    tbl[(int64_t)k * n_cap + lo] = (int32_t)i;
  }
}

// out[k][tbl[k][j]] = j for every valid entry: turns "which input feeds output o under offset k"
// into "which output does input j feed under offset k" (and back).  Every (k, value) pair occurs
// at most once in the tables of this file, so the scatter is race-free and deterministic.
__global__ void table_invert_kernel(const int32_t* __restrict__ tbl, int K, int64_t n_cols,
                                    int64_t stride_in, const int32_t* __restrict__ n_cols_dev,
                                    int32_t* __restrict__ out, int64_t stride_out) {
  const int64_t cols = n_cols_dev ? min((int64_t)*n_cols_dev, n_cols) : n_cols;
  const int64_t total = (int64_t)K * cols;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t k = e / cols, j = e % cols;
    const int32_t v = tbl[k * stride_in + j];
    if (v >= 0) out[k * stride_out + v] = (int32_t)j;
  }
}

// mask[i] = bit k set when tbl[k][i] >= 0 (K <= 64): the key the output-stationary conv sorts its
// rows by, so that the rows of one tile use the same few kernel offsets.
__global__ void table_masks_kernel(const int32_t* __restrict__ tbl, int K, int64_t n_cols,
                                   int64_t stride_in, const int32_t* __restrict__ n_cols_dev,
                                   int64_t* __restrict__ mask) {
  const int64_t cols = n_cols_dev ? min((int64_t)*n_cols_dev, n_cols) : n_cols;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_cols; i += stride) {
    uint64_t m = 0;
    if (i < cols)
      for (int k = 0; k < K; ++k) m |= (uint64_t)(tbl[(int64_t)k * stride_in + i] >= 0) << k;
    else
      m = ~0ull >> 1;  // columns past the valid count sort last
    mask[i] = (int64_t)m;
  }
}

}  // namespace

extern "C" {

int pv2_table_invert(const int32_t* tbl, int K, int64_t n_cols, int64_t stride_in,
                     const int32_t* n_cols_dev, int32_t* out, int64_t n_out_cols,
                     pv2_stream_t stream) {
  PV2_REQUIRE(K >= 1 && n_cols >= 0 && n_out_cols >= 0 && stride_in >= n_cols,
              "pv2_table_invert: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  if (n_out_cols > 0)
    hipLaunchKernelGGL(fill_i32_kernel, dim3(pv2::grid_for(K * n_out_cols, 256)), dim3(256), 0, s,
                       out, (int64_t)K * n_out_cols, -1);
  if (n_cols > 0)
    hipLaunchKernelGGL(table_invert_kernel, dim3(pv2::grid_for(K * n_cols, 256)), dim3(256), 0, s,
                       tbl, K, n_cols, stride_in, n_cols_dev, out, n_out_cols);
  return pv2::check_launch("table_invert");
}

int pv2_table_masks(const int32_t* tbl, int K, int64_t n_cols, int64_t stride_in,
                    const int32_t* n_cols_dev, int64_t* mask, pv2_stream_t stream) {
  PV2_REQUIRE(K >= 1 && K <= 63, "pv2_table_masks: 1 <= K <= 63");
  if (n_cols == 0) return PV2_OK;
  hipLaunchKernelGGL(table_masks_kernel, dim3(pv2::grid_for(n_cols, 256)), dim3(256), 0,
                     (hipStream_t)stream, tbl, K, n_cols, stride_in, n_cols_dev, mask);
  return pv2::check_launch("table_masks");
}

int pv2_hash_build(const int32_t* coords, int64_t n, uint64_t* table_keys, int32_t* table_vals,
                   int64_t table_size, pv2_stream_t stream) {
  PV2_REQUIRE(table_size > 0 && (table_size & (table_size - 1)) == 0,
              "pv2_hash_build: table_size must be a power of two");
  PV2_REQUIRE(table_size >= 2 * n, "pv2_hash_build: table_size must be >= 2*n");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(hash_init_kernel, dim3(pv2::grid_for(table_size, 256)), dim3(256), 0, s,
                     table_keys, table_vals, table_size);
  if (n > 0)
    hipLaunchKernelGGL(hash_insert_kernel, dim3(pv2::grid_for(n, 256)), dim3(256), 0, s,
                       (const int4*)coords, n, (unsigned long long*)table_keys, table_vals,
                       (uint64_t)(table_size - 1));
  return pv2::check_launch("hash_build");
}

int pv2_subm_neighbor_table(const int32_t* coords, int64_t n, int ksize,
                            const uint64_t* table_keys, const int32_t* table_vals,
                            int64_t table_size, int32_t* nbr, pv2_stream_t stream) {
  PV2_REQUIRE(ksize >= 1 && (ksize & 1) && ksize <= 2 * kBias + 1,
              "pv2_subm_neighbor_table: ksize must be odd and <= 33");
  if (n == 0) return PV2_OK;
  hipLaunchKernelGGL(subm_table_kernel, dim3(pv2::grid_for(n * ksize * ksize, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const int4*)coords, n, ksize, table_keys, table_vals,
                     (uint64_t)(table_size - 1), nbr);
  return pv2::check_launch("subm_table");
}

size_t pv2_downsample_workspace_bytes(int64_t n) {
  size_t a = 0, b = 0;
  if (n <= 0) return 256;
  (void)rocprim::radix_sort_keys(nullptr, a, (uint64_t*)nullptr, (uint64_t*)nullptr, (size_t)n);
  (void)rocprim::unique(nullptr, b, (uint64_t*)nullptr, (uint64_t*)nullptr, (int32_t*)nullptr,
                  (size_t)n, rocprim::equal_to<uint64_t>());
  size_t m = a > b ? a : b;
  return m + 256;
}

int pv2_downsample_unique(const int32_t* coords, int64_t n, int stride, const int32_t* out_shape,
                          uint64_t* keys_tmp, uint64_t* keys_sorted, int32_t* out_coords,
                          int32_t* n_out, void* workspace, size_t workspace_bytes,
                          pv2_stream_t stream) {
  PV2_REQUIRE(stride >= 1, "pv2_downsample_unique: stride must be >= 1");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) return pv2::zero_words(n_out, 1, s);
  // raw keys -> keys_sorted (as scratch), sorted -> keys_tmp, unique -> keys_sorted.
  hipLaunchKernelGGL(down_keys_kernel, dim3(pv2::grid_for(n, 256)), dim3(256), 0, s,
                     (const int4*)coords, n, stride, out_shape[0], out_shape[1], out_shape[2],
                     keys_sorted);
  size_t need = 0;
  (void)rocprim::radix_sort_keys(nullptr, need, keys_sorted, keys_tmp, (size_t)n);
  if (need > workspace_bytes) {
    pv2::set_error("pv2_downsample_unique: workspace too small (sort)");
    return PV2_E_WORKSPACE;
  }
  hipError_t e = rocprim::radix_sort_keys(workspace, need, keys_sorted, keys_tmp, (size_t)n, 0,
                                          64, s);
  if (e != hipSuccess) return pv2::hip_status(e);
  need = 0;
  (void)rocprim::unique(nullptr, need, keys_tmp, keys_sorted, n_out, (size_t)n,
                  rocprim::equal_to<uint64_t>(), s);
  if (need > workspace_bytes) {
    pv2::set_error("pv2_downsample_unique: workspace too small (unique)");
    return PV2_E_WORKSPACE;
  }
  e = rocprim::unique(workspace, need, keys_tmp, keys_sorted, n_out, (size_t)n,
                      rocprim::equal_to<uint64_t>(), s);
  if (e != hipSuccess) return pv2::hip_status(e);
  hipLaunchKernelGGL(down_fix_count_kernel, dim3(1), dim3(1), 0, s, keys_sorted, n_out);
  hipLaunchKernelGGL(down_decode_kernel, dim3(pv2::grid_for(n, 256)), dim3(256), 0, s,
                     keys_sorted, n_out, n, (int4*)out_coords);
  return pv2::check_launch("downsample_unique");
}

int pv2_downsample_table(const int32_t* coords, int64_t n, int stride, const int32_t* out_shape,
                         const uint64_t* keys_sorted, const int32_t* n_out, int32_t* tbl,
                         int64_t n_cap, pv2_stream_t stream) {
  PV2_REQUIRE(stride >= 1 && stride <= 8, "pv2_downsample_table: stride must be in [1,8]");
  hipStream_t s = (hipStream_t)stream;
  const int64_t K = (int64_t)stride * stride * stride;
  if (n_cap == 0) return PV2_OK;
  hipLaunchKernelGGL(fill_i32_kernel, dim3(pv2::grid_for(K * n_cap, 256)), dim3(256), 0, s, tbl,
                     K * n_cap, -1);
  if (n > 0)
    hipLaunchKernelGGL(down_table_kernel, dim3(pv2::grid_for(n, 256)), dim3(256), 0, s,
                       (const int4*)coords, n, stride, out_shape[0], out_shape[1], out_shape[2],
                       keys_sorted, n_out, tbl, n_cap);
  return pv2::check_launch("downsample_table");
}

int pv2_table_count(const int32_t* tbl, int K, int64_t n, const int32_t* n_rows_dev,
                    int32_t* block_sums, int32_t* kstart, pv2_stream_t stream) {
  PV2_REQUIRE(K >= 1, "pv2_table_count: K must be >= 1");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) return pv2::zero_words(kstart, K + 1, s);
  const int nchunks = (int)((n + PV2_SCAN_CHUNK - 1) / PV2_SCAN_CHUNK);
  hipLaunchKernelGGL(table_count_kernel, dim3(nchunks, K), dim3(256), 0, s, tbl, n, n_rows_dev,
                     block_sums);
  hipLaunchKernelGGL(table_scan_kernel, dim3(1), dim3(1024), 0, s, block_sums, K, nchunks,
                     kstart);
  return pv2::check_launch("table_count");
}

int pv2_tile_prefix(const int32_t* kstart, int K, const int32_t* tile_sizes, int n_sizes,
                    int32_t* out, pv2_stream_t stream) {
  PV2_REQUIRE(K >= 1 && n_sizes >= 1 && n_sizes <= 4, "pv2_tile_prefix: 1..4 tile sizes");
  int4 t = make_int4(1, 1, 1, 1);
  int* tp = &t.x;
  for (int i = 0; i < n_sizes; ++i) {
    PV2_REQUIRE(tile_sizes[i] >= 1, "pv2_tile_prefix: tile sizes must be positive");
    tp[i] = tile_sizes[i];
  }
  hipLaunchKernelGGL(tile_prefix_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, kstart, K, t,
                     n_sizes, out);
  return pv2::check_launch("tile_prefix");
}

int pv2_table_compact(const int32_t* tbl, int K, int64_t n, const int32_t* n_rows_dev,
                      const int32_t* block_sums, int32_t* pair_other, int32_t* pair_row,
                      pv2_stream_t stream) {
  if (n == 0) return PV2_OK;
  const int nchunks = (int)((n + PV2_SCAN_CHUNK - 1) / PV2_SCAN_CHUNK);
  hipLaunchKernelGGL(table_compact_kernel, dim3(nchunks, K), dim3(256), 0, (hipStream_t)stream,
                     tbl, n, n_rows_dev, block_sums, pair_other, pair_row);
  return pv2::check_launch("table_compact");
}

}  // extern "C"
