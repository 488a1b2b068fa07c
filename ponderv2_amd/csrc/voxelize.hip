// Train-mode GridSample on the device: one representative point per occupied voxel, voxels in
// ascending UNSIGNED key order - what the reference's DataLoader workers compute with numpy
// (ponder/datasets/transform.py:1103-1145 GridSample.__call__: floor(coord / grid_size) - min,
// fnv_hash_vec :1180-1197 / ravel_hash_vec :1199-1213, argsort, unique, one random member per voxel).
//
// The torch composition this replaces sorted all M raw points twice (two 64-bit ATen radix sorts
// of ~1 M keys per step).  Here the points are de-duplicated by a hash table first - one atomicCAS
// per point, keyed by the reference's OWN 64-bit key, so voxels whose keys collide merge exactly as
// np.unique merges them - and only the ~N unique keys (N ~ M / 20) are sorted:
//   voxel_bounds    per-axis minimum (and maximum, for the ravel key) of floor(coord / grid_size)
//   voxel_insert    grid coordinates, key, table slot and member count per point
//   voxel_compact   occupied slots -> (key, slot) list                       [n_vox read by the host]
//   rocPRIM         radix sort of the n_vox (key, slot) pairs, exclusive scan of the member counts
//   voxel_rank / voxel_members   slot -> rank; members of each voxel (CSR, filled through a cursor)
//   voxel_pick      the (pick % count)-th member of each voxel IN POINT-INDEX ORDER (what a stable
//                   argsort leaves in that position), its index and its grid coordinates
// HBM / latency-bound integer work: ~60 B per raw point in all.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.h"

namespace {

constexpr uint64_t kEmpty = ~0ull;
constexpr uint64_t kFnvOffset = 14695981039346656037ull, kFnvPrime = 1099511628211ull;

template <typename T>
__device__ __forceinline__ int64_t cell_of(const T* __restrict__ coord, int64_t p, int axis,
                                           double grid_size) {
  // the host transform divides a float64 copy of the coordinates by a float64 grid size
  return (int64_t)floor((double)coord[p * 3 + axis] / grid_size);
}

__global__ void voxel_bounds_init_kernel(long long* __restrict__ mm) {
  if (threadIdx.x < 3) mm[threadIdx.x] = 0x7fffffffffffffffll;
  else if (threadIdx.x < 6) mm[threadIdx.x] = -0x7fffffffffffffffll - 1;
}

template <typename T>
__global__ __launch_bounds__(256) void voxel_bounds_kernel(const T* __restrict__ coord, int64_t n,
                                                           double grid_size,
                                                           long long* __restrict__ mm) {
  long long lo[3] = {0x7fffffffffffffffll, 0x7fffffffffffffffll, 0x7fffffffffffffffll};
  long long hi[3] = {-0x7fffffffffffffffll - 1, -0x7fffffffffffffffll - 1, -0x7fffffffffffffffll - 1};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const long long c = cell_of(coord, p, a, grid_size);
      lo[a] = c < lo[a] ? c : lo[a];
      hi[a] = c > hi[a] ? c : hi[a];
    }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    for (int o = 32; o > 0; o >>= 1) {
      const long long l2 = __shfl_xor(lo[a], o), h2 = __shfl_xor(hi[a], o);
      lo[a] = l2 < lo[a] ? l2 : lo[a];
      hi[a] = h2 > hi[a] ? h2 : hi[a];
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(&mm[a], lo[a]);
      atomicMax(&mm[3 + a], hi[a]);
    }
  }
}

// per point: grid - min, key, table slot (linear probing on the key), member count of the slot
template <typename T>
__global__ __launch_bounds__(256) void voxel_insert_kernel(
    const T* __restrict__ coord, int64_t n, double grid_size, const long long* __restrict__ mm,
    int ravel, unsigned long long* __restrict__ tkeys, int32_t* __restrict__ tcount, uint64_t mask,
    int32_t* __restrict__ grid, int32_t* __restrict__ slot_of) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
    uint64_t g[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      g[a] = (uint64_t)(cell_of(coord, p, a, grid_size) - mm[a]);
      grid[p * 3 + a] = (int32_t)g[a];
    }
    uint64_t key;
    if (ravel) {  // ((x * ey) + y) * ez + z, extents = max - min + 1 (transform.py:1199-1213)
      const uint64_t ey = (uint64_t)(mm[4] - mm[1] + 1), ez = (uint64_t)(mm[5] - mm[2] + 1);
      key = (g[0] * ey + g[1]) * ez + g[2];
    } else {      // FNV-1a over the three coordinates as uint64 (transform.py:1180-1197)
      key = kFnvOffset;
#pragma unroll
      for (int a = 0; a < 3; ++a) key = (key * kFnvPrime) ^ g[a];
    }
    // (a key equal to the empty marker, probability 2^-64, is stored one below it)
    const unsigned long long k = key == kEmpty ? kEmpty - 1 : key;
    uint64_t s = (k * 0x9E3779B97F4A7C15ull) >> 20 & mask;
    for (;;) {
      const unsigned long long seen = atomicCAS(&tkeys[s], (unsigned long long)kEmpty, k);
      if (seen == kEmpty || seen == k) break;
      s = (s + 1) & mask;
    }
    atomicAdd(&tcount[s], 1);
    slot_of[p] = (int32_t)s;
  }
}

__global__ __launch_bounds__(256) void voxel_compact_kernel(
    const unsigned long long* __restrict__ tkeys, int64_t m, unsigned long long* __restrict__ ukeys,
    int32_t* __restrict__ uslot, int32_t* __restrict__ n_vox) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < m; s += stride) {
    const unsigned long long k = tkeys[s];
    if (k != kEmpty) {
      const int at = atomicAdd(n_vox, 1);   // (order irrelevant: the list is sorted next)
      ukeys[at] = k;
      uslot[at] = (int32_t)s;
    }
  }
}

// sorted slot list -> rank of each slot, member count per voxel (in rank order), the largest count
__global__ __launch_bounds__(256) void voxel_rank_kernel(const int32_t* __restrict__ sslot, int n_vox,
                                                         const int32_t* __restrict__ tcount,
                                                         int32_t* __restrict__ rank_of_slot,
                                                         int32_t* __restrict__ count,
                                                         int32_t* __restrict__ max_count) {
  int mx = 0;
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < n_vox; v += gridDim.x * blockDim.x) {
    const int s = sslot[v];
    rank_of_slot[s] = v;
    const int c = tcount[s];
    count[v] = c;
    mx = c > mx ? c : mx;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const int m2 = __shfl_xor(mx, o);
    mx = m2 > mx ? m2 : mx;
  }
  if ((threadIdx.x & 63) == 0 && mx > 0) atomicMax(max_count, mx);
}

__global__ __launch_bounds__(256) void voxel_members_kernel(
    const int32_t* __restrict__ slot_of, int64_t n, const int32_t* __restrict__ rank_of_slot,
    const int32_t* __restrict__ start, int32_t* __restrict__ cursor, int32_t* __restrict__ members) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
    const int v = rank_of_slot[slot_of[p]];
    members[start[v] + atomicAdd(&cursor[v], 1)] = (int32_t)p;
  }
}

// One thread per voxel: the member with exactly t = (pick mod max_count) mod count smaller members,
// i.e. the t-th in point-index order.  Counts are ~20 (raw points per 2 cm voxel): the quadratic
// scan is ~400 comparisons.  `draws` are either the caller's integers (the reference's
// np.random.randint(0, count.max(), n_vox)) or raw 31-bit random numbers (reduced here).
__global__ __launch_bounds__(256) void voxel_pick_kernel(
    const int32_t* __restrict__ members, const int32_t* __restrict__ start,
    const int32_t* __restrict__ count, int n_vox, const int64_t* __restrict__ draws,
    const int32_t* __restrict__ max_count, const int32_t* __restrict__ grid,
    int64_t* __restrict__ idx_unique, int64_t* __restrict__ grid_coord) {
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < n_vox; v += gridDim.x * blockDim.x) {
    const int c = count[v], s0 = start[v];
    const int t = (int)((draws[v] % (int64_t)max_count[0]) % c);
    int chosen = members[s0];
    for (int a = 0; a < c; ++a) {
      const int ma = members[s0 + a];
      int smaller = 0;
      for (int b = 0; b < c; ++b) smaller += members[s0 + b] < ma;
      if (smaller == t) {
        chosen = ma;
        break;
      }
    }
    idx_unique[v] = chosen;
#pragma unroll
    for (int a = 0; a < 3; ++a) grid_coord[(int64_t)v * 3 + a] = grid[(int64_t)chosen * 3 + a];
  }
}

template <typename T>
int stage1(const T* coord, int64_t n, double grid_size, int ravel, long long* minmax,
           unsigned long long* tkeys, int32_t* tcount, int64_t table_size, int32_t* grid,
           int32_t* slot_of, unsigned long long* ukeys, int32_t* uslot, int32_t* n_vox, hipStream_t s) {
  hipLaunchKernelGGL(voxel_bounds_init_kernel, dim3(1), dim3(64), 0, s, minmax);
  hipLaunchKernelGGL((voxel_bounds_kernel<T>), dim3(pv2::grid_for(n, 256)), dim3(256), 0, s, coord, n,
                     grid_size, minmax);
  hipLaunchKernelGGL((voxel_insert_kernel<T>), dim3(pv2::grid_for(n, 256)), dim3(256), 0, s, coord, n,
                     grid_size, minmax, ravel, tkeys, tcount, (uint64_t)(table_size - 1), grid, slot_of);
  hipLaunchKernelGGL(voxel_compact_kernel, dim3(pv2::grid_for(table_size, 256)), dim3(256), 0, s, tkeys,
                     table_size, ukeys, uslot, n_vox);
  return pv2::check_launch("voxelize_stage1");
}

}  // namespace

extern "C" {

// table_size: a power of two >= 2 n.  Inputs that must be prepared by the caller: tkeys filled with
// 0xff bytes, tcount and n_vox zeroed.
int pv2_voxelize_stage1(const void* coord, int coord_is_f64, int64_t n, double grid_size, int ravel_hash,
                        int64_t* minmax, uint64_t* table_keys, int32_t* table_count, int64_t table_size,
                        int32_t* grid, int32_t* slot_of, uint64_t* uniq_keys, int32_t* uniq_slot,
                        int32_t* n_vox, pv2_stream_t stream) {
  PV2_REQUIRE(coord != nullptr && n > 0 && grid_size > 0.0, "voxelize_stage1: bad input");
  PV2_REQUIRE(table_size >= 2 * n && (table_size & (table_size - 1)) == 0,
              "voxelize_stage1: table_size must be a power of two >= 2 n");
  hipStream_t s = (hipStream_t)stream;
  if (coord_is_f64)
    return stage1((const double*)coord, n, grid_size, ravel_hash, (long long*)minmax,
                  (unsigned long long*)table_keys, table_count, table_size, grid, slot_of,
                  (unsigned long long*)uniq_keys, uniq_slot, n_vox, s);
  return stage1((const float*)coord, n, grid_size, ravel_hash, (long long*)minmax,
                (unsigned long long*)table_keys, table_count, table_size, grid, slot_of,
                (unsigned long long*)uniq_keys, uniq_slot, n_vox, s);
}

size_t pv2_voxelize_workspace_bytes(int64_t n_vox) {
  size_t a = 0, b = 0;
  (void)rocprim::radix_sort_pairs(nullptr, a, (uint64_t*)nullptr, (uint64_t*)nullptr, (int32_t*)nullptr,
                                  (int32_t*)nullptr, (size_t)n_vox);
  (void)rocprim::exclusive_scan(nullptr, b, (int32_t*)nullptr, (int32_t*)nullptr, 0, (size_t)n_vox,
                                rocprim::plus<int32_t>());
  return (a > b ? a : b) + 256;
}

// n_vox: the count stage 1 left in *n_vox, read by the caller.  sorted_keys / sorted_slot / rank_of_slot
// [table_size] / count / start / cursor (zeroed) [n_vox] / members [n] / max_count (zeroed) are scratch;
// draws [n_vox] int64.  Outputs: idx_unique [n_vox], grid_coord [n_vox, 3] (int64).
int pv2_voxelize_stage2(int64_t n, int64_t n_vox, const uint64_t* uniq_keys, const int32_t* uniq_slot,
                        const int32_t* table_count, const int32_t* slot_of, const int32_t* grid,
                        const int64_t* draws, uint64_t* sorted_keys, int32_t* sorted_slot,
                        int32_t* rank_of_slot, int32_t* count, int32_t* start, int32_t* cursor,
                        int32_t* members, int32_t* max_count, void* workspace, size_t workspace_bytes,
                        int64_t* idx_unique, int64_t* grid_coord, pv2_stream_t stream) {
  PV2_REQUIRE(n > 0 && n_vox > 0 && n_vox <= n, "voxelize_stage2: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  size_t need = 0;
  (void)rocprim::radix_sort_pairs(nullptr, need, uniq_keys, sorted_keys, uniq_slot, sorted_slot, (size_t)n_vox);
  PV2_REQUIRE(need <= workspace_bytes, "voxelize_stage2: workspace too small (sort)");
  hipError_t e = rocprim::radix_sort_pairs(workspace, need, uniq_keys, sorted_keys, uniq_slot, sorted_slot,
                                           (size_t)n_vox, 0, 64, s);
  if (e != hipSuccess) return pv2::hip_status(e);
  hipLaunchKernelGGL(voxel_rank_kernel, dim3(pv2::grid_for(n_vox, 256)), dim3(256), 0, s, sorted_slot,
                     (int)n_vox, table_count, rank_of_slot, count, max_count);
  need = 0;
  (void)rocprim::exclusive_scan(nullptr, need, count, start, 0, (size_t)n_vox, rocprim::plus<int32_t>(), s);
  PV2_REQUIRE(need <= workspace_bytes, "voxelize_stage2: workspace too small (scan)");
  e = rocprim::exclusive_scan(workspace, need, count, start, 0, (size_t)n_vox, rocprim::plus<int32_t>(), s);
  if (e != hipSuccess) return pv2::hip_status(e);
  hipLaunchKernelGGL(voxel_members_kernel, dim3(pv2::grid_for(n, 256)), dim3(256), 0, s, slot_of, n,
                     rank_of_slot, start, cursor, members);
  hipLaunchKernelGGL(voxel_pick_kernel, dim3(pv2::grid_for(n_vox, 256)), dim3(256), 0, s, members, start,
                     count, (int)n_vox, draws, max_count, grid, idx_unique, grid_coord);
  return pv2::check_launch("voxelize_stage2");
}

}  // extern "C"
