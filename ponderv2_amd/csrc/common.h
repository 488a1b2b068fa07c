// Shared helpers for the gfx950 kernels of libponderv2_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ponderv2_hip.h"

namespace pv2 {

void set_error(const char* msg);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(hipGetErrorString(e));
    (void)what;
    return (int)e;
  }
  return PV2_OK;
}

inline int hip_status(hipError_t e) {
  if (e != hipSuccess) {
    set_error(hipGetErrorString(e));
    return (int)e;
  }
  return PV2_OK;
}

constexpr int kWave = 64;  // gfx950 wavefront

// Zero `n32` 32-bit words with a kernel launch.  Used instead of hipMemsetAsync for the small
// scratch buffers: a kernel node is what a hipGraph capture of the caller's stream records
// reliably (the render head of a training step is replayed as a graph).
__global__ void zero_words_kernel(uint32_t* p, int64_t n32);
int zero_words(void* p, int64_t n32, hipStream_t s);

// Grid size for a grid-stride elementwise launch: enough blocks to fill 256 CUs x 8, no more.
inline int grid_for(int64_t work_items, int block) {
  int64_t b = (work_items + block - 1) / block;
  if (b < 1) b = 1;
  if (b > 256 * 8) b = 256 * 8;
  return (int)b;
}

// ---- internal (C++ linkage) entry points shared between translation units; the C ABI wraps them
// sparse_conv.hip
int spconv_products(bool trans, const float* in_feat, int c_in, const float* weight, int K,
                    int c_out, const int32_t* pair_in, const int32_t* kstart,
                    const int32_t* tile_start, int64_t n_tiles, float* prod, hipStream_t s);
int spconv_wgrad(const float* in_feat, int64_t n_in, int c_in, const float* dout, int64_t n_out,
                 int c_out, int K, const int32_t* pair_in, const int32_t* pair_out,
                 const int32_t* kstart, const int32_t* tile_start, int tile_pairs, int64_t n_tiles,
                 float* dweight, float* part, hipStream_t s);
// sparse_conv_pr.hip
int fork_event_for(hipEvent_t* ev);   // a reusable event that orders a side stream behind the caller's
// The conv + BatchNorm unit whose OUTPUT is the activation a grad-input launch completes the gradient of.
struct BnProducer {
  const float* y_conv;
  const float* out_or_null;   // the activation when a ReLU follows the BatchNorm
  const float* mean_invstd;
  float* gsum;                // [2 c]: sum g, sum g * xhat
};
int convbn_backward(const pv2_conv_geom* g, const float* grad_out, const float* x, int c_in,
                    const float* weight, int c_out, const float* y_conv, const float* out_or_null,
                    const float* mean_invstd, const float* bn_weight, float* prod_ws,
                    float* stats_ws, float* gsum, float* dy, float* dres_or_null, float* dx_or_null,
                    int dx_accumulate, float* dweight_or_null, float* part_ws, hipStream_t s,
                    hipStream_t side, int bn_sums_ready, const BnProducer* dx_producer,
                    int* dx_sums_done);
// rownorm.hip: the statistics kernels' partial-sum geometry (blocks <= 1024, rows per block) and the
// second half of the fused BatchNorm forward - combine the per-block partial sums (written by
// col_partials or by row_reduce_kernel's epilogue, shifted by row 0 of x) and apply.
void bn_partial_geometry(int64_t n, int c, int* blocks, int64_t* rows_per_block);
int bn_forward_from_partials(const float* x, int64_t n, int c, const float* partial, int blocks,
                             const float* weight, const float* bias, const float* residual,
                             int relu, float eps, float momentum, float* running_mean,
                             float* running_var, float* mean_invstd, float* y, hipStream_t s);

int bn_apply(const float* x, int64_t n, int c, const float* mean_invstd, const float* weight,
             const float* bias, const float* residual, int relu, float* y, hipStream_t s);
int bn_backward_combine(const float* partial, int blocks, int c, float* gsum, hipStream_t s);
int bn_backward_apply(const float* dy, const float* x, const float* y_or_null, const float* mean_invstd,
                      const float* weight, const float* gsum, int64_t n, int c, float* dx, float* dres,
                      hipStream_t s);

int bn_forward_from_block_stats(const float* x, int64_t n, int c, const float* partial, int blocks,
                                int64_t rows_per_block, const float* weight, const float* bias,
                                const float* residual, int relu, float eps, float momentum,
                                float* running_mean, float* running_var, float* mean_invstd, float* y,
                                hipStream_t s);
// sparse_conv_osm.hip: the mask-grouped output-stationary conv (pv2_spconv_osm)
int spconv_osm(bool trans, const float* in_feat, int c_in, const float* weight, int K, int c_out,
               const pv2_osm_plan_t* plan, int64_t n_out, const float* zero_row, const float* addend,
               float* out, float* bn_partial, int* bn_blocks, int* bn_rows_per_block, hipStream_t s);
// whether a conv of this shape takes that route (PV2_CONV_OSM = 0 / 1 / auto)
bool use_osm(const pv2_osm_plan_t* plan, const float* zero_row, int K, int64_t n_rows, int64_t n_other,
             int c_red, int c_cols);

}  // namespace pv2

#define PV2_REQUIRE(cond, msg)   \
  do {                           \
    if (!(cond)) {               \
      pv2::set_error(msg);       \
      return PV2_E_BADARG;       \
    }                            \
  } while (0)
