// Library-wide state of libponderv2_hip.so: ABI version and the per-thread error string.
#include <string.h>

#include "common.h"

namespace pv2 {
static thread_local char g_error[256] = "";
void set_error(const char* msg) {
  strncpy(g_error, msg ? msg : "", sizeof(g_error) - 1);
  g_error[sizeof(g_error) - 1] = 0;
}
}  // namespace pv2

namespace pv2 {
// 16-byte stores over the aligned body, single words either side of it (round 6: the 134 MB volume gradient of
// the render head took 96 us with one word per lane and trip - store-issue bound at 1.4 TB/s)
__global__ void zero_words_kernel(uint32_t* p, int64_t n32) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t head = (int64_t)(((16u - (unsigned)(reinterpret_cast<uintptr_t>(p) & 15u)) & 15u) >> 2);
  if (head > n32) head = n32;
  if (tid < head) p[tid] = 0u;
  uint4* q = reinterpret_cast<uint4*>(p + head);
  const int64_t n4 = (n32 - head) >> 2;
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (int64_t i = tid; i < n4; i += stride) q[i] = z;
  const int64_t done = head + 4 * n4;
  if (tid < n32 - done) p[done + tid] = 0u;
}
int zero_words(void* p, int64_t n32, hipStream_t s) {
  if (n32 <= 0) return PV2_OK;
  hipLaunchKernelGGL(zero_words_kernel, dim3(grid_for((n32 + 3) / 4 + 8, 256)), dim3(256), 0, s, (uint32_t*)p,
                     n32);
  return check_launch("zero_words");
}
}  // namespace pv2

extern "C" {
int pv2_zero_fill(void* ptr, int64_t nbytes, pv2_stream_t stream) {
  if (nbytes % 4 != 0) {
    pv2::set_error("pv2_zero_fill: nbytes must be a multiple of 4");
    return PV2_E_BADARG;
  }
  return pv2::zero_words(ptr, nbytes / 4, (hipStream_t)stream);
}
int pv2_abi_version(void) { return 16; }  // 16: PV2_UNET_CONV_BN16 (16-bit units in the native U-Net executor), pv2_ray_* / pv2_semantic_ce_* (per-ray epilogue + loss node); 15: pv2_unet_op.dx_producer (BatchNorm backward sums in the grad-input row reduce of an activation's last consumer); 14: pv2_osm_plan / pv2_spconv_osm (mask-grouped output-stationary convs), pv2_conv_geom.osm_*; 13: mode argument of pv2_dconv3_pack_weights / _packed_floats (bf16-piece weights for mode 0); 12: pv2_cells_* (first projection level from the occupied cells), pv2_bn_*_padded; 11: out_mask_src of pv2_dconv3_forward, relu_mask_src of pv2_maxpool3d_cl_backward_add, pv2_voxelize_*, pv2_bn_statistics; 10: pv2_dconv3_* (dense 3x3x3 convolutions), pv2_trilinear_*_16 (half sampler); 9: pv2_narrow_* (narrow-decoder render head), pv2_unet_op.weight_t, pv2_small_inverse; 8: product-row convs (pv2_spconv_products / _reduce_rows / pv2_convbn_*), deterministic weight gradient; 7: folded final convolution (pv2_neus_fold_*); 5: output-stationary convs, pv2_spconv_forward_wt (4: pv2_bn_* workspace)
const char* pv2_last_error(void) { return pv2::g_error; }
}
