// The sparse U-Net's forward and backward pass as ONE native call each: a flat list of conv +
// BatchNorm units and skip concatenations, walked in C++.
//
// Stands in for the module-by-module execution of SpUNetBase.forward
// (ponder/models/sparse_unet/spconv_unet_v1m1_base.py:242-278: conv_input, 4 x (down + residual
// blocks), 4 x (up + skip concat + residual blocks)) and of its autograd graph.  The reference pays
// a Python call, an autograd node and a handful of launches per module and direction; on MI355X
// that made the training step HOST-bound (~10 ms of the 27 ms the host needs to enqueue a step
// were spent walking the backbone).  Here Python hands over an array of `pv2_unet_op` records
// (device pointers into arenas it allocated, host sizes); this file launches the same kernels the
// per-unit entry points launch (pv2_convbn_forward / _backward, sparse_conv_pr.hip), ~4 us each.
//
// Gradient bookkeeping is static: every activation has at most three consumers (conv, shortcut,
// skip), visited in reverse order; the FIRST gradient to arrive is written, later ones are added
// by the row-reduce kernel's addend input (`dx_accumulate`) - no zero-fills, no separate add
// kernels, fixed order.  Weight gradients go to the side stream behind one event per unit.
#include <cstdlib>
#include <vector>

#include <stdlib.h>
#include "common.h"

namespace {

// out[r, :] = [a[r, :ca] | b[r, :cb]]   (16-byte pieces; ca % 4 == 0, cb % 4 == 0)
__global__ __launch_bounds__(256) void concat_rows_kernel(const float4* __restrict__ a, int ca4,
                                                          const float4* __restrict__ b, int cb4,
                                                          int64_t n, float4* __restrict__ out) {
  const int c4 = ca4 + cb4;
  const int64_t total = n * c4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t r = e / c4;
    const int c = (int)(e - r * c4);
    out[e] = c < ca4 ? a[r * ca4 + c] : b[r * cb4 + (c - ca4)];
  }
}

// ga[r, :] = g[r, :ca], gb[r, :] (+)= g[r, ca:]
__global__ __launch_bounds__(256) void split_rows_kernel(const float4* __restrict__ g, int ca4,
                                                         int cb4, int64_t n, float4* __restrict__ ga,
                                                         float4* __restrict__ gb, int accumulate_b) {
  const int c4 = ca4 + cb4;
  const int64_t total = n * c4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t r = e / c4;
    const int c = (int)(e - r * c4);
    const float4 v = g[e];
    if (c < ca4) {
      ga[r * ca4 + c] = v;
    } else {
      float4* dst = gb + r * cb4 + (c - ca4);
      if (accumulate_b) {
        const float4 o = *dst;
        *dst = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
      } else {
        *dst = v;
      }
    }
  }
}

// ---- 16-bit activations (the reduced-precision training mode): eight elements per 16-byte piece
struct B16 {};
struct H16 {};
template <typename T> __device__ __forceinline__ float to_f32(unsigned short v);
template <> __device__ __forceinline__ float to_f32<B16>(unsigned short v) { return __uint_as_float((uint32_t)v << 16); }
template <> __device__ __forceinline__ float to_f32<H16>(unsigned short v) { return (float)__builtin_bit_cast(_Float16, v); }
template <typename T> __device__ __forceinline__ unsigned short from_f32(float v);
template <> __device__ __forceinline__ unsigned short from_f32<B16>(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
template <> __device__ __forceinline__ unsigned short from_f32<H16>(float v) { return __builtin_bit_cast(unsigned short, (_Float16)v); }

template <typename T>
__device__ __forceinline__ uint4 add8(const uint4& a, const uint4& b) {
  const uint32_t x[4] = {a.x, a.y, a.z, a.w}, y[4] = {b.x, b.y, b.z, b.w};
  uint32_t r[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float lo = to_f32<T>((unsigned short)(x[j] & 0xffffu)) + to_f32<T>((unsigned short)(y[j] & 0xffffu));
    const float hi = to_f32<T>((unsigned short)(x[j] >> 16)) + to_f32<T>((unsigned short)(y[j] >> 16));
    r[j] = (uint32_t)from_f32<T>(lo) | ((uint32_t)from_f32<T>(hi) << 16);
  }
  return make_uint4(r[0], r[1], r[2], r[3]);
}

// dst += src over n8 16-byte pieces (the sum rounded to the element type, as a tensor add would)
template <typename T>
__global__ __launch_bounds__(256) void add_rows16_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src,
                                                         int64_t n8) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n8; e += stride)
    dst[e] = add8<T>(dst[e], src[e]);
}

// split_rows_kernel for 16-bit rows (widths in 16-byte pieces)
template <typename T>
__global__ __launch_bounds__(256) void split_rows16_kernel(const uint4* __restrict__ g, int ca8, int cb8,
                                                           int64_t n, uint4* __restrict__ ga,
                                                           uint4* __restrict__ gb, int accumulate_b) {
  const int c8 = ca8 + cb8;
  const int64_t total = n * c8;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t r = e / c8;
    const int c = (int)(e - r * c8);
    const uint4 v = g[e];
    if (c < ca8) {
      ga[r * ca8 + c] = v;
    } else {
      uint4* dst = gb + r * cb8 + (c - ca8);
      *dst = accumulate_b ? add8<T>(*dst, v) : v;
    }
  }
}

// One conv + BatchNorm unit on 16-bit activations: the output-stationary 16-bit conv (one launch, no
// product rows) and the mixed-type BatchNorm of rownorm.hip.
int convbn16_forward(const pv2_unet_op& u, float* stats_ws, pv2_stream_t stream) {
  if (int e = pv2_spconv16_os_forward(u.x, u.n_in, u.c_in, u.packed_fwd, u.K, u.c_out, u.dtype, u.nbr,
                                      u.nbr_stride, u.perm, u.kflip, nullptr, u.y_conv, u.n_out, stream))
    return e;
  return pv2_bn_forward_mixed(u.y_conv, u.dtype, u.n_out, u.c_out, u.bn_weight, u.bn_bias, u.residual,
                              u.relu, u.eps, u.momentum, u.running_mean, u.running_var, stats_ws,
                              u.mean_invstd, u.out, u.dtype, stream);
}

int convbn16_backward(const pv2_unet_op& u, float* stats_ws, hipStream_t s, hipStream_t side) {
  if (int e = pv2_bn_backward_mixed(u.grad_out, u.y_conv, u.dtype, u.relu ? u.out : nullptr, u.dtype,
                                    u.mean_invstd, u.bn_weight, u.n_out, u.c_out, stats_ws, u.gsum, u.dy,
                                    u.dres, (pv2_stream_t)s))
    return e;
  // (where the weight gradient forks: in front of the grad-input launch by default - the fp32 units fork BEHIND
  // it, see convbn_backward in sparse_conv_pr.hip, but the 16-bit step is bound by its host and measured
  // 14.5 ms that way against 14.8; PV2_WGRAD_LATE16=1 for the other order)
  static const int wgrad_late = [] {
    const char* e = getenv("PV2_WGRAD_LATE16");
    return e ? atoi(e) : 0;
  }();
  auto weight_gradient = [&]() -> int {
    if (side != s) {
      hipEvent_t ev = nullptr;
      if (int e = pv2::fork_event_for(&ev)) return e;
      if (int e = pv2::hip_status(hipEventRecord(ev, s))) return e;
      if (int e = pv2::hip_status(hipStreamWaitEvent(side, ev, 0))) return e;
    }
    // (the 16-bit weight gradient ADDS its chunks into dweight)
    if (int e = pv2::zero_words(u.dweight, (int64_t)u.c_out * u.K * u.c_in, side)) return e;
    return pv2_spconv16_backward_weight(u.x, u.n_in, u.c_in, u.dy, u.n_out, u.c_out, u.dtype, u.K,
                                        u.geom->pair_in, u.geom->pair_out, u.geom->kstart, u.tile_start16, 512,
                                        u.n_tiles16, u.dweight, (pv2_stream_t)side);
  };
  const bool late = wgrad_late && u.dweight && u.dx && side != s;
  if (u.dweight && !late)
    if (int e = weight_gradient()) return e;
  if (u.dx) {
    void* target = u.dx_accumulate ? u.dx_tmp : (void*)u.dx;
    PV2_REQUIRE(target != nullptr, "pv2_unet_backward: 16-bit grad-input needs dx_tmp to accumulate");
    if (int e = pv2_spconv16_os_forward(u.dy, u.n_out, u.c_out, u.packed_bwd, u.K, u.c_in, u.dtype, u.nbr_t,
                                        u.nbr_t_stride, u.perm_t, u.kflip_t, nullptr, target, u.n_in,
                                        (pv2_stream_t)s))
      return e;
    if (late)
      if (int e = weight_gradient()) return e;
    if (u.dx_accumulate) {
      const int64_t n8 = u.n_in * u.c_in / 8;
      if (u.dtype == PV2_BF16)
        hipLaunchKernelGGL(add_rows16_kernel<B16>, dim3(pv2::grid_for(n8, 256)), dim3(256), 0, s,
                           (uint4*)u.dx, (const uint4*)u.dx_tmp, n8);
      else
        hipLaunchKernelGGL(add_rows16_kernel<H16>, dim3(pv2::grid_for(n8, 256)), dim3(256), 0, s,
                           (uint4*)u.dx, (const uint4*)u.dx_tmp, n8);
      return pv2::check_launch("unet_add16");
    }
  }
  return PV2_OK;
}

}  // namespace

extern "C" {

int pv2_unet_forward(const pv2_unet_op* ops, int n_ops, float* prod_ws, float* stats_ws,
                     pv2_stream_t stream) {
  PV2_REQUIRE(ops != nullptr && n_ops >= 0, "pv2_unet_forward: bad plan");
  hipStream_t s = (hipStream_t)stream;
  for (int i = 0; i < n_ops; ++i) {
    const pv2_unet_op& u = ops[i];
    int e = PV2_OK;
    switch (u.kind) {
      case PV2_UNET_CONV_BN:
        e = pv2_convbn_forward(u.geom, u.x, u.c_in, u.weight, u.c_out, u.bn_weight, u.bn_bias,
                               u.residual, u.relu, u.eps, u.momentum, u.running_mean, u.running_var,
                               prod_ws, stats_ws, u.y_conv, u.mean_invstd, u.out, stream);
        break;
      case PV2_UNET_CONV_BN16:
        PV2_REQUIRE(u.dtype == PV2_BF16 || u.dtype == PV2_F16, "pv2_unet_forward: CONV_BN16 needs a 16-bit dtype");
        e = convbn16_forward(u, stats_ws, stream);
        break;
      case PV2_UNET_STEM:
        e = pv2_spconv_os_forward(u.x, u.n_in, u.c_in, u.weight, u.K, u.c_out, u.nbr, u.nbr_stride,
                                  nullptr, u.kflip, nullptr, u.y_conv, u.n_out, stream);
        if (e == PV2_OK)   // (dtype != PV2_F32: the fp32 -> 16-bit edge of the reduced-precision mode)
          e = pv2_bn_forward_mixed(u.y_conv, PV2_F32, u.n_out, u.c_out, u.bn_weight, u.bn_bias, u.residual,
                                   u.relu, u.eps, u.momentum, u.running_mean, u.running_var, stats_ws,
                                   u.mean_invstd, u.out, u.dtype, stream);
        break;
      case PV2_UNET_CONCAT: {
        // (pieces of 16 bytes: four floats or eight 16-bit elements)
        const int per = u.dtype == PV2_F32 ? 4 : 8;
        PV2_REQUIRE((u.c_in % per) == 0 && (u.c_out % per) == 0, "pv2_unet_forward: concat widths");
        if (u.n_out > 0)
          hipLaunchKernelGGL(concat_rows_kernel,
                             dim3(pv2::grid_for(u.n_out * ((u.c_in + u.c_out) / per), 256)), dim3(256),
                             0, s, (const float4*)u.x, u.c_in / per, (const float4*)u.residual,
                             u.c_out / per, u.n_out, (float4*)u.out);
        e = pv2::check_launch("unet_concat");
        break;
      }
      default:
        pv2::set_error("pv2_unet_forward: unknown op kind");
        return PV2_E_BADARG;
    }
    if (e != PV2_OK) return e;
  }
  return PV2_OK;
}

int pv2_unet_backward(const pv2_unet_op* ops, int n_ops, float* prod_ws, float* stats_ws,
                      float* part_ws, pv2_stream_t stream, pv2_stream_t side_stream) {
  return pv2_unet_backward_ev(ops, n_ops, prod_ws, stats_ws, part_ws, stream, side_stream, 0, nullptr,
                              nullptr, nullptr);
}

int pv2_unet_backward_ev(const pv2_unet_op* ops, int n_ops, float* prod_ws, float* stats_ws,
                         float* part_ws, pv2_stream_t stream, pv2_stream_t side_stream, int n_ckpt,
                         const int32_t* ckpt_unit, void* const* ckpt_event_main,
                         void* const* ckpt_event_side) {
  PV2_REQUIRE(ops != nullptr && n_ops >= 0, "pv2_unet_backward: bad plan");
  PV2_REQUIRE(n_ckpt == 0 || (ckpt_unit != nullptr && ckpt_event_main != nullptr && ckpt_event_side != nullptr),
              "pv2_unet_backward_ev: checkpoint arrays missing");
  for (int j = 0; j < n_ckpt; ++j)
    PV2_REQUIRE(ckpt_unit[j] >= 0 && ckpt_unit[j] < n_ops && (j == 0 || ckpt_unit[j] < ckpt_unit[j - 1]),
                "pv2_unet_backward_ev: checkpoints must be unit indices in descending order");
  hipStream_t s = (hipStream_t)stream;
  int next_ckpt = 0;   // checkpoints are listed in the order the units finish: descending unit index
  // sums_ready[i]: the BatchNorm backward sums of unit i were taken by the launch that completed the
  // gradient of its output (the grad-input row reduce of the unit's LAST consumer in backward order,
  // op.dx_producer) - unit i then runs only the elementwise half of its BatchNorm backward
  std::vector<char> sums_ready((size_t)n_ops, 0);
  static const bool fuse_sums = [] {
    const char* e = getenv("PV2_BN_BWD_FUSED");   // 0: every BatchNorm backward takes its own sums (A / B)
    return !(e && e[0] == '0');
  }();
  for (int i = n_ops - 1; i >= 0; --i) {
    const pv2_unet_op& u = ops[i];
    int e = PV2_OK;
    switch (u.kind) {
      case PV2_UNET_CONV_BN: {
        pv2::BnProducer prod{};
        const int pi = u.dx_producer - 1;
        const bool fuse = fuse_sums && u.dx != nullptr && pi >= 0 && pi < i && ops[pi].kind != PV2_UNET_CONCAT &&
                          ops[pi].out == u.x && ops[pi].c_out == u.c_in && ops[pi].gsum != nullptr;
        if (fuse) {
          prod.y_conv = ops[pi].y_conv;
          prod.out_or_null = ops[pi].relu ? ops[pi].out : nullptr;
          prod.mean_invstd = ops[pi].mean_invstd;
          prod.gsum = ops[pi].gsum;
        }
        int done = 0;
        e = pv2::convbn_backward(u.geom, u.grad_out, u.x, u.c_in, u.weight, u.c_out, u.y_conv,
                                 u.relu ? u.out : nullptr, u.mean_invstd, u.bn_weight, prod_ws,
                                 stats_ws, u.gsum, u.dy, u.dres, u.dx, u.dx_accumulate, u.dweight,
                                 part_ws, s, side_stream ? (hipStream_t)side_stream : s, sums_ready[i],
                                 fuse ? &prod : nullptr, &done);
        if (done) sums_ready[pi] = 1;
        break;
      }
      case PV2_UNET_CONV_BN16:
        e = convbn16_backward(u, stats_ws, s, side_stream ? (hipStream_t)side_stream : s);
        break;
      case PV2_UNET_STEM: {
        // (the stem is the last unit of the backward pass: the product-row workspace is free and
        // serves as the weight gradient's partial-sum buffer, on the caller's stream)
        if (u.dtype != PV2_F32)
          e = pv2_bn_backward_mixed(u.grad_out, u.y_conv, PV2_F32, u.relu ? u.out : nullptr, u.dtype,
                                    u.mean_invstd, u.bn_weight, u.n_out, u.c_out, stats_ws, u.gsum, u.dy,
                                    u.dres, stream);
        else if (sums_ready[i])
          e = pv2::bn_backward_apply(u.grad_out, u.y_conv, u.relu ? u.out : nullptr, u.mean_invstd,
                                     u.bn_weight, u.gsum, u.n_out, u.c_out, u.dy, u.dres, s);
        else
          e = pv2_bn_backward(u.grad_out, u.y_conv, u.relu ? u.out : nullptr, u.mean_invstd,
                              u.bn_weight, u.n_out, u.c_out, stats_ws, u.gsum, u.dy, u.dres, stream);
        if (e == PV2_OK && u.dweight)
          e = pv2::spconv_wgrad(u.x, u.n_in, u.c_in, u.dy, u.n_out, u.c_out, u.K, u.geom->pair_in,
                                u.geom->pair_out, u.geom->kstart, u.geom->tile_start_w,
                                u.geom->tile_pairs_w, u.geom->n_tiles_w, u.dweight, prod_ws, s);
        if (e == PV2_OK && u.dx) {
          // the input features carry a gradient (a learnable mask token was written into them): the same
          // gather table walked with mirrored offsets and the transposed weight (submanifold: in == out)
          PV2_REQUIRE(u.weight_t != nullptr && u.n_in == u.n_out && !u.dx_accumulate,
                      "pv2_unet_backward: stem grad-input needs weight_t");
          e = pv2_spconv_os_forward(u.dy, u.n_out, u.c_out, u.weight_t, u.K, u.c_in, u.nbr, u.nbr_stride,
                                    nullptr, !u.kflip, nullptr, u.dx, u.n_in, stream);
        }
        break;
      }
      case PV2_UNET_CONCAT:
        if (u.n_out > 0) {
          if (u.dtype == PV2_F32)
            hipLaunchKernelGGL(split_rows_kernel,
                               dim3(pv2::grid_for(u.n_out * ((u.c_in + u.c_out) / 4), 256)), dim3(256),
                               0, s, (const float4*)u.grad_out, u.c_in / 4, u.c_out / 4, u.n_out,
                               (float4*)u.dx, (float4*)u.dres, u.dx_accumulate);
          else if (u.dtype == PV2_BF16)
            hipLaunchKernelGGL(split_rows16_kernel<B16>,
                               dim3(pv2::grid_for(u.n_out * ((u.c_in + u.c_out) / 8), 256)), dim3(256),
                               0, s, (const uint4*)u.grad_out, u.c_in / 8, u.c_out / 8, u.n_out,
                               (uint4*)u.dx, (uint4*)u.dres, u.dx_accumulate);
          else
            hipLaunchKernelGGL(split_rows16_kernel<H16>,
                               dim3(pv2::grid_for(u.n_out * ((u.c_in + u.c_out) / 8), 256)), dim3(256),
                               0, s, (const uint4*)u.grad_out, u.c_in / 8, u.c_out / 8, u.n_out,
                               (uint4*)u.dx, (uint4*)u.dres, u.dx_accumulate);
        }
        e = pv2::check_launch("unet_split");
        break;
      default:
        pv2::set_error("pv2_unet_backward: unknown op kind");
        return PV2_E_BADARG;
    }
    if (e != PV2_OK) return e;
    while (next_ckpt < n_ckpt && ckpt_unit[next_ckpt] >= i) {
      // every parameter gradient of units >= i is now queued: BatchNorm gradients on `stream`, weight
      // gradients on the side stream.  Whoever waits for both events may read them (the gradient
      // reduction of a data-parallel step starts on this slab while the units below still run).
      if (int err = pv2::hip_status(hipEventRecord((hipEvent_t)ckpt_event_main[next_ckpt], s))) return err;
      hipStream_t side = side_stream ? (hipStream_t)side_stream : s;
      if (int err = pv2::hip_status(hipEventRecord((hipEvent_t)ckpt_event_side[next_ckpt], side))) return err;
      ++next_ckpt;
    }
  }
  return PV2_OK;
}

}  // extern "C"
