// Sustained fp32 MFMA rate of the machine, and what the companions of a convolution's inner loop
// cost beside it: waves that do v_mfma_f32_32x32x2_f32 on registers (ACC independent accumulators
// each), optionally with LDS reads (DS float4 per 16 MFMAs), L2-resident global loads (GL float4 per
// 16 MFMAs) and register copies (MOV v_mov_b64 per 16 MFMAs), interleaved one by one.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int ACC, int DS, int GL, int MOV>
__global__ __launch_bounds__(256) void mfma_loop(float* out, const float4* __restrict__ w, int iters, float a0, float b0) {
  __shared__ float4 lds[2048];
  f32x16 acc[ACC];
  for (int k = 0; k < ACC; ++k)
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  for (int k = threadIdx.x; k < 2048; k += 256) lds[k] = make_float4(a0, a0, a0, a0);
  __syncthreads();
  float4 a = make_float4(a0 + threadIdx.x * 1e-6f, a0, a0, a0), b = make_float4(b0, b0, b0, b0);
  float4 an[DS > 0 ? DS : 1], bn[GL > 0 ? GL : 1];
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DS; ++d) an[d] = lds[(lane * 5 + d * 64 + it) & 2047];
#pragma unroll
    for (int d = 0; d < GL; ++d) bn[d] = w[((it * 4 + d) & 1023) * 64 + lane];
#pragma unroll
    for (int u = 0; u < 16 / ACC; ++u)
#pragma unroll
      for (int k = 0; k < ACC; ++k) {
        const float av = (u & 3) == 0 ? a.x : (u & 3) == 1 ? a.y : (u & 3) == 2 ? a.z : a.w;
        const float bv = (u & 3) == 0 ? b.x : (u & 3) == 1 ? b.y : (u & 3) == 2 ? b.z : b.w;
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[k], 0, 0, 0);
      }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x126, 2, 0);
    }
    if (DS > 0) a = an[0];
    if (GL > 0) b = bn[0];
#pragma unroll
    for (int d = 1; d < DS; ++d) a.x += an[d].x;
#pragma unroll
    for (int d = 1; d < GL; ++d) b.x += bn[d].x;
    if (MOV > 0) {
#pragma unroll
      for (int d = 0; d < MOV; ++d) asm volatile("v_mov_b32 %0, %0" : "+v"(a.y));
    }
  }
  float s = 0.f;
  for (int k = 0; k < ACC; ++k)
    for (int r = 0; r < 16; ++r) s += acc[k][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int ACC, int DS, int GL, int MOV>
void run(int blocks, int iters, float* out, const float4* w) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((mfma_loop<ACC, DS, GL, MOV>), dim3(blocks), dim3(256), 0, 0, out, w, iters, 1.0f, 0.5f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double flop = (double)blocks * 4 * iters * 16 * 4096.0;
  printf("ACC=%d DS=%d GL=%d MOV=%d waves/SIMD=%d: %.1f us, %.1f TFLOP/s\n", ACC, DS, GL, MOV, blocks / 256, best * 1e3,
         flop / best / 1e9);
}
int main(int argc, char** argv) {
  float* out;
  float4* w;
  hipMalloc(&out, 4096 * 256 * 4);
  hipMalloc(&w, 1024 * 64 * 16);
  hipMemset(w, 0, 1024 * 64 * 16);
  const int it = argc > 1 ? atoi(argv[1]) : 1500;
  run<2, 0, 0, 0>(512, it, out, w);
  run<4, 0, 0, 0>(512, it, out, w);
  run<2, 0, 0, 0>(1024, it / 2, out, w);
  run<1, 0, 0, 0>(1024, it / 2, out, w);
  run<2, 4, 0, 0>(512, it, out, w);
  run<2, 0, 2, 0>(512, it, out, w);
  run<2, 4, 2, 0>(512, it, out, w);
  run<2, 4, 2, 12>(512, it, out, w);
  run<2, 4, 2, 0>(256, it, out, w);
  run<2, 4, 2, 0>(768, it / 2, out, w);
  return 0;
}
