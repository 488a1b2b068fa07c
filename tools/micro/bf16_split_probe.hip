// fp32 products on the bf16 matrix cores: x = x1 + x2 + x3 with three bf16 pieces (exact: 3 x 8
// mantissa bits, split by truncation), a*b ~ a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1 (the dropped terms
// are <= 2^-24 relative), six v_mfma_f32_32x32x16_bf16 in place of eight v_mfma_f32_32x32x2_f32 per
// 16 reduction steps: 6 x 32 cycles against 8 x 64.
//   1. accuracy of a 32 x 32 x K product against double, beside the fp32 MFMA's own
//   2. sustained rate of the inner loop with its companions (A split in registers per step, B pieces
//      read from LDS), in fp32-equivalent TFLOP/s
// build: hipcc --offload-arch=gfx950 -O3 -o bf16_split_probe bf16_split_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// the upper halves of two floats as one dword of two bf16 (lo = a, hi = b)
__device__ __forceinline__ unsigned pack_hi(float a, float b) {
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
// 8 floats -> three bf16x8 pieces (truncation split: exact)
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& p1, bf16x8& p2, bf16x8& p3) {
  float r1[8], r2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    r1[j] = x[j] - __uint_as_float(__float_as_uint(x[j]) & 0xffff0000u);
    r2[j] = r1[j] - __uint_as_float(__float_as_uint(r1[j]) & 0xffff0000u);
  }
  u32x4 a, b, c;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    a[j] = pack_hi(x[2 * j], x[2 * j + 1]);
    b[j] = pack_hi(r1[2 * j], r1[2 * j + 1]);
    c[j] = pack_hi(r2[2 * j], r2[2 * j + 1]);
  }
  p1 = __builtin_bit_cast(bf16x8, a);
  p2 = __builtin_bit_cast(bf16x8, b);
  p3 = __builtin_bit_cast(bf16x8, c);
}

__device__ __forceinline__ f32x16 mfma6(const bf16x8& a1, const bf16x8& a2, const bf16x8& a3, const bf16x8& b1,
                                         const bf16x8& b2, const bf16x8& b3, f32x16 acc) {
  // small terms first
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
  return acc;
}

// C[32, 32] = A[32, K] . B[32, K]^T with one wave; mode 0: fp32 MFMA, 1: six-term split, 2: three-term
__global__ void gemm_probe(const float* A, const float* B, int K, float* C, int mode) {
  const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (mode == 0) {
    for (int k = 0; k < K; k += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k + h], B[i * K + k + h], acc, 0, 0, 0);
  } else {
    for (int k = 0; k < K; k += 16) {
      float a[8], b[8];
      for (int j = 0; j < 8; ++j) a[j] = A[i * K + k + 8 * h + j], b[j] = B[i * K + k + 8 * h + j];
      bf16x8 a1, a2, a3, b1, b2, b3;
      split8(a, a1, a2, a3);
      split8(b, b1, b2, b3);
      if (mode == 1) {
        acc = mfma6(a1, a2, a3, b1, b2, b3, acc);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
      }
    }
  }
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}

// rate: per iteration one 32-wide reduction slab: A = 16 floats per lane split in registers, B pieces
// of NB column blocks read from LDS (3 x ds_read_b128 per block and 16-step), 12 * NB MFMAs
template <int NB, bool SPLIT_A>
__global__ __launch_bounds__(256) void rate_loop(float* out, const float4* __restrict__ g, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned lds[3 * 128 * 20];   // 3 pieces x 128 rows x (64 + 16) bytes
  for (int k = threadIdx.x; k < 3 * 128 * 20; k += 256) lds[k] = 0x3f803f80u;
  __syncthreads();
  const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  f32x16 acc[NB];
  for (int n = 0; n < NB; ++n)
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  float4 nxt[4];
  for (int s = 0; s < 4; ++s) nxt[s] = g[(s * 64 + lane) & 1023];
  bf16x8 pa[2][3];
  for (int it = 0; it < iters; ++it) {
    float4 cur[4];
    for (int s = 0; s < 4; ++s) cur[s] = nxt[s];
    for (int s = 0; s < 4; ++s) nxt[s] = g[((it * 4 + s) * 64 + lane) & 1023];   // the next slab's rows
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      if (SPLIT_A || it == 0) {
        const float x[8] = {cur[2 * st].x, cur[2 * st].y, cur[2 * st].z, cur[2 * st].w,
                            cur[2 * st + 1].x, cur[2 * st + 1].y, cur[2 * st + 1].z, cur[2 * st + 1].w};
        split8(x, pa[st][0], pa[st][1], pa[st][2]);
      }
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        bf16x8 b[3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
          b[p] = *reinterpret_cast<const bf16x8*>(&lds[(p * 128 + ((n * 32 + i) & 127)) * 20 + 8 * st + 4 * h]);
        acc[n] = mfma6(pa[st][0], pa[st][1], pa[st][2], b[0], b[1], b[2], acc[n]);
      }
    }
  }
  float s = 0.f;
  for (int n = 0; n < NB; ++n)
    for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[blockIdx.x * 256 + threadIdx.x] = s + nxt[0].x;
}

template <int NB, bool SPLIT_A>
void rate(int blocks, int iters, float* out, const float4* g) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((rate_loop<NB, SPLIT_A>), dim3(blocks), dim3(256), 0, 0, out, g, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flop = (double)blocks * 4 * iters * NB * 32.0 * 32 * 32 * 2;
  printf("NB=%d split_a=%d waves/SIMD=%d: %.1f us, %.1f fp32-equivalent TFLOP/s\n", NB, (int)SPLIT_A, blocks / 256,
         best * 1e3, flop / best / 1e9);
}

int main() {
  const int K = 256;
  std::vector<float> A(32 * K), B(32 * K);
  srand(1);
  for (auto& v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
  for (auto& v : B) v = (rand() / (float)RAND_MAX - 0.5f) * expf((rand() % 9) - 4.f);
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4), hipMalloc(&dB, B.size() * 4), hipMalloc(&dC, 1024 * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  std::vector<double> ref(1024), mag(1024);
  for (int m = 0; m < 32; ++m)
    for (int n = 0; n < 32; ++n) {
      double s = 0, a = 0;
      for (int k = 0; k < K; ++k) s += (double)A[m * K + k] * B[n * K + k], a += fabs((double)A[m * K + k] * B[n * K + k]);
      ref[m * 32 + n] = s, mag[m * 32 + n] = a;
    }
  const char* names[3] = {"fp32 MFMA (32x32x2_f32)", "bf16 pieces, six terms", "bf16 pieces, three terms"};
  for (int mode = 0; mode < 3; ++mode) {
    std::vector<float> C(1024);
    hipLaunchKernelGGL(gemm_probe, dim3(1), dim3(64), 0, 0, dA, dB, K, dC, mode);
    hipMemcpy(C.data(), dC, 1024 * 4, hipMemcpyDeviceToHost);
    double worst = 0, worst_rel = 0;
    for (int e = 0; e < 1024; ++e) {
      worst = fmax(worst, fabs(C[e] - ref[e]) / mag[e]);          // relative to sum |a b|
      worst_rel = fmax(worst_rel, fabs(C[e] - ref[e]) / fmax(fabs(ref[e]), 1e-30));
    }
    printf("%-28s K=%d: max |err| / sum|ab| = %.2e   max |err| / |c| = %.2e\n", names[mode], K, worst, worst_rel);
  }
  float* out;
  float4* g;
  hipMalloc(&out, 4096 * 256 * 4);
  hipMalloc(&g, 1024 * 16);
  hipMemset(g, 0, 1024 * 16);
  rate<4, true>(512, 400, out, g);
  rate<4, false>(512, 400, out, g);
  rate<4, true>(256, 400, out, g);
  rate<2, true>(512, 400, out, g);
  rate<1, true>(512, 400, out, g);
  rate<4, true>(1024, 200, out, g);
  return 0;
}
