// fp32 products on the bf16 matrix cores of gfx950.
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the rate of v_mfma_f32_32x32x16_bf16 (256 against 4096
// multiply-adds per 8-pass slot).  A float is EXACTLY the sum of three bf16 numbers (3 x 8 mantissa
// bits; split by truncation: hi = upper 16 bits, then the same of the exact remainders), products of
// bf16 pieces are exact in the fp32 accumulator, so
//     a * b = a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1) + [terms <= 2^-24 |a b|, dropped]
// - six bf16 MFMAs per 16 reduction steps instead of eight fp32 MFMAs of twice the length: 2.67 x the
// fp32 matrix rate at fp32 accuracy (measured on MI355X, tools/micro/bf16_split_probe.hip: max error
// 2.2e-7 of sum |a b| at K = 256, the fp32 MFMA's own is 3.1e-7; 307 against 134 TFLOP/s in the inner
// loop with its companions).  The sparse and dense convolutions use it for every fp32 product.
//
// Range (tests/test_gpu_split_range.py, round 5).  The split is exact for every finite float whose pieces
// are normal bf16 numbers, i.e. down to ~2^-110; below that the smaller pieces are bf16 subnormals, which
// the matrix pipe flushes: operands in 2^-126 .. 2^-110 lose their low bits gradually (measured 1.2e-6 of
// sum|ab| at 2^-120 .. 2^-112), fp32 SUBNORMAL operands may vanish altogether.  Near FLT_MAX nothing
// overflows that fp32 would not (pieces only shrink).  Non-finite operands: inf splits into (inf, NaN, NaN)
// (inf - inf in bf16_rest) and a product with a zero piece of the other operand is inf * 0 - so an Inf
// input surfaces as NaN where the fp32 MFMA would give Inf.  The SET of non-finite outputs is the same
// (asserted per kernel family); an isfinite check - all a GradScaler does with it - cannot tell them apart.
//
// Operand layout of v_mfma_f32_32x32x16_bf16: lane (i = lane & 31, h = lane >> 5) holds the eight
// reduction steps 8 h .. 8 h + 7 of row i (A) / column i (B); D as the fp32 32x32 MFMA.
#pragma once
#include <hip/hip_runtime.h>

namespace pv2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// the upper halves of two floats as one dword holding two bf16 (low half: a, high half: b)
__device__ __forceinline__ unsigned pack_hi(float a, float b) {
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
// x minus its bf16 truncation (exact)
__device__ __forceinline__ float bf16_rest(float x) {
#ifdef PV2_FAKE_SPLIT   // timing probe only (tools/r06_fake_split.sh): what the kernels would cost with operands cut beforehand
  return x;
#else
  return x - __uint_as_float(__float_as_uint(x) & 0xffff0000u);
#endif
}

struct Split8 {
  bf16x8 p[3];
};

// eight floats (reduction steps in order) -> the three bf16 pieces of an MFMA operand
__device__ __forceinline__ Split8 split8(const float4& lo, const float4& hi) {
  const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  float r1[8], r2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    r1[j] = bf16_rest(x[j]);
    r2[j] = bf16_rest(r1[j]);
  }
  u32x4 a, b, c;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    a[j] = pack_hi(x[2 * j], x[2 * j + 1]);
    b[j] = pack_hi(r1[2 * j], r1[2 * j + 1]);
    c[j] = pack_hi(r2[2 * j], r2[2 * j + 1]);
  }
  Split8 s;
  s.p[0] = __builtin_bit_cast(bf16x8, a);
  s.p[1] = __builtin_bit_cast(bf16x8, b);
  s.p[2] = __builtin_bit_cast(bf16x8, c);
  return s;
}

// the six terms in ascending magnitude, (A piece, B piece): PV2_SPLIT_TERMS(F) expands F(a, b) six times
#define PV2_SPLIT_TERMS(F) F(2, 0) F(0, 2) F(1, 1) F(1, 0) F(0, 1) F(0, 0)

__device__ __forceinline__ f32x16 mfma_bf16(const bf16x8& a, const bf16x8& b, f32x16 acc) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}

}  // namespace pv2
