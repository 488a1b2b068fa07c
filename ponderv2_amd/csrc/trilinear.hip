// Twice-differentiable trilinear grid sampler for gfx950 (forward, backward, backward-of-backward).
//
// Stands in for libs/smooth-sampler (smooth_sampler_kernel.cu:39-153 forward, :155-356 backward,
// :358-619 backward-backward; bindings smooth_sampler.cpp:36-98).  The maths is re-derived from
// the separable weight model below rather than transliterated:
//
//   per axis a: source coordinate x_a = T_a(g_a) (un-normalise, then clip / reflect by padding
//   mode) with slope m_a = dT_a/dg_a; t_a = x_a - floor(x_a); s_a = S(t_a) where S is identity or
//   smoothstep; corner bit b in {0,1} has weight om_a(b) = b ? s_a : 1 - s_a,
//   d om_a(b)/dg_a = (2b-1) S'(t_a) m_a,  d2 om_a(b)/dg_a^2 = (2b-1) S''(t_a) m_a^2.
//   corner weight w_c = om_x(cx) om_y(cy) om_z(cz).
//
//   forward :  out[ch]   = sum_c w_c V[ch,c]
//   backward:  gV[ch,c] += w_c gO[ch]            gG_a = sum_ch gO[ch] sum_c (d_a w_c) V[ch,c]
//   bwd-bwd (upstream hV for gV, hG for gG), with D_c = sum_a hG_a d_a w_c:
//     ggO[ch]   = sum_c (hV[ch,c] w_c + V[ch,c] D_c)
//     gV2[ch,c]+= gO[ch] D_c
//     gG2_b     = sum_ch gO[ch] sum_c ( hV[ch,c] d_b w_c + V[ch,c] sum_a hG_a d_a d_b w_c )
//
// Mapping to the machine, generic kernels (any strides, float / double): ONE WAVE PER SAMPLE POINT,
// lanes run along the channel axis.  With a channels-last (NDHWC) volume each of the 8 corners is
// one contiguous C*4-byte read (512 B at C=128) and the volume-gradient atomics are contiguous
// too; with the reference's NCDHW layout the same kernels still work through the stride
// descriptor, just uncoalesced.  The three grid gradients are wave-reduced with cross-lane
// shuffles, no LDS.  The fp32 channels-last fast path (lane groups per point, float4 corner reads,
// measured atomic mappings, scrambled visiting order) follows further down.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace {

// Storage types of the 16-bit instantiations (the reference dispatches its kernels over half too:
// smooth_sampler_kernel.cu:630,670,726 AT_DISPATCH_FLOATING_TYPES_AND_HALF).  Arithmetic is fp32
// (T), operands are widened on load and results rounded to nearest-even on store (S); the
// atomically accumulated volume gradient stays an fp32 buffer - the caller narrows it once.
struct H16 { unsigned short v; };   // IEEE binary16
struct B16 { unsigned short v; };   // bfloat16
template <typename T, typename S>
__device__ __forceinline__ T ldv(const S* __restrict__ p, int64_t i) { return (T)p[i]; }
template <>
__device__ __forceinline__ float ldv<float, H16>(const H16* __restrict__ p, int64_t i) {
  return (float)__builtin_bit_cast(_Float16, p[i].v);
}
template <>
__device__ __forceinline__ float ldv<float, B16>(const B16* __restrict__ p, int64_t i) {
  return __builtin_bit_cast(float, (unsigned)p[i].v << 16);
}
template <typename T, typename S>
__device__ __forceinline__ void stv(S* __restrict__ p, int64_t i, T v) { p[i] = (S)v; }
template <>
__device__ __forceinline__ void stv<float, H16>(H16* __restrict__ p, int64_t i, float v) {
  p[i].v = __builtin_bit_cast(unsigned short, (_Float16)v);
}
template <>
__device__ __forceinline__ void stv<float, B16>(B16* __restrict__ p, int64_t i, float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  // round to nearest even; NaN keeps a quiet payload
  p[i].v = (v != v) ? (unsigned short)((u >> 16) | 0x40) : (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <typename T>
struct Axis {
  T w0, w1;   // om(0), om(1)
  T dw;       // d om(1) / dg   (d om(0)/dg = -dw)
  T d2w;      // d2 om(1) / dg2
  int64_t i0; // floor(x)
};

template <typename T>
__device__ __forceinline__ T reflect_coord(T x, int64_t twice_low, int64_t twice_high, T* mult) {
  if (twice_low == twice_high) {
    *mult = T(0);
    return T(0);
  }
  const T lo = T(twice_low) / T(2);
  const T span = T(twice_high - twice_low) / T(2);
  T v = x - lo;
  T sgn = T(1);
  if (v < T(0)) {
    v = -v;
    sgn = T(-1);
  }
  const T extra = fmod(v, span);
  const int64_t flips = (int64_t)floor(v / span);
  if ((flips & 1) == 0) {
    *mult = sgn;
    return extra + lo;
  }
  *mult = -sgn;
  return span - extra + lo;
}

template <typename T>
__device__ __forceinline__ T clip_coord(T x, int64_t size, T* mult) {
  if (x <= T(0)) {
    *mult = T(0);
    return T(0);
  }
  const T hi = T(size - 1);
  if (x >= hi) {
    *mult = T(0);
    return hi;
  }
  *mult = T(1);
  return x;
}

template <typename T>
__device__ __forceinline__ Axis<T> make_axis(T g, int64_t size, int padding, bool align,
                                             bool smooth) {
  T x, m;
  if (align) {
    x = ((g + T(1)) / T(2)) * T(size - 1);
    m = T(size - 1) / T(2);
  } else {
    x = ((g + T(1)) * T(size) - T(1)) / T(2);
    m = T(size) / T(2);
  }
  if (padding == 1) {
    T cm;
    x = clip_coord(x, size, &cm);
    m *= cm;
  } else if (padding == 2) {
    T rm, cm;
    if (align) x = reflect_coord(x, 0, 2 * (size - 1), &rm);
    else x = reflect_coord(x, -1, 2 * size - 1, &rm);
    x = clip_coord(x, size, &cm);
    m *= rm * cm;
  }
  Axis<T> a;
  const T fl = floor(x);
  a.i0 = (int64_t)fl;
  const T t = x - fl;
  T s = t, sp = T(1), spp = T(0);
  if (smooth) {
    s = t * t * (T(3) - T(2) * t);
    sp = T(6) * t * (T(1) - t);
    spp = T(6) - T(12) * t;
  }
  a.w1 = s;
  a.w0 = T(1) - s;
  a.dw = sp * m;
  a.d2w = spp * m * m;
  return a;
}

struct Geom {
  int64_t off[8];
  bool inb[8];
};

template <typename T>
struct Point {
  Axis<T> ax, ay, az;
  Geom g;
};

template <typename T, typename S = T>
__device__ __forceinline__ Point<T> make_point(const S* __restrict__ grid, int64_t pt,
                                               const pv2_volume_desc& v, int padding, bool align,
                                               bool smooth) {
  Point<T> p;
  p.ax = make_axis<T>(ldv<T, S>(grid, pt * 3 + 0), v.w, padding, align, smooth);
  p.ay = make_axis<T>(ldv<T, S>(grid, pt * 3 + 1), v.h, padding, align, smooth);
  p.az = make_axis<T>(ldv<T, S>(grid, pt * 3 + 2), v.d, padding, align, smooth);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int64_t ix = p.ax.i0 + (c & 1), iy = p.ay.i0 + ((c >> 1) & 1), iz = p.az.i0 + (c >> 2);
    p.g.inb[c] = ix >= 0 && ix < v.w && iy >= 0 && iy < v.h && iz >= 0 && iz < v.d;
    p.g.off[c] = iz * v.sd + iy * v.sh + ix * v.sw;
  }
  return p;
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// corner order: x fastest, then y, then z (the reference's tnw,tne,tsw,tse,bnw,... order)
#define PV2_CORNER_WEIGHTS(p, w)                                        \
  _Pragma("unroll") for (int c = 0; c < 8; ++c) {                       \
    const T wx = (c & 1) ? p.ax.w1 : p.ax.w0;                           \
    const T wy = (c & 2) ? p.ay.w1 : p.ay.w0;                           \
    const T wz = (c & 4) ? p.az.w1 : p.az.w0;                           \
    w[c] = wx * wy * wz;                                                \
  }

template <typename T, typename S = T>
__global__ __launch_bounds__(256) void tri_fwd_kernel(const S* __restrict__ in, pv2_volume_desc v,
                                                      const S* __restrict__ grid,
                                                      pv2_points_desc pd, S* __restrict__ out,
                                                      int padding, int align, int smooth) {
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t pt = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); pt < pd.n_points; pt += nwaves) {
    const int64_t n = pt / pd.points_per_n, q = pt % pd.points_per_n;
    const Point<T> p = make_point<T, S>(grid, pt, v, padding, align != 0, smooth != 0);
    T w[8];
    PV2_CORNER_WEIGHTS(p, w)
    const S* base = in + n * v.sn;
    S* obase = out + n * pd.o_sn + q * pd.o_sp;
    for (int64_t ch = lane; ch < v.c; ch += 64) {
      const S* src = base + ch * v.sc;
      T acc = T(0);
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (p.g.inb[c]) acc += ldv<T, S>(src, p.g.off[c]) * w[c];
      stv<T, S>(obase, ch * pd.o_sc, acc);
    }
  }
}

template <typename T, typename S = T>
__global__ __launch_bounds__(256) void tri_bwd_kernel(const S* __restrict__ gout,
                                                      const S* __restrict__ in, pv2_volume_desc v,
                                                      const S* __restrict__ grid,
                                                      pv2_points_desc pd, T* __restrict__ gin,
                                                      S* __restrict__ ggrid, int padding,
                                                      int align, int smooth) {
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t pt = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); pt < pd.n_points; pt += nwaves) {
    const int64_t n = pt / pd.points_per_n, q = pt % pd.points_per_n;
    const Point<T> p = make_point<T, S>(grid, pt, v, padding, align != 0, smooth != 0);
    T w[8], dx[8], dy[8], dz[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const T wx = (c & 1) ? p.ax.w1 : p.ax.w0, sx = (c & 1) ? p.ax.dw : -p.ax.dw;
      const T wy = (c & 2) ? p.ay.w1 : p.ay.w0, sy = (c & 2) ? p.ay.dw : -p.ay.dw;
      const T wz = (c & 4) ? p.az.w1 : p.az.w0, sz = (c & 4) ? p.az.dw : -p.az.dw;
      w[c] = wx * wy * wz;
      dx[c] = sx * wy * wz;
      dy[c] = wx * sy * wz;
      dz[c] = wx * wy * sz;
    }
    const S* base = in + n * v.sn;
    T* gbase = gin ? gin + n * v.sn : nullptr;
    const S* gobase = gout + n * pd.o_sn + q * pd.o_sp;
    T gx = T(0), gy = T(0), gz = T(0);
    for (int64_t ch = lane; ch < v.c; ch += 64) {
      const T go = ldv<T, S>(gobase, ch * pd.o_sc);
      const S* src = base + ch * v.sc;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (p.g.inb[c]) {
          const T val = ldv<T, S>(src, p.g.off[c]);
          gx += go * val * dx[c];
          gy += go * val * dy[c];
          gz += go * val * dz[c];
          if (gbase) unsafeAtomicAdd(gbase + ch * v.sc + p.g.off[c], w[c] * go);
        }
      }
    }
    gx = wave_sum(gx);
    gy = wave_sum(gy);
    gz = wave_sum(gz);
    if (lane == 0) {
      stv<T, S>(ggrid, pt * 3 + 0, gx);
      stv<T, S>(ggrid, pt * 3 + 1, gy);
      stv<T, S>(ggrid, pt * 3 + 2, gz);
    }
  }
}

template <typename T, typename S = T>
__global__ __launch_bounds__(256) void tri_bwdbwd_kernel(
    const S* __restrict__ hV, const S* __restrict__ hG, const S* __restrict__ in,
    pv2_volume_desc v, const S* __restrict__ grid, const S* __restrict__ gout, pv2_points_desc pd,
    T* __restrict__ gin2, S* __restrict__ ggrid2, S* __restrict__ ggout, int padding, int align,
    int smooth) {
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t pt = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); pt < pd.n_points; pt += nwaves) {
    const int64_t n = pt / pd.points_per_n, q = pt % pd.points_per_n;
    const Point<T> p = make_point<T, S>(grid, pt, v, padding, align != 0, smooth != 0);
    const T hx = ldv<T, S>(hG, pt * 3 + 0), hy = ldv<T, S>(hG, pt * 3 + 1),
            hz = ldv<T, S>(hG, pt * 3 + 2);
    T w[8], dx[8], dy[8], dz[8], D[8], ex[8], ey[8], ez[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const T sgx = (c & 1) ? T(1) : T(-1), sgy = (c & 2) ? T(1) : T(-1),
              sgz = (c & 4) ? T(1) : T(-1);
      const T wx = (c & 1) ? p.ax.w1 : p.ax.w0, sx = sgx * p.ax.dw, cx = sgx * p.ax.d2w;
      const T wy = (c & 2) ? p.ay.w1 : p.ay.w0, sy = sgy * p.ay.dw, cy = sgy * p.ay.d2w;
      const T wz = (c & 4) ? p.az.w1 : p.az.w0, sz = sgz * p.az.dw, cz = sgz * p.az.d2w;
      w[c] = wx * wy * wz;
      dx[c] = sx * wy * wz;
      dy[c] = wx * sy * wz;
      dz[c] = wx * wy * sz;
      D[c] = hx * dx[c] + hy * dy[c] + hz * dz[c];
      // E_b = sum_a hG_a d_a d_b w_c
      ex[c] = hx * (cx * wy * wz) + hy * (sx * sy * wz) + hz * (sx * wy * sz);
      ey[c] = hx * (sx * sy * wz) + hy * (wx * cy * wz) + hz * (wx * sy * sz);
      ez[c] = hx * (sx * wy * sz) + hy * (wx * sy * sz) + hz * (wx * wy * cz);
    }
    const S* base = in + n * v.sn;
    const S* hbase = hV ? hV + n * v.sn : nullptr;
    T* gbase = gin2 ? gin2 + n * v.sn : nullptr;
    const S* gobase = gout + n * pd.o_sn + q * pd.o_sp;
    S* ggobase = ggout + n * pd.o_sn + q * pd.o_sp;
    T gx = T(0), gy = T(0), gz = T(0);
    for (int64_t ch = lane; ch < v.c; ch += 64) {
      const T go = ldv<T, S>(gobase, ch * pd.o_sc);
      const int64_t choff = ch * v.sc;
      T ggo = T(0);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (p.g.inb[c]) {
          const T val = ldv<T, S>(base, choff + p.g.off[c]);
          const T hv = hbase ? ldv<T, S>(hbase, choff + p.g.off[c]) : T(0);
          ggo += val * D[c] + hv * w[c];
          gx += go * (hv * dx[c] + val * ex[c]);
          gy += go * (hv * dy[c] + val * ey[c]);
          gz += go * (hv * dz[c] + val * ez[c]);
          if (gbase) unsafeAtomicAdd(gbase + choff + p.g.off[c], go * D[c]);
        }
      }
      stv<T, S>(ggobase, ch * pd.o_sc, ggo);
    }
    gx = wave_sum(gx);
    gy = wave_sum(gy);
    gz = wave_sum(gz);
    if (lane == 0) {
      stv<T, S>(ggrid2, pt * 3 + 0, gx);
      stv<T, S>(ggrid2, pt * 3 + 1, gy);
      stv<T, S>(ggrid2, pt * 3 + 2, gz);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Channels-last fp32 fast path (what the render head actually runs: NDHWC volume, (points, C)
// outputs).  LPP lanes share one sample point, each lane owning 4 consecutive channels of every
// LPP*4-channel panel, so a wave handles 64/LPP points at once (2 at C=128, 8 at C=32): every
// corner read is one 16-byte load per lane, per-point geometry is computed 64/LPP times per wave
// instead of once, and the grid-gradient reductions are log2(LPP) shuffles inside the lane group.
// Selected by `vec_ok` below; anything else (f64, NCDHW, unaligned views) uses the generic kernels.
// ---------------------------------------------------------------------------------------------
struct Geom32 {
  int off[8];
  bool inb[8];
};

__device__ __forceinline__ void make_geom32(const Axis<float>& ax, const Axis<float>& ay,
                                            const Axis<float>& az, const pv2_volume_desc& v,
                                            Geom32* g) {
  const int x0 = (int)ax.i0, y0 = (int)ay.i0, z0 = (int)az.i0;
  const int W = (int)v.w, H = (int)v.h, D = (int)v.d;
  const int sw = (int)v.sw, sh = (int)v.sh, sd = (int)v.sd;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int ix = x0 + (c & 1), iy = y0 + ((c >> 1) & 1), iz = z0 + (c >> 2);
    g->inb[c] = ix >= 0 && ix < W && iy >= 0 && iy < H && iz >= 0 && iz < D;
    g->off[c] = iz * sd + iy * sh + ix * sw;
  }
}

template <int LPP>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

#define PV2_VEC_PROLOGUE                                                                        \
  constexpr int PPW = 64 / LPP;                                                                 \
  const int lane = threadIdx.x & 63, sub = lane % LPP, slot = lane / LPP;                       \
  const int64_t nw = (int64_t)gridDim.x * 4;                                                    \
  const int c4 = (int)(v.c >> 2);                                                               \
  for (int64_t p0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * PPW; p0 < pd.n_points;     \
       p0 += nw * PPW) {                                                                        \
    const bool valid = p0 + slot < pd.n_points;                                                 \
    int64_t pt = valid ? p0 + slot : pd.n_points - 1;                                           \
    if (perm) pt = (int64_t)((uint64_t)pt * perm % (uint64_t)pd.n_points);                      \
    const int64_t n = pt / pd.points_per_n, q = pt % pd.points_per_n;                           \
    const Axis<float> ax = make_axis<float>(grid[pt * 3 + 0], v.w, padding, align != 0, smooth != 0); \
    const Axis<float> ay = make_axis<float>(grid[pt * 3 + 1], v.h, padding, align != 0, smooth != 0); \
    const Axis<float> az = make_axis<float>(grid[pt * 3 + 2], v.d, padding, align != 0, smooth != 0); \
    Geom32 g;                                                                                   \
    make_geom32(ax, ay, az, v, &g);

template <int LPP>
__global__ __launch_bounds__(256) void tri_fwd_vec_kernel(const float* __restrict__ in,
                                                          pv2_volume_desc v,
                                                          const float* __restrict__ grid,
                                                          pv2_points_desc pd,
                                                          float* __restrict__ out, int padding,
                                                          int align, int smooth, uint32_t perm) {
  PV2_VEC_PROLOGUE
    float w[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
      w[c] = ((c & 1) ? ax.w1 : ax.w0) * ((c & 2) ? ay.w1 : ay.w0) * ((c & 4) ? az.w1 : az.w0);
    const float* base = in + n * v.sn;
    float* obase = out + n * pd.o_sn + q * pd.o_sp;
    for (int ch4 = sub; ch4 < c4; ch4 += LPP) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (g.inb[c]) {
          const float4 val = *reinterpret_cast<const float4*>(base + g.off[c] + 4 * ch4);
          acc.x += val.x * w[c];
          acc.y += val.y * w[c];
          acc.z += val.z * w[c];
          acc.w += val.w * w[c];
        }
      }
      if (valid) *reinterpret_cast<float4*>(obase + 4 * ch4) = acc;
    }
  }
}

// ATOM selects how the volume-gradient atomics are issued: 0 = not here (tri_scatter_kernel does
// them), 1 = from the float4 lanes (each lane 4 consecutive channels, 16-byte lane stride),
// 2 = transposed (instruction j covers the contiguous channels pb + j*LPP .. +LPP of each group).
template <int LPP, int ATOM>
__global__ __launch_bounds__(256) void tri_bwd_vec_kernel(
    const float* __restrict__ gout, const float* __restrict__ in, pv2_volume_desc v,
    const float* __restrict__ grid, pv2_points_desc pd, float* __restrict__ gin,
    float* __restrict__ ggrid, int padding, int align, int smooth, uint32_t perm) {
  PV2_VEC_PROLOGUE
    float w[8], dx[8], dy[8], dz[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float wx = (c & 1) ? ax.w1 : ax.w0, sx = (c & 1) ? ax.dw : -ax.dw;
      const float wy = (c & 2) ? ay.w1 : ay.w0, sy = (c & 2) ? ay.dw : -ay.dw;
      const float wz = (c & 4) ? az.w1 : az.w0, sz = (c & 4) ? az.dw : -az.dw;
      w[c] = wx * wy * wz;
      dx[c] = sx * wy * wz;
      dy[c] = wx * sy * wz;
      dz[c] = wx * wy * sz;
    }
    const float* base = in + n * v.sn;
    float* gbase = gin ? gin + n * v.sn : nullptr;
    const float* gobase = gout + n * pd.o_sn + q * pd.o_sp;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    const int C = (int)v.c;
    for (int pb = 0; pb < C; pb += 4 * LPP) {  // channel panels of 4*LPP
      const int ch4 = (pb >> 2) + sub;
      const bool lane_on = ch4 < c4;
      const float4 go = lane_on ? *reinterpret_cast<const float4*>(gobase + 4 * ch4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
      float got[4];
      if (ATOM == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ch = pb + j * LPP + sub;
          got[j] = ch < C ? gobase[ch] : 0.f;
        }
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (g.inb[c]) {
          if (lane_on) {
            const float4 val = *reinterpret_cast<const float4*>(base + g.off[c] + 4 * ch4);
            const float d = dot4(go, val);
            gx += d * dx[c];
            gy += d * dy[c];
            gz += d * dz[c];
          }
          if (ATOM == 1 && gbase && valid && lane_on) {
            float* dst = gbase + g.off[c] + 4 * ch4;
            unsafeAtomicAdd(dst + 0, w[c] * go.x);
            unsafeAtomicAdd(dst + 1, w[c] * go.y);
            unsafeAtomicAdd(dst + 2, w[c] * go.z);
            unsafeAtomicAdd(dst + 3, w[c] * go.w);
          }
          if (ATOM == 2 && gbase && valid) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int ch = pb + j * LPP + sub;
              if (ch < C) unsafeAtomicAdd(gbase + g.off[c] + ch, w[c] * got[j]);
            }
          }
        }
      }
    }
    gx = group_sum<LPP>(gx);
    gy = group_sum<LPP>(gy);
    gz = group_sum<LPP>(gz);
    if (sub == 0 && valid) {
      ggrid[pt * 3 + 0] = gx;
      ggrid[pt * 3 + 1] = gy;
      ggrid[pt * 3 + 2] = gz;
    }
  }
}

template <int LPP, int ATOM>
__global__ __launch_bounds__(256) void tri_bwdbwd_vec_kernel(
    const float* __restrict__ hV, const float* __restrict__ hG, const float* __restrict__ in,
    pv2_volume_desc v, const float* __restrict__ grid, const float* __restrict__ gout,
    pv2_points_desc pd, float* __restrict__ gin2, float* __restrict__ ggrid2,
    float* __restrict__ ggout, int padding, int align, int smooth, uint32_t perm) {
  PV2_VEC_PROLOGUE
    const float hx = hG[pt * 3 + 0], hy = hG[pt * 3 + 1], hz = hG[pt * 3 + 2];
    float w[8], dx[8], dy[8], dz[8], D[8], ex[8], ey[8], ez[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float sgx = (c & 1) ? 1.f : -1.f, sgy = (c & 2) ? 1.f : -1.f, sgz = (c & 4) ? 1.f : -1.f;
      const float wx = (c & 1) ? ax.w1 : ax.w0, sx = sgx * ax.dw, cx = sgx * ax.d2w;
      const float wy = (c & 2) ? ay.w1 : ay.w0, sy = sgy * ay.dw, cy = sgy * ay.d2w;
      const float wz = (c & 4) ? az.w1 : az.w0, sz = sgz * az.dw, cz = sgz * az.d2w;
      w[c] = wx * wy * wz;
      dx[c] = sx * wy * wz;
      dy[c] = wx * sy * wz;
      dz[c] = wx * wy * sz;
      D[c] = hx * dx[c] + hy * dy[c] + hz * dz[c];
      ex[c] = hx * (cx * wy * wz) + hy * (sx * sy * wz) + hz * (sx * wy * sz);
      ey[c] = hx * (sx * sy * wz) + hy * (wx * cy * wz) + hz * (wx * sy * sz);
      ez[c] = hx * (sx * wy * sz) + hy * (wx * sy * sz) + hz * (wx * wy * cz);
    }
    const float* base = in + n * v.sn;
    const float* hbase = hV ? hV + n * v.sn : nullptr;
    float* gbase = gin2 ? gin2 + n * v.sn : nullptr;
    const float* gobase = gout + n * pd.o_sn + q * pd.o_sp;
    float* ggobase = ggout + n * pd.o_sn + q * pd.o_sp;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    const int C = (int)v.c;
    for (int pb = 0; pb < C; pb += 4 * LPP) {
      const int ch4 = (pb >> 2) + sub;
      const bool lane_on = ch4 < c4;
      const float4 go = lane_on ? *reinterpret_cast<const float4*>(gobase + 4 * ch4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
      float got[4];
      if (ATOM == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ch = pb + j * LPP + sub;
          got[j] = ch < C ? gobase[ch] : 0.f;
        }
      }
      float4 ggo = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (g.inb[c]) {
          if (lane_on) {
            const int o = g.off[c] + 4 * ch4;
            const float4 val = *reinterpret_cast<const float4*>(base + o);
            const float4 hv = hbase ? *reinterpret_cast<const float4*>(hbase + o)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
            ggo.x += val.x * D[c] + hv.x * w[c];
            ggo.y += val.y * D[c] + hv.y * w[c];
            ggo.z += val.z * D[c] + hv.z * w[c];
            ggo.w += val.w * D[c] + hv.w * w[c];
            const float dv = dot4(go, val), dh = dot4(go, hv);
            gx += dh * dx[c] + dv * ex[c];
            gy += dh * dy[c] + dv * ey[c];
            gz += dh * dz[c] + dv * ez[c];
            if (ATOM == 1 && gbase && valid) {
              float* dst = gbase + o;
              unsafeAtomicAdd(dst + 0, go.x * D[c]);
              unsafeAtomicAdd(dst + 1, go.y * D[c]);
              unsafeAtomicAdd(dst + 2, go.z * D[c]);
              unsafeAtomicAdd(dst + 3, go.w * D[c]);
            }
          }
          if (ATOM == 2 && gbase && valid) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int ch = pb + j * LPP + sub;
              if (ch < C) unsafeAtomicAdd(gbase + g.off[c] + ch, got[j] * D[c]);
            }
          }
        }
      }
      if (valid && lane_on) *reinterpret_cast<float4*>(ggobase + 4 * ch4) = ggo;
    }
    gx = group_sum<LPP>(gx);
    gy = group_sum<LPP>(gy);
    gz = group_sum<LPP>(gz);
    if (sub == 0 && valid) {
      ggrid2[pt * 3 + 0] = gx;
      ggrid2[pt * 3 + 1] = gy;
      ggrid2[pt * 3 + 2] = gz;
    }
  }
}
#undef PV2_VEC_PROLOGUE

// Volume-gradient scatter on its own (mode 3): G consecutive lanes own one point and walk its
// channels one per lane, so every atomic instruction covers whole contiguous channel runs.
// coefficient per corner: w_c (first order) or D_c = sum_a hG_a d_a w_c (second order).
template <int G, bool SECOND>
__global__ __launch_bounds__(256) void tri_scatter_kernel(
    const float* __restrict__ gout, pv2_volume_desc v, const float* __restrict__ grid,
    const float* __restrict__ hG, pv2_points_desc pd, float* __restrict__ gin, int padding,
    int align, int smooth, uint32_t perm) {
  constexpr int PPW = 64 / G;
  const int lane = threadIdx.x & 63, sub = lane % G, slot = lane / G;
  const int64_t nw = (int64_t)gridDim.x * 4;
  const int C = (int)v.c;
  for (int64_t p0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * PPW; p0 < pd.n_points;
       p0 += nw * PPW) {
    int64_t pt = p0 + slot;
    if (pt >= pd.n_points) continue;
    if (perm) pt = (int64_t)((uint64_t)pt * perm % (uint64_t)pd.n_points);
    const int64_t n = pt / pd.points_per_n, q = pt % pd.points_per_n;
    const Axis<float> ax = make_axis<float>(grid[pt * 3 + 0], v.w, padding, align != 0, smooth != 0);
    const Axis<float> ay = make_axis<float>(grid[pt * 3 + 1], v.h, padding, align != 0, smooth != 0);
    const Axis<float> az = make_axis<float>(grid[pt * 3 + 2], v.d, padding, align != 0, smooth != 0);
    Geom32 g;
    make_geom32(ax, ay, az, v, &g);
    float coef[8];
    float hx = 0.f, hy = 0.f, hz = 0.f;
    if (SECOND) {
      hx = hG[pt * 3 + 0];
      hy = hG[pt * 3 + 1];
      hz = hG[pt * 3 + 2];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float wx = (c & 1) ? ax.w1 : ax.w0, sx = (c & 1) ? ax.dw : -ax.dw;
      const float wy = (c & 2) ? ay.w1 : ay.w0, sy = (c & 2) ? ay.dw : -ay.dw;
      const float wz = (c & 4) ? az.w1 : az.w0, sz = (c & 4) ? az.dw : -az.dw;
      coef[c] = SECOND ? hx * (sx * wy * wz) + hy * (wx * sy * wz) + hz * (wx * wy * sz)
                       : wx * wy * wz;
    }
    float* gbase = gin + n * v.sn;
    const float* gobase = gout + n * pd.o_sn + q * pd.o_sp;
    for (int ch = sub; ch < C; ch += G) {
      const float go = gobase[ch];
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (g.inb[c]) unsafeAtomicAdd(gbase + g.off[c] + ch, coef[c] * go);
    }
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// lanes per point for the vector path, 0 if the layout does not qualify
inline int vec_lanes(const pv2_volume_desc& v, const pv2_points_desc& pd, const void* vol_ptr,
                     const void* pts_ptr, const void* extra0, const void* extra1) {
  const bool ok = v.sc == 1 && pd.o_sc == 1 && (v.c % 4) == 0 && (v.sn % 4) == 0 &&
                  (v.sd % 4) == 0 && (v.sh % 4) == 0 && (v.sw % 4) == 0 && (pd.o_sn % 4) == 0 &&
                  (pd.o_sp % 4) == 0 && v.sn < 0x7fffffffLL && aligned16(vol_ptr) &&
                  aligned16(pts_ptr) && aligned16(extra0) && aligned16(extra1);
  if (!ok) return 0;
  const int64_t c4 = v.c / 4;
  return c4 <= 8 ? 8 : c4 <= 16 ? 16 : c4 <= 32 ? 32 : 64;
}

// How the volume-gradient atomics are issued, measured on MI355X (tools/bench_sampler.py,
// profiles/r01_sampler_modes.txt): their cost follows the number of separate contiguous runs an
// atomic instruction touches, so every lane group should cover >= 128 contiguous bytes.
//   1  from the float4 lanes (16-byte lane stride)        - 2-3x slower, kept for the comparison
//   2  transposed inside the vector kernel (runs of LPP*4 bytes per group): best when LPP >= 32
//   3  separate tri_scatter_kernel with 32/64 lanes per point: best for narrow volumes (C <= 64)
// Default (4) picks 2 or 3 by LPP.  PV2_TRI_MODE=0..3 forces one (0 = generic kernels only).
inline int tri_mode() {
  static const int m = [] {
    const char* e = getenv("PV2_TRI_MODE");
    return (e != nullptr && e[0] >= '0' && e[0] <= '3') ? e[0] - '0' : 4;
  }();
  return m;
}
inline int atomics_mode(int lpp) { return tri_mode() == 4 ? (lpp >= 32 ? 2 : 3) : tri_mode(); }

// The kernels that issue volume-gradient atomics visit the points in a scrambled order
// pt = (i * a) mod n (a prime that does not divide n: a bijection).  In ray order, the lane groups
// of one wave and the waves in flight together hold consecutive samples of the same few rays,
// which share corner voxels - their atomics then pile up on the same addresses.  PV2_TRI_PERMUTE=0
// keeps ray order (for the comparison in profiles/).
inline uint32_t point_permutation(int64_t n_points) {
  static const bool on = [] {
    const char* e = getenv("PV2_TRI_PERMUTE");
    return !(e != nullptr && e[0] == '0');
  }();
  if (!on || n_points < 64 || n_points >= 0x7fffffffLL) return 0;
  for (uint32_t a : {7919u, 7907u, 7901u, 7883u})
    if (n_points % a != 0) return a;
  return 0;
}

inline int vec_grid(int64_t n_points, int lpp) {
  const int64_t waves = (n_points + (64 / lpp) - 1) / (64 / lpp);
  return pv2::grid_for(waves * 64, 256);
}

#define PV2_VEC_DISPATCH(LPP_VALUE, KERNEL, ...)                                                 \
  switch (LPP_VALUE) {                                                                           \
    case 8: hipLaunchKernelGGL((KERNEL<8>), __VA_ARGS__); break;                                 \
    case 16: hipLaunchKernelGGL((KERNEL<16>), __VA_ARGS__); break;                               \
    case 32: hipLaunchKernelGGL((KERNEL<32>), __VA_ARGS__); break;                               \
    default: hipLaunchKernelGGL((KERNEL<64>), __VA_ARGS__); break;                               \
  }
#define PV2_VEC_DISPATCH_ATOM(LPP_VALUE, ATOM_VALUE, KERNEL, ...)                                \
  switch ((LPP_VALUE) * 4 + (ATOM_VALUE)) {                                                      \
    case 8 * 4 + 0: hipLaunchKernelGGL((KERNEL<8, 0>), __VA_ARGS__); break;                      \
    case 8 * 4 + 1: hipLaunchKernelGGL((KERNEL<8, 1>), __VA_ARGS__); break;                      \
    case 8 * 4 + 2: hipLaunchKernelGGL((KERNEL<8, 2>), __VA_ARGS__); break;                      \
    case 16 * 4 + 0: hipLaunchKernelGGL((KERNEL<16, 0>), __VA_ARGS__); break;                    \
    case 16 * 4 + 1: hipLaunchKernelGGL((KERNEL<16, 1>), __VA_ARGS__); break;                    \
    case 16 * 4 + 2: hipLaunchKernelGGL((KERNEL<16, 2>), __VA_ARGS__); break;                    \
    case 32 * 4 + 0: hipLaunchKernelGGL((KERNEL<32, 0>), __VA_ARGS__); break;                    \
    case 32 * 4 + 1: hipLaunchKernelGGL((KERNEL<32, 1>), __VA_ARGS__); break;                    \
    case 32 * 4 + 2: hipLaunchKernelGGL((KERNEL<32, 2>), __VA_ARGS__); break;                    \
    case 64 * 4 + 0: hipLaunchKernelGGL((KERNEL<64, 0>), __VA_ARGS__); break;                    \
    case 64 * 4 + 1: hipLaunchKernelGGL((KERNEL<64, 1>), __VA_ARGS__); break;                    \
    default: hipLaunchKernelGGL((KERNEL<64, 2>), __VA_ARGS__); break;                            \
  }

// volume-gradient scatter as its own launch; G = lanes per point
template <bool SECOND>
void launch_scatter(const float* gout, const pv2_volume_desc& v, const float* grid,
                    const float* hG, const pv2_points_desc& pd, float* gin, int padding, int align,
                    int smooth, hipStream_t s) {
  if (v.c <= 32) {
    hipLaunchKernelGGL((tri_scatter_kernel<32, SECOND>), dim3(vec_grid(pd.n_points, 32)),
                       dim3(256), 0, s, gout, v, grid, hG, pd, gin, padding, align, smooth,
                       point_permutation(pd.n_points));
  } else {
    hipLaunchKernelGGL((tri_scatter_kernel<64, SECOND>), dim3(vec_grid(pd.n_points, 64)),
                       dim3(256), 0, s, gout, v, grid, hG, pd, gin, padding, align, smooth,
                       point_permutation(pd.n_points));
  }
}

int check_desc(const pv2_volume_desc* vol, const pv2_points_desc* pts) {
  PV2_REQUIRE(vol != nullptr && pts != nullptr, "trilinear: null descriptor");
  PV2_REQUIRE(vol->c >= 1 && vol->d >= 1 && vol->h >= 1 && vol->w >= 1, "trilinear: empty volume");
  PV2_REQUIRE(pts->points_per_n >= 1 || pts->n_points == 0, "trilinear: bad points_per_n");
  return PV2_OK;
}

template <typename T>
int run_fwd(const T* input, const pv2_volume_desc* vol, const T* grid, const pv2_points_desc* pts,
            T* output, int padding, int align, int smooth, pv2_stream_t stream) {
  if (int e = check_desc(vol, pts)) return e;
  PV2_REQUIRE(padding >= 0 && padding <= 2, "trilinear: padding_mode must be 0, 1 or 2");
  if (pts->n_points == 0) return PV2_OK;
  if constexpr (std::is_same<T, float>::value) {
    if (const int lpp = tri_mode() ? vec_lanes(*vol, *pts, input, output, nullptr, nullptr) : 0) {
      PV2_VEC_DISPATCH(lpp, tri_fwd_vec_kernel, dim3(vec_grid(pts->n_points, lpp)), dim3(256), 0,
                       (hipStream_t)stream, input, *vol, grid, *pts, output, padding, align, smooth, 0u)
      return pv2::check_launch("trilinear_forward");
    }
  }
  hipLaunchKernelGGL((tri_fwd_kernel<T>), dim3(pv2::grid_for(pts->n_points * 64, 256)), dim3(256),
                     0, (hipStream_t)stream, input, *vol, grid, *pts, output, padding, align,
                     smooth);
  return pv2::check_launch("trilinear_forward");
}

template <typename T>
int run_bwd(const T* gout, const T* input, const pv2_volume_desc* vol, const T* grid,
            const pv2_points_desc* pts, T* gin, T* ggrid, int padding, int align, int smooth,
            pv2_stream_t stream) {
  if (int e = check_desc(vol, pts)) return e;
  PV2_REQUIRE(padding >= 0 && padding <= 2, "trilinear: padding_mode must be 0, 1 or 2");
  if (pts->n_points == 0) return PV2_OK;
  if constexpr (std::is_same<T, float>::value) {
    if (const int lpp = tri_mode() ? vec_lanes(*vol, *pts, input, gout, gin, nullptr) : 0) {
      const int mode = atomics_mode(lpp);
      const int atom = (gin == nullptr || mode == 3) ? 0 : mode;
      PV2_VEC_DISPATCH_ATOM(lpp, atom, tri_bwd_vec_kernel, dim3(vec_grid(pts->n_points, lpp)),
                            dim3(256), 0, (hipStream_t)stream, gout, input, *vol, grid, *pts, gin,
                            ggrid, padding, align, smooth, atom ? point_permutation(pts->n_points) : 0u)
      if (gin != nullptr && mode == 3)
        launch_scatter<false>(gout, *vol, grid, nullptr, *pts, gin, padding, align, smooth,
                              (hipStream_t)stream);
      return pv2::check_launch("trilinear_backward");
    }
  }
  hipLaunchKernelGGL((tri_bwd_kernel<T>), dim3(pv2::grid_for(pts->n_points * 64, 256)), dim3(256),
                     0, (hipStream_t)stream, gout, input, *vol, grid, *pts, gin, ggrid, padding,
                     align, smooth);
  return pv2::check_launch("trilinear_backward");
}

template <typename T>
int run_bwdbwd(const T* hV, const T* hG, const T* input, const pv2_volume_desc* vol, const T* grid,
               const T* gout, const pv2_points_desc* pts, T* gin2, T* ggrid2, T* ggout,
               int padding, int align, int smooth, pv2_stream_t stream) {
  if (int e = check_desc(vol, pts)) return e;
  PV2_REQUIRE(padding >= 0 && padding <= 2, "trilinear: padding_mode must be 0, 1 or 2");
  if (pts->n_points == 0) return PV2_OK;
  if constexpr (std::is_same<T, float>::value) {
    if (const int lpp = tri_mode() ? vec_lanes(*vol, *pts, input, gout, hV, gin2) : 0) {
      if (aligned16(ggout)) {
        const int mode = atomics_mode(lpp);
        const int atom = (gin2 == nullptr || mode == 3) ? 0 : mode;
        PV2_VEC_DISPATCH_ATOM(lpp, atom, tri_bwdbwd_vec_kernel,
                              dim3(vec_grid(pts->n_points, lpp)), dim3(256), 0,
                              (hipStream_t)stream, hV, hG, input, *vol, grid, gout, *pts, gin2,
                              ggrid2, ggout, padding, align, smooth,
                              atom ? point_permutation(pts->n_points) : 0u)
        if (gin2 != nullptr && mode == 3)
          launch_scatter<true>(gout, *vol, grid, hG, *pts, gin2, padding, align, smooth,
                               (hipStream_t)stream);
        return pv2::check_launch("trilinear_backward_backward");
      }
    }
  }
  hipLaunchKernelGGL((tri_bwdbwd_kernel<T>), dim3(pv2::grid_for(pts->n_points * 64, 256)),
                     dim3(256), 0, (hipStream_t)stream, hV, hG, input, *vol, grid, gout, *pts,
                     gin2, ggrid2, ggout, padding, align, smooth);
  return pv2::check_launch("trilinear_backward_backward");
}

// 16-bit storage (dtype: 1 = bfloat16, 2 = float16, the codes of the other mixed-precision entry
// points), fp32 arithmetic; grad_input / grad_input2 are FP32 accumulation buffers.
template <typename S>
int run_fwd16(const void* input, const pv2_volume_desc* vol, const void* grid,
              const pv2_points_desc* pts, void* output, int padding, int align, int smooth,
              pv2_stream_t stream) {
  hipLaunchKernelGGL((tri_fwd_kernel<float, S>), dim3(pv2::grid_for(pts->n_points * 64, 256)),
                     dim3(256), 0, (hipStream_t)stream, (const S*)input, *vol, (const S*)grid, *pts,
                     (S*)output, padding, align, smooth);
  return pv2::check_launch("trilinear_forward_16");
}
template <typename S>
int run_bwd16(const void* gout, const void* input, const pv2_volume_desc* vol, const void* grid,
              const pv2_points_desc* pts, float* gin, void* ggrid, int padding, int align,
              int smooth, pv2_stream_t stream) {
  hipLaunchKernelGGL((tri_bwd_kernel<float, S>), dim3(pv2::grid_for(pts->n_points * 64, 256)),
                     dim3(256), 0, (hipStream_t)stream, (const S*)gout, (const S*)input, *vol,
                     (const S*)grid, *pts, gin, (S*)ggrid, padding, align, smooth);
  return pv2::check_launch("trilinear_backward_16");
}
template <typename S>
int run_bwdbwd16(const void* hV, const void* hG, const void* input, const pv2_volume_desc* vol,
                 const void* grid, const void* gout, const pv2_points_desc* pts, float* gin2,
                 void* ggrid2, void* ggout, int padding, int align, int smooth,
                 pv2_stream_t stream) {
  hipLaunchKernelGGL((tri_bwdbwd_kernel<float, S>), dim3(pv2::grid_for(pts->n_points * 64, 256)),
                     dim3(256), 0, (hipStream_t)stream, (const S*)hV, (const S*)hG,
                     (const S*)input, *vol, (const S*)grid, (const S*)gout, *pts, gin2,
                     (S*)ggrid2, (S*)ggout, padding, align, smooth);
  return pv2::check_launch("trilinear_backward_backward_16");
}

int check16(const pv2_volume_desc* vol, const pv2_points_desc* pts, int dtype, int padding) {
  if (int e = check_desc(vol, pts)) return e;
  PV2_REQUIRE(dtype == 1 || dtype == 2, "trilinear: 16-bit dtype must be 1 (bf16) or 2 (f16)");
  PV2_REQUIRE(padding >= 0 && padding <= 2, "trilinear: padding_mode must be 0, 1 or 2");
  return PV2_OK;
}

}  // namespace

extern "C" {

int pv2_trilinear_forward_16(const void* input, int dtype, const pv2_volume_desc* vol,
                             const void* grid, const pv2_points_desc* pts, void* output,
                             int padding_mode, int align_corners, int apply_smoothstep,
                             pv2_stream_t stream) {
  if (int e = check16(vol, pts, dtype, padding_mode)) return e;
  if (pts->n_points == 0) return PV2_OK;
  return dtype == 1 ? run_fwd16<B16>(input, vol, grid, pts, output, padding_mode, align_corners,
                                     apply_smoothstep, stream)
                    : run_fwd16<H16>(input, vol, grid, pts, output, padding_mode, align_corners,
                                     apply_smoothstep, stream);
}
int pv2_trilinear_backward_16(const void* grad_output, const void* input, int dtype,
                              const pv2_volume_desc* vol, const void* grid,
                              const pv2_points_desc* pts, float* grad_input_f32, void* grad_grid,
                              int padding_mode, int align_corners, int apply_smoothstep,
                              pv2_stream_t stream) {
  if (int e = check16(vol, pts, dtype, padding_mode)) return e;
  if (pts->n_points == 0) return PV2_OK;
  return dtype == 1 ? run_bwd16<B16>(grad_output, input, vol, grid, pts, grad_input_f32, grad_grid,
                                     padding_mode, align_corners, apply_smoothstep, stream)
                    : run_bwd16<H16>(grad_output, input, vol, grid, pts, grad_input_f32, grad_grid,
                                     padding_mode, align_corners, apply_smoothstep, stream);
}
int pv2_trilinear_backward_backward_16(const void* g_ginput, const void* g_ggrid, const void* input,
                                       int dtype, const pv2_volume_desc* vol, const void* grid,
                                       const void* grad_output, const pv2_points_desc* pts,
                                       float* grad_input2_f32, void* grad_grid2,
                                       void* grad_grad_output, int padding_mode, int align_corners,
                                       int apply_smoothstep, pv2_stream_t stream) {
  if (int e = check16(vol, pts, dtype, padding_mode)) return e;
  if (pts->n_points == 0) return PV2_OK;
  return dtype == 1
             ? run_bwdbwd16<B16>(g_ginput, g_ggrid, input, vol, grid, grad_output, pts,
                                 grad_input2_f32, grad_grid2, grad_grad_output, padding_mode,
                                 align_corners, apply_smoothstep, stream)
             : run_bwdbwd16<H16>(g_ginput, g_ggrid, input, vol, grid, grad_output, pts,
                                 grad_input2_f32, grad_grid2, grad_grad_output, padding_mode,
                                 align_corners, apply_smoothstep, stream);
}

int pv2_trilinear_forward_f32(const float* input, const pv2_volume_desc* vol, const float* grid,
                              const pv2_points_desc* pts, float* output, int padding_mode,
                              int align_corners, int apply_smoothstep, pv2_stream_t stream) {
  return run_fwd<float>(input, vol, grid, pts, output, padding_mode, align_corners,
                        apply_smoothstep, stream);
}
int pv2_trilinear_forward_f64(const double* input, const pv2_volume_desc* vol, const double* grid,
                              const pv2_points_desc* pts, double* output, int padding_mode,
                              int align_corners, int apply_smoothstep, pv2_stream_t stream) {
  return run_fwd<double>(input, vol, grid, pts, output, padding_mode, align_corners,
                         apply_smoothstep, stream);
}
int pv2_trilinear_backward_f32(const float* grad_output, const float* input,
                               const pv2_volume_desc* vol, const float* grid,
                               const pv2_points_desc* pts, float* grad_input, float* grad_grid,
                               int padding_mode, int align_corners, int apply_smoothstep,
                               pv2_stream_t stream) {
  return run_bwd<float>(grad_output, input, vol, grid, pts, grad_input, grad_grid, padding_mode,
                        align_corners, apply_smoothstep, stream);
}
int pv2_trilinear_backward_f64(const double* grad_output, const double* input,
                               const pv2_volume_desc* vol, const double* grid,
                               const pv2_points_desc* pts, double* grad_input, double* grad_grid,
                               int padding_mode, int align_corners, int apply_smoothstep,
                               pv2_stream_t stream) {
  return run_bwd<double>(grad_output, input, vol, grid, pts, grad_input, grad_grid, padding_mode,
                         align_corners, apply_smoothstep, stream);
}
int pv2_trilinear_backward_backward_f32(const float* g_ginput, const float* g_ggrid,
                                        const float* input, const pv2_volume_desc* vol,
                                        const float* grid, const float* grad_output,
                                        const pv2_points_desc* pts, float* grad_input2,
                                        float* grad_grid2, float* grad_grad_output,
                                        int padding_mode, int align_corners, int apply_smoothstep,
                                        pv2_stream_t stream) {
  return run_bwdbwd<float>(g_ginput, g_ggrid, input, vol, grid, grad_output, pts, grad_input2,
                           grad_grid2, grad_grad_output, padding_mode, align_corners,
                           apply_smoothstep, stream);
}
int pv2_trilinear_backward_backward_f64(const double* g_ginput, const double* g_ggrid,
                                        const double* input, const pv2_volume_desc* vol,
                                        const double* grid, const double* grad_output,
                                        const pv2_points_desc* pts, double* grad_input2,
                                        double* grad_grid2, double* grad_grad_output,
                                        int padding_mode, int align_corners, int apply_smoothstep,
                                        pv2_stream_t stream) {
  return run_bwdbwd<double>(g_ginput, g_ggrid, input, vol, grid, grad_output, pts, grad_input2,
                            grad_grid2, grad_grad_output, padding_mode, align_corners,
                            apply_smoothstep, stream);
}

}  // extern "C"
