// Sparse 3-D convolution with 16-bit operands (bf16 / fp16) on the gfx950 matrix cores, fp32
// accumulation: the arithmetic of the reference's shipped training mode (enable_amp = True,
// configs/scannet/pretrain-ponder-spunet-v1m1-0-base.py:12; autocast around the model call,
// ponder/engines/train.py:183-196) for the same call sites as sparse_conv.hip
// (ponder/models/sparse_unet/spconv_unet_v1m1_base.py:41,47,58,112,135,171).
//
// v_mfma_f32_32x32x16_{bf16,f16} runs at 16x the fp32 MFMA rate, so the balance of the kernels is
// the opposite of the fp32 ones: arithmetic is nearly free, bytes and latency are what is paid for.
// Hence
//   * forward and grad-input are OUTPUT-STATIONARY (gather table, no atomics, no zero-fill, every
//     element written once, in 16 bits, bitwise reproducible): the wasted MFMA work on absent
//     neighbours that rules this form out in fp32 costs little here;
//   * the weights are re-packed once per optimiser step (pv2_spconv16_pack_weights: the fp32 master
//     weights have to be cast anyway) into MFMA-fragment order, so that a B fragment is ONE fully
//     coalesced 1 KiB wave load - no LDS staging, no barriers in the main loop;
//   * the weight gradient stays a pair-major reduction (fp32 atomics on dW once per workgroup) with
//     both operands staged through LDS, where the 16-bit MFMA's reduction axis (8 consecutive
//     PAIRS per lane) is assembled by 2-byte LDS reads.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct BF16 {
  static __device__ __forceinline__ f32x16 mfma(const u16x8& a, const u16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ unsigned short from_float(float v) {
    return __builtin_bit_cast(unsigned short, (__bf16)v);  // round to nearest even
  }
};

struct F16 {
  static __device__ __forceinline__ f32x16 mfma(const u16x8& a, const u16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ unsigned short from_float(float v) {
    return __builtin_bit_cast(unsigned short, (_Float16)v);
  }
};

__device__ __forceinline__ u16x8 ld8(const unsigned short* __restrict__ p, bool ok) {
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (ok) v = *reinterpret_cast<const uint4*>(p);
  return __builtin_bit_cast(u16x8, v);
}

__device__ __forceinline__ int find_offset(const int32_t* __restrict__ tile_start, int K,
                                           int tile) {
  int lo = 0, hi = K;  // invariant: tile_start[lo] <= tile < tile_start[hi]
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (tile_start[mid] <= tile) lo = mid; else hi = mid;
  }
  return lo;
}

// ------------------------------------------------------------------------------------------
// Weight packing.  Fragment order: for offset k, 32-row block rb and 16-wide reduction slice sl,
// the 64 lanes' 8-element operands are contiguous:
//     packed[((k * n_rb + rb) * n_sl + sl) * 512 + lane * 8 + j] = M_k[rb*32 + (lane & 31)][sl*16 + 8*(lane >> 5) + j]
// with M_k = W[:, k, :] ("rows" = output channels, reduction = input channels: the forward pass)
// or its transpose (rows = input channels, reduction = output channels: the grad-input pass).
// Rows / reduction indices past the matrix are zero.
template <typename T>
__global__ __launch_bounds__(256) void pack_weights16_kernel(
    const float* __restrict__ W, int c_out, int K, int c_in, unsigned short* __restrict__ fwd,
    unsigned short* __restrict__ bwd) {
  const int rb_f = (c_out + 31) / 32, sl_f = (c_in + 15) / 16;
  const int rb_b = (c_in + 31) / 32, sl_b = (c_out + 15) / 16;
  const int64_t n_f = (int64_t)K * rb_f * sl_f * 512, n_b = (int64_t)K * rb_b * sl_b * 512;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_f + n_b; e += stride) {
    const bool f = e < n_f;
    const int64_t q = f ? e : e - n_f;
    const int n_rb = f ? rb_f : rb_b, n_sl = f ? sl_f : sl_b;
    const int j = (int)(q & 7), lane = (int)((q >> 3) & 63);
    const int64_t frag = q >> 9;
    const int sl = (int)(frag % n_sl), rb = (int)((frag / n_sl) % n_rb), k = (int)(frag / ((int64_t)n_sl * n_rb));
    const int row = rb * 32 + (lane & 31), red = sl * 16 + 8 * (lane >> 5) + j;
    const int n = f ? row : red, c = f ? red : row;
    const float v = (n < c_out && c < c_in) ? W[((int64_t)n * K + k) * c_in + c] : 0.f;
    (f ? fwd : bwd)[q] = T::from_float(v);
  }
}

// ------------------------------------------------------------------------------------------
// Output-stationary conv:  Y[o, n] = bias[n] + sum_k sum_c X[nbr[k][o], c] * W[n, kw(k), c]
//
// A workgroup of NW waves owns 32*RB output rows (taken in the order `perm`) x 32*NB output
// channels.  The tile's slice of the gather table goes to LDS once, with the set of offsets that
// at least one of its rows has.  The (present offset, reduction chunk) steps are dealt round-robin
// to the waves; per step a wave works through rounds of U 16-wide reduction slices: A = its
// gathered rows (global -> registers, 16 bytes per lane, RB row blocks), B = the packed weight
// fragments of the offset (one coalesced 1 KiB load each, shared by the RB row blocks).  The NW
// partial tiles are added in LDS in the fixed order wave 0..NW-1 and stored as 16-byte pieces of 8
// channels.  What bounds it: every tile streams all present offsets' weights from L2, so the
// traffic is (n_out / (32 RB)) * K * c_in * c_out * 2 bytes - RB = 2 halves it.
constexpr int kMaxK16 = 128;  // offsets per conv (5^3 = 125)

template <typename T, int NB, int U, int RB, int NW>
__global__ __launch_bounds__(64 * NW) void spconv_os16_kernel(
    const unsigned short* __restrict__ X, int c_in, const uint4* __restrict__ Wp, int K, int c_out,
    const int32_t* __restrict__ nbr, int64_t nbr_stride, const int32_t* __restrict__ perm,
    int kflip, const float* __restrict__ bias, int64_t n_out, unsigned short* __restrict__ Y) {
  constexpr int NT = 32 * NB, LD = NT + 4, TR = 32 * RB, NTH = 64 * NW;
  __shared__ __attribute__((aligned(16))) float tile[TR * LD];
  __shared__ int s_row[TR];
  __shared__ int s_idx[kMaxK16 * TR];
  __shared__ unsigned s_pres[4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 31, h = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * TR;
  if (tid < 4) s_pres[tid] = 0u;
  for (int t = tid; t < TR; t += NTH) {
    const int64_t r = row0 + t;
    s_row[t] = r < n_out ? (perm ? perm[r] : (int)r) : -1;
  }
  __syncthreads();
  // gather table -> LDS (each wave-load reads 64 consecutive rows of one offset when perm is null)
  for (int e = tid; e < K * TR; e += NTH) {
    const int k = e / TR, t = e % TR;
    const int o = s_row[t];
    const int idx = o >= 0 ? nbr[(int64_t)k * nbr_stride + o] : -1;
    s_idx[e] = idx;
    if (idx >= 0) atomicOr(&s_pres[k >> 5], 1u << (k & 31));
  }
  __syncthreads();
  unsigned pres[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) pres[q] = __builtin_amdgcn_readfirstlane(s_pres[q]);

  const int nb0 = blockIdx.y * NB;
  const int n_rb = (c_out + 31) / 32, n_sl = (c_in + 15) / 16;
  // with few offsets (1x1 convs, the 8 children of a strided conv) the reduction axis is cut into
  // 32-channel chunks so that every wave has work
  const int spc = K >= 4 ? n_sl : 2;
  const int n_chunk = (n_sl + spc - 1) / spc;

  f32x16 acc[RB][NB];
#pragma unroll
  for (int j = 0; j < RB; ++j)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[j][nb][q] = 0.f;

  int step = 0;
  for (int k = 0; k < K; ++k) {
    if (!((pres[k >> 5] >> (k & 31)) & 1u)) continue;  // no row of the tile has this offset
    if (n_chunk == 1 && (step % NW) != wave) {          // another wave's offset
      ++step;
      continue;
    }
    const unsigned short* xrow[RB];
    bool pv[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int idx = s_idx[k * TR + j * 32 + i];
      pv[j] = idx >= 0;
      xrow[j] = X + (int64_t)(pv[j] ? idx : 0) * c_in + 8 * h;
    }
    const int kw = kflip ? K - 1 - k : k;
    const uint4* wk = Wp + ((int64_t)kw * n_rb + nb0) * n_sl * 64 + lane;
    for (int ch = 0; ch < n_chunk; ++ch) {
      if ((step++ % NW) != wave) continue;  // another wave's step
      const int sl1 = min(n_sl, (ch + 1) * spc);
      // U slices per round: their U * (RB + NB) loads are all in flight before the first MFMA
      for (int sl = ch * spc; sl < sl1; sl += U) {
        u16x8 a[U][RB], b[U][NB];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool in = sl + u < sl1;
#pragma unroll
          for (int j = 0; j < RB; ++j)
            a[u][j] = ld8(xrow[j] + (sl + u) * 16, in && pv[j] && (sl + u) * 16 + 8 * h + 8 <= c_in);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            uint4 w = make_uint4(0u, 0u, 0u, 0u);
            if (in && nb0 + nb < n_rb) w = wk[((int64_t)nb * n_sl + sl + u) * 64];
            b[u][nb] = __builtin_bit_cast(u16x8, w);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int j = 0; j < RB; ++j)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[j][nb] = T::mfma(a[u][j], b[u][nb], acc[j][nb]);
      }
    }
  }

  // the partial tiles meet in LDS, added in the fixed order wave 0 .. NW-1
#pragma unroll 1
  for (int w = 0; w < NW; ++w) {
    if (wave == w) {
#pragma unroll
      for (int j = 0; j < RB; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            float* p = &tile[(j * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * LD + nb * 32 + i];
            *p = w == 0 ? acc[j][nb][q] : *p + acc[j][nb][q];
          }
    }
    __syncthreads();
  }
  for (int e = tid; e < TR * (NT / 8); e += NTH) {
    const int row = e / (NT / 8), cg = e % (NT / 8);
    const int dst = s_row[row];
    const int n = nb0 * 32 + cg * 8;
    if (dst < 0 || n >= c_out) continue;
    const float4 v0 = *reinterpret_cast<const float4*>(&tile[row * LD + cg * 8]);
    const float4 v1 = *reinterpret_cast<const float4*>(&tile[row * LD + cg * 8 + 4]);
    float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    u16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = T::from_float(v[j] + (bias ? bias[n + j] : 0.f));
    *reinterpret_cast<uint4*>(Y + (int64_t)dst * c_out + n) = __builtin_bit_cast(uint4, o);
  }
}

// ------------------------------------------------------------------------------------------
// Weight gradient  dW[n, k, c] += sum_{pairs p of offset k} dY[out(p), n] * X[in(p), c]   (fp32 out)
//
// A workgroup owns `tile_pairs` pairs of ONE offset and a (64*WN) x (64*WC) block of dW[:, k, :].
// Per step kStep pairs are staged - their dY rows and X rows, the block's column ranges only,
// global -> registers -> LDS, double buffered.  Waves form a WN x WC x WK grid: each owns a 64x64
// sub-block and every WK-th 16-pair chunk of the step.  The MFMA wants 8 consecutive PAIRS of one
// channel per lane, i.e. the transpose of how rows arrive: the fragment is assembled from eight
// 2-byte LDS reads (consecutive lanes read consecutive channels: conflict-free) - or, TR = true, by
// gfx950's transposing read: ds_read_b64_tr_b16 hands lane c of a 16-lane group column c of the
// 4 x 16 block whose 16 8-byte pieces the group's lanes point at (lane t: row t / 4, columns
// 4 (t % 4) ..), i.e. four consecutive pairs of one channel in ONE LDS instruction where the scalar
// form needs four reads and two packs.  Rows are padded by 32 bytes so that the four rows of a
// block fall into different banks.
constexpr int kMaxWgradTile16 = 512;
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u16x8 tr_fragment(const unsigned short* p, int row_stride) {
  // p: this lane's piece of pairs +0..3; the second read takes pairs +4..7
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * row_stride));
  u16x8 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = (unsigned short)lo[j];
    v[4 + j] = (unsigned short)hi[j];
  }
  return v;
}

template <typename T, int WN, int WC, bool TR>
__global__ __launch_bounds__(256) void spconv_wgrad16_kernel(
    const unsigned short* __restrict__ X, int c_in, const unsigned short* __restrict__ dY, int c_out,
    int K, const int32_t* __restrict__ pair_in, const int32_t* __restrict__ pair_out,
    const int32_t* __restrict__ kstart, const int32_t* __restrict__ tile_start, int tile_pairs,
    int n_ntile, int n_ctile, float* __restrict__ dW) {
  constexpr int WK = 4 / (WN * WC);
  constexpr int kStep = WK == 4 ? 64 : 32;      // pairs staged per step
  constexpr int CW = kStep / 16 / WK;           // 16-pair chunks per wave per step
  constexpr int TN = 64 * WN, TC = 64 * WC;
  constexpr int UA = kStep * TN / 8 / 256, UB = kStep * TC / 8 / 256;  // 16-byte pieces per thread
  constexpr int LDA = TN + (TR ? 16 : 0), LDB = TC + (TR ? 16 : 0);     // LDS row strides
  __shared__ __attribute__((aligned(16))) unsigned short sA[2][kStep * LDA];
  __shared__ __attribute__((aligned(16))) unsigned short sB[2][kStep * LDB];
  __shared__ int s_in[kMaxWgradTile16];
  __shared__ int s_out[kMaxWgradTile16];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int per_tile = n_ntile * n_ctile;
  const int tile = blockIdx.x / per_tile, sub = blockIdx.x % per_tile;
  const int n0 = (sub / n_ctile) * TN, c0 = (sub % n_ctile) * TC;
  if (tile >= tile_start[K]) return;  // grid sized from an upper bound of the pair counts
  const int k = find_offset(tile_start, K, tile);
  const int p0 = kstart[k] + (tile - tile_start[k]) * tile_pairs;
  const int cnt = min(kstart[k + 1] - p0, tile_pairs);
  for (int t = tid; t < tile_pairs; t += 256) {
    s_in[t] = t < cnt ? pair_in[p0 + t] : -1;
    s_out[t] = t < cnt ? pair_out[p0 + t] : -1;
  }
  __syncthreads();

  const int wk = wave % WK, wc = (wave / WK) % WC, wn = wave / (WK * WC);
  const int i = lane & 31, h = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

  u16x8 ra[UA], rb[UB];
  auto load_step = [&](int s) {
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      const int q = tid + 256 * u;
      const int row = q / (TN / 8), col = (q % (TN / 8)) * 8;
      const int idx = s * kStep + row;
      const int o = idx < tile_pairs ? s_out[idx] : -1;
      ra[u] = ld8(dY + (int64_t)max(o, 0) * c_out + n0 + col, o >= 0 && n0 + col < c_out);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int q = tid + 256 * u;
      const int row = q / (TC / 8), col = (q % (TC / 8)) * 8;
      const int idx = s * kStep + row;
      const int r_in = idx < tile_pairs ? s_in[idx] : -1;
      rb[u] = ld8(X + (int64_t)max(r_in, 0) * c_in + c0 + col, r_in >= 0 && c0 + col < c_in);
    }
  };
  auto store_step = [&](int buf) {
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      const int q = tid + 256 * u;
      *reinterpret_cast<u16x8*>(&sA[buf][(q / (TN / 8)) * LDA + (q % (TN / 8)) * 8]) = ra[u];
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int q = tid + 256 * u;
      *reinterpret_cast<u16x8*>(&sB[buf][(q / (TC / 8)) * LDB + (q % (TC / 8)) * 8]) = rb[u];
    }
  };

  const int nsteps = (cnt + kStep - 1) / kStep;
  load_step(0);
  store_step(0);
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    const bool more = (s + 1) < nsteps;
    if (more) load_step(s + 1);
#pragma unroll
    for (int cw = 0; cw < CW; ++cw) {
      u16x8 a0, a1, b0, b1;
      if (TR) {
        // 16-lane group g: channels 16 (g & 1) .., pairs 8 (g >> 1) ..; lane t of the group points
        // at row t / 4, columns 4 (t % 4) .. of the group's 4 x 16 block
        const int g = lane >> 4, t = lane & 15;
        const int prow = (wk + WK * cw) * 16 + 8 * (g >> 1) + (t >> 2);
        const int pcol = 16 * (g & 1) + 4 * (t & 3);
        const unsigned short* A = &sA[buf][prow * LDA + wn * 64 + pcol];
        const unsigned short* B = &sB[buf][prow * LDB + wc * 64 + pcol];
        a0 = tr_fragment(A, LDA);
        a1 = tr_fragment(A + 32, LDA);
        b0 = tr_fragment(B, LDB);
        b1 = tr_fragment(B + 32, LDB);
      } else {
        const int pr = (wk + WK * cw) * 16 + 8 * h;  // first of this lane's 8 pairs
        const unsigned short* A = &sA[buf][pr * LDA + wn * 64 + i];
        const unsigned short* B = &sB[buf][pr * LDB + wc * 64 + i];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          a0[j] = A[j * LDA];
          a1[j] = A[j * LDA + 32];
          b0[j] = B[j * LDB];
          b1[j] = B[j * LDB + 32];
        }
      }
      acc[0][0] = T::mfma(a0, b0, acc[0][0]);
      acc[0][1] = T::mfma(a0, b1, acc[0][1]);
      acc[1][0] = T::mfma(a1, b0, acc[1][0]);
      acc[1][1] = T::mfma(a1, b1, acc[1][1]);
    }
    if (more) store_step(buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int n = n0 + wn * 64 + a * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
      if (n < c_out) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int c = c0 + wc * 64 + b * 32 + i;
          if (c < c_in) unsafeAtomicAdd(dW + ((int64_t)n * K + k) * c_in + c, acc[a][b][q]);
        }
      }
    }
}

int os16_variant();

template <typename T>
int launch_os16(const unsigned short* X, int c_in, const uint4* Wp, int K, int c_out,
                const int32_t* nbr, int64_t nbr_stride, const int32_t* perm, int kflip,
                const float* bias, int64_t n_out, unsigned short* Y, hipStream_t s) {
  if (K > kMaxK16) {
    pv2::set_error("pv2_spconv16_os_forward: more than 128 offsets");
    return PV2_E_UNSUPPORTED;
  }
  const int nblk = (c_out + 31) / 32;
  const int var = os16_variant();
  // 64-row tiles stream the weights half as often; they need enough rows to fill 256 CUs
  const bool rb2 = (var & 1) != 0 && n_out >= 64;
  const bool nw8 = (var & 2) != 0;
  const int64_t row_tiles = (n_out + (rb2 ? 63 : 31)) / (rb2 ? 64 : 32);
  int nb = nblk >= 4 ? 4 : nblk;
  if ((nb == 4 && row_tiles * ((nblk + 3) / 4) < 1024) || rb2) nb = nb >= 2 ? 2 : 1;
  if (nb == 3) nb = 4;
  const int groups = (nblk + nb - 1) / nb;
  if (row_tiles >= 0x7fffffffLL || groups >= 65536) {
    pv2::set_error("pv2_spconv16_os_forward: grid too large");
    return PV2_E_BADARG;
  }
  const dim3 grid((unsigned)row_tiles, (unsigned)groups);
  const int n_sl = (c_in + 15) / 16;
  const bool u4 = (n_sl % 4) == 0 && K >= 4 && !rb2;  // reduction slices per round of loads
#define PV2_OS16(NB_, U_, RB_, NW_)                                                            \
  hipLaunchKernelGGL((spconv_os16_kernel<T, NB_, U_, RB_, NW_>), grid, dim3(64 * NW_), 0, s, X, \
                     c_in, Wp, K, c_out, nbr, nbr_stride, perm, kflip, bias, n_out, Y)
#define PV2_OS16_NB(NB_)                                        \
  do {                                                          \
    if (rb2) {                                                  \
      if (nw8) PV2_OS16(NB_, 2, 2, 8); else PV2_OS16(NB_, 2, 2, 4); \
    } else if (u4) {                                            \
      if (nw8) PV2_OS16(NB_, 4, 1, 8); else PV2_OS16(NB_, 4, 1, 4); \
    } else {                                                    \
      if (nw8) PV2_OS16(NB_, 2, 1, 8); else PV2_OS16(NB_, 2, 1, 4); \
    }                                                           \
  } while (0)
  switch (nb) {
    case 1: PV2_OS16_NB(1); break;
    case 2: PV2_OS16_NB(2); break;
    default:
      if (nw8) PV2_OS16(4, 2, 1, 8); else PV2_OS16(4, 2, 1, 4);
      break;
  }
#undef PV2_OS16_NB
#undef PV2_OS16
  return pv2::check_launch("spconv16_os_forward");
}

template <typename T>
int launch_wgrad16(const unsigned short* X, int c_in, const unsigned short* dY, int c_out, int K,
                   const int32_t* pi, const int32_t* po, const int32_t* ks, const int32_t* ts,
                   int tile_pairs, int64_t n_tiles, float* dW, hipStream_t s) {
  const bool small = c_in <= 64 && c_out <= 64;
  const int tn = small ? 64 : 128, tc = small ? 64 : 128;
  const int n_ntile = (c_out + tn - 1) / tn, n_ctile = (c_in + tc - 1) / tc;
  const int64_t blocks = n_tiles * n_ntile * n_ctile;
  if (blocks >= 0x7fffffffLL) {
    pv2::set_error("pv2_spconv16_backward_weight: grid too large");
    return PV2_E_BADARG;
  }
  const bool tr = (os16_variant() & 8) == 0;  // knob bit 3: the scalar-read form
#define PV2_WGRAD16(WN_, WC_, TR_)                                                               \
  hipLaunchKernelGGL((spconv_wgrad16_kernel<T, WN_, WC_, TR_>), dim3((unsigned)blocks), dim3(256), \
                     0, s, X, c_in, dY, c_out, K, pi, po, ks, ts, tile_pairs, n_ntile, n_ctile, dW)
  if (small) {
    if (tr) PV2_WGRAD16(1, 1, true); else PV2_WGRAD16(1, 1, false);
  } else {
    if (tr) PV2_WGRAD16(2, 2, true); else PV2_WGRAD16(2, 2, false);
  }
#undef PV2_WGRAD16
  return pv2::check_launch("spconv16_backward_weight");
}

}  // namespace

// Tuning knob (tools/bench_spconv16.py): bit 0 = 64-row tiles, bit 1 = 8 waves per workgroup,
// bit 3 = weight-gradient fragments by scalar LDS reads instead of ds_read_b64_tr_b16.
static int g_os16_variant = 0;
namespace {
int os16_variant() { return g_os16_variant; }
}

extern "C" {

int pv2_debug_set_os16_variant(int v) {
  g_os16_variant = v;
  return PV2_OK;
}

int64_t pv2_spconv16_packed_elems(int rows, int K, int reduction) {
  return (int64_t)K * ((rows + 31) / 32) * ((reduction + 15) / 16) * 512;
}

int pv2_spconv16_pack_weights(const float* weight, int c_out, int K, int c_in, int dtype,
                              void* packed_fwd, void* packed_bwd, pv2_stream_t stream) {
  PV2_REQUIRE(dtype == PV2_BF16 || dtype == PV2_F16, "pv2_spconv16_pack_weights: dtype must be PV2_BF16 or PV2_F16");
  PV2_REQUIRE(c_in >= 1 && c_out >= 1 && K >= 1, "pv2_spconv16_pack_weights: bad sizes");
  const int64_t total = pv2_spconv16_packed_elems(c_out, K, c_in) + pv2_spconv16_packed_elems(c_in, K, c_out);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(pv2::grid_for(total, 256));
  if (dtype == PV2_BF16)
    hipLaunchKernelGGL((pack_weights16_kernel<BF16>), grid, dim3(256), 0, s, weight, c_out, K, c_in,
                       (unsigned short*)packed_fwd, (unsigned short*)packed_bwd);
  else
    hipLaunchKernelGGL((pack_weights16_kernel<F16>), grid, dim3(256), 0, s, weight, c_out, K, c_in,
                       (unsigned short*)packed_fwd, (unsigned short*)packed_bwd);
  return pv2::check_launch("spconv16_pack_weights");
}

int pv2_spconv16_os_forward(const void* in_feat, int64_t n_in, int c_in, const void* packed_weight,
                            int K, int c_out, int dtype, const int32_t* nbr, int64_t nbr_stride,
                            const int32_t* perm, int kflip, const float* bias, void* out_feat,
                            int64_t n_out, pv2_stream_t stream) {
  PV2_REQUIRE(dtype == PV2_BF16 || dtype == PV2_F16, "pv2_spconv16_os_forward: dtype must be PV2_BF16 or PV2_F16");
  PV2_REQUIRE(c_in >= 8 && c_out >= 8 && (c_in % 8) == 0 && (c_out % 8) == 0 && K >= 1,
              "pv2_spconv16_os_forward: channel counts must be multiples of 8");
  PV2_REQUIRE(nbr_stride >= n_out && n_out < 0x7fffffffLL, "pv2_spconv16_os_forward: bad row count");
  (void)n_in;
  if (n_out == 0) return PV2_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == PV2_BF16)
    return launch_os16<BF16>((const unsigned short*)in_feat, c_in, (const uint4*)packed_weight, K,
                             c_out, nbr, nbr_stride, perm, kflip, bias, n_out,
                             (unsigned short*)out_feat, s);
  return launch_os16<F16>((const unsigned short*)in_feat, c_in, (const uint4*)packed_weight, K, c_out,
                          nbr, nbr_stride, perm, kflip, bias, n_out, (unsigned short*)out_feat, s);
}

int pv2_spconv16_backward_weight(const void* in_feat, int64_t n_in, int c_in, const void* grad_out,
                                 int64_t n_out, int c_out, int dtype, int K, const int32_t* pair_in,
                                 const int32_t* pair_out, const int32_t* kstart,
                                 const int32_t* tile_start, int tile_pairs, int64_t n_tiles,
                                 float* grad_weight, pv2_stream_t stream) {
  PV2_REQUIRE(dtype == PV2_BF16 || dtype == PV2_F16, "pv2_spconv16_backward_weight: dtype must be PV2_BF16 or PV2_F16");
  PV2_REQUIRE(c_in >= 8 && c_out >= 8 && (c_in % 8) == 0 && (c_out % 8) == 0 && K >= 1,
              "pv2_spconv16_backward_weight: channel counts must be multiples of 8");
  PV2_REQUIRE(tile_pairs >= 64 && tile_pairs <= kMaxWgradTile16 && (tile_pairs % 64) == 0,
              "pv2_spconv16_backward_weight: tile_pairs must be a multiple of 64, at most 512");
  (void)n_in;
  (void)n_out;
  if (n_tiles == 0) return PV2_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == PV2_BF16)
    return launch_wgrad16<BF16>((const unsigned short*)in_feat, c_in, (const unsigned short*)grad_out,
                                c_out, K, pair_in, pair_out, kstart, tile_start, tile_pairs, n_tiles,
                                grad_weight, s);
  return launch_wgrad16<F16>((const unsigned short*)in_feat, c_in, (const unsigned short*)grad_out,
                             c_out, K, pair_in, pair_out, kstart, tile_start, tile_pairs, n_tiles,
                             grad_weight, s);
}

}  // extern "C"
