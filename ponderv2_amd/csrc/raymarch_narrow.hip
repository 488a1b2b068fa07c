// Fused ray-march kernels of the NeuS render head for NARROW SDF decoders on gfx950 (MI355X): the head
// the reference's nuScenes configuration builds (configs/nuscenes/pretrain-ponder-spunet-v1m1-0-base.py:
// SDFField with sdf_decoder = dict(in_dim=32, out_dim=16+1, hidden_size=16, n_blocks=5), no colour /
// semantic decoder, depth loss only).  Paths relative to the reference checkout:
//   ponder/models/ponder/render_utils/decoders.py:6-36      SDFDecoder: x = fc_p(p) * points_factor, then
//                                                           x = lin_l(x + fc_c[l](feat)), Softplus(100)
//   .../fields/sdf_field.py:185-197, 211-284, 122-146       get_sdf, grad sdf by autograd, NeuS alphas
//   .../rays.py:83-105, renderers.py:33-45                  compositing weights, expected depth
//   .../ray_samplers.py:355-463                             coarse pass + importance sampling
// The stock route is ~500 launches per step for this head (six 16-wide linear layers per pass, their
// double backward for grad sdf, three volume-gradient scatters): 15 ms of a 34 ms step.
//
// Parameters arrive as ONE flat vector theta (built by torch.cat on the host, so autograd splits the
// gradient back onto the nn.Linear parameters):
//     Wp [H,3] bp [H] | l = 0..L-1: Wc_l [H,C] bc_l [H] | l = 0..L-2: W_l [H,H] b_l [H] | w_last [H] b_last
// (C = 32, H = 16, L = 6; of the last layer only the SDF row enters - the geometry features of this
// head feed nothing).  Per sample, with f = trilinear feature, F' = d f / d p (C x 3):
//     x_0 = pf (Wp p + bp)                    X_0 = pf Wp
//     u_l = x_l + Wc_l f + bc_l               U_l = X_l + Wc_l F'
//     z_l = W_l u_l + b_l                     Z_l = W_l U_l
//     x_{l+1} = softplus(z_l)                 X_{l+1} = softplus'(z_l) * Z_l
//     sdf = w_last . u_{L-1} + b_last         grad sdf = U_{L-1}^T w_last
// The value and its three tangents obey the SAME linear recursion, so ONE SAMPLE = ONE QUAD OF LANES:
// lane 0 of the quad carries (p, 1 | f), lane j = 1..3 carries (e_j, 0 | F'[:, j]); every lane runs the
// same 16-wide multiply-adds with the weights as SCALAR operands (uniform addresses: s_load), the
// activation slope crosses the quad by DPP.  fp32 MFMA has no rate advantage over fp32 FMA on this
// machine, so the forward is plain VALU work; the matrix cores do the one thing that needs a
// reduction ACROSS samples, the weight gradients:
//     dWc_l = sum_k ub_l[k] (x) fk[k]         dW_l = sum_k zb_l[k] (x) u_l[k]      k = 4 sample + slot
// with both operands staged once per layer in LDS as [k][16] and fed to v_mfma_f32_16x16x4_f32;
// accumulators stay in registers over a persistent loop and leave as one slab per workgroup (summed
// in slab order afterwards: reproducible).  Backward = reverse mode through value AND tangent
// recursion (second-order terms through grad sdf included), hand-derived; oracle/narrow_head.py states
// the same formulas in torch and tests check them against autograd.
#include "common.h"
#include "raymarch_sampling.h"

namespace {
using namespace pv2rm;

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kC = 32;    // channels of the (channels-last) volume
constexpr int kHd = 16;   // hidden width
constexpr int kL = 6;     // linear layers = n_blocks + 1
// offsets into theta
constexpr int oWp = 0;
constexpr int oBp = oWp + 3 * kHd;
constexpr int oWc = oBp + kHd;
constexpr int sWc = kHd * kC + kHd;          // Wc_l then bc_l
constexpr int oW = oWc + kL * sWc;
constexpr int sW = kHd * kHd + kHd;          // W_l then b_l
constexpr int oWl = oW + (kL - 1) * sW;
constexpr int oBl = oWl + kHd;
constexpr int kTheta = oBl + 1;              // 4609
constexpr int kMaxSlabs = 1024;

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
#define PV2_QUAD_BCAST0(v) dpp_f<0x00>(v)                                         /* quad_perm [0,0,0,0] */
#define PV2_QUAD_SUM(v) ((v) = (v) + dpp_f<0xB1>(v), (v) = (v) + dpp_f<0x4E>(v))  /* [1,0,3,2], [2,3,0,1] */

// this lane's feature vector: f (slot 0) or d f / d p_j (slot j) - the trilinear corner model with the
// lane's own corner coefficients; `offs` / `coef` are kept for the scatter of the backward
__device__ __forceinline__ void gather_fk(const Vol& vol, int scene, float px, float py, float pz, int q,
                                          float (&fk)[kC], int64_t (&offs)[8], float (&coef)[8]) {
  const Axes ax = make_axes(px, py, pz, vol);
#pragma unroll
  for (int j = 0; j < kC; ++j) fk[j] = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float w, dx, dy, dz;
    const bool ok = corner(ax, vol, scene, c, kC, &offs[c], &w, &dx, &dy, &dz);
    coef[c] = q == 0 ? w : (q == 1 ? dx : (q == 2 ? dy : dz));
    if (ok) {
#pragma unroll
      for (int j = 0; j < kC / 4; ++j) {
        const float4 v = ldg4(vol.p + offs[c] + 4 * j);
        fk[4 * j + 0] += coef[c] * v.x;
        fk[4 * j + 1] += coef[c] * v.y;
        fk[4 * j + 2] += coef[c] * v.z;
        fk[4 * j + 3] += coef[c] * v.w;
      }
    } else {
      offs[c] = -1;
    }
  }
}

// u = x + Wc_l fk + bc_l a3   (a3 = 1 on value lanes, 0 on tangent lanes)
__device__ __forceinline__ void layer_in(const float* __restrict__ wc, const float (&x)[kHd],
                                         const float (&fk)[kC], float a3, float (&u)[kHd]) {
#pragma unroll
  for (int m = 0; m < kHd; ++m) {
    float acc = fmaf(wc[kHd * kC + m], a3, x[m]);
#pragma unroll
    for (int c = 0; c < kC; ++c) acc = fmaf(wc[m * kC + c], fk[c], acc);
    u[m] = acc;
  }
}
// z = W_l u + b_l a3
__device__ __forceinline__ void layer_lin(const float* __restrict__ w, const float (&u)[kHd], float a3,
                                          float (&z)[kHd]) {
#pragma unroll
  for (int m = 0; m < kHd; ++m) {
    float acc = w[kHd * kHd + m] * a3;
#pragma unroll
    for (int h = 0; h < kHd; ++h) acc = fmaf(w[m * kHd + h], u[h], acc);
    z[m] = acc;
  }
}

// The recursion on one lane.  QUAD: lanes 1..3 of a quad are tangent lanes and take the activation
// slope from lane 0; otherwise every lane is a value lane (coarse pass).  u_lds != nullptr: this
// lane's u_l row is stored at u_lds + l * 64 * kHd (backward).  Returns sdf (value) / grad_j (tangent).
template <bool QUAD>
__device__ __forceinline__ float narrow_forward(const float* __restrict__ th, float pf,
                                                const float (&fk)[kC], float a0, float a1, float a2,
                                                float a3, bool is_value, float* u_lds) {
  float x[kHd], u[kHd], z[kHd];
#pragma unroll
  for (int m = 0; m < kHd; ++m)
    x[m] = pf * (th[oWp + 3 * m] * a0 + th[oWp + 3 * m + 1] * a1 + th[oWp + 3 * m + 2] * a2 +
                 th[oBp + m] * a3);
#pragma unroll 1
  for (int l = 0; l < kL - 1; ++l) {
    layer_in(th + oWc + l * sWc, x, fk, a3, u);
    if (u_lds) {
#pragma unroll
      for (int m = 0; m < kHd; m += 4)
        *reinterpret_cast<float4*>(u_lds + l * 64 * kHd + m) = make_float4(u[m], u[m + 1], u[m + 2], u[m + 3]);
    }
    layer_lin(th + oW + l * sW, u, a3, z);
#pragma unroll
    for (int m = 0; m < kHd; ++m) {
      float sp, d1, d2;
      softplus100(z[m], &sp, &d1, &d2);
      if (QUAD) {
        const float s = PV2_QUAD_BCAST0(d1);
        x[m] = is_value ? sp : s * z[m];
      } else {
        x[m] = sp;
      }
    }
  }
  layer_in(th + oWc + (kL - 1) * sWc, x, fk, a3, u);
  if (u_lds) {
#pragma unroll
    for (int m = 0; m < kHd; m += 4)
      *reinterpret_cast<float4*>(u_lds + (kL - 1) * 64 * kHd + m) =
          make_float4(u[m], u[m + 1], u[m + 2], u[m + 3]);
  }
  float out = th[oBl] * a3;
#pragma unroll
  for (int m = 0; m < kHd; ++m) out = fmaf(th[oWl + m], u[m], out);
  return out;
}

// ------------------------------------------------------------------------------------------
// Coarse pass: one workgroup per ray, one THREAD per coarse sample (value only), then the shared
// importance sampling / merge.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void narrow_coarse_kernel(
    Vol vol, const float* __restrict__ th, float pf, const float* __restrict__ origins,
    const float* __restrict__ dirs, const float* __restrict__ nears, const float* __restrict__ fars,
    int S0, int n_imp, const float* __restrict__ lin_bins, const float* __restrict__ t_rand,
    int t_rand_cols, const float* __restrict__ lin_u, const float* __restrict__ u_rand, int u_rand_cols,
    float base_inv_s, float* __restrict__ bins_out, float* __restrict__ starts_out,
    float* __restrict__ deltas_out, int32_t* __restrict__ dbg_idx, float* __restrict__ dbg_sdf,
    float* __restrict__ dbg_w) {
  __shared__ SampleLds L;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t ray = blockIdx.x;
  const float nearv = nears[ray], farv = fars[ray];
  const int nthreads = blockDim.x;
  coarse_bins(L, ray, nearv, farv, S0, lin_bins, t_rand, t_rand_cols, tid, nthreads);
  if (tid < S0) {   // SDF at the start positions
    const int scene = (int)(ray / vol.rays_per_scene);
    const float t = L.e[tid];
    const float px = origins[ray * 3 + 0] + dirs[ray * 3 + 0] * t;
    const float py = origins[ray * 3 + 1] + dirs[ray * 3 + 1] * t;
    const float pz = origins[ray * 3 + 2] + dirs[ray * 3 + 2] * t;
    float fk[kC], coef[8];
    int64_t offs[8];
    gather_fk(vol, scene, px, py, pz, 0, fk, offs, coef);
    L.sdf[tid] = narrow_forward<false>(th, pf, fk, px, py, pz, 1.f, true, nullptr);
  }
  __syncthreads();
  importance_merge(L, ray, nearv, farv, S0, n_imp, lin_u, u_rand, u_rand_cols, base_inv_s, bins_out,
                   starts_out, deltas_out, dbg_idx, dbg_sdf, dbg_w, tid, nthreads, wave, lane);
}

// ------------------------------------------------------------------------------------------
// Main pass, forward: sdf and grad sdf of every sample (one quad of lanes per sample).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void narrow_field_fwd_kernel(
    Vol vol, const float* __restrict__ th, float pf, const float* __restrict__ origins,
    const float* __restrict__ dirs, const float* __restrict__ starts, int64_t n_samples, int S,
    float* __restrict__ sdf, float* __restrict__ grad) {
  const int q = threadIdx.x & 3;
  const int64_t n = (int64_t)blockIdx.x * 64 + (threadIdx.x >> 2);
  const bool valid = n < n_samples;
  const int64_t nn = valid ? n : n_samples - 1;
  const int64_t ray = nn / S;
  const int scene = (int)(ray / vol.rays_per_scene);
  const float t = starts[nn];
  const float px = origins[ray * 3 + 0] + dirs[ray * 3 + 0] * t;
  const float py = origins[ray * 3 + 1] + dirs[ray * 3 + 1] * t;
  const float pz = origins[ray * 3 + 2] + dirs[ray * 3 + 2] * t;
  float fk[kC], coef[8];
  int64_t offs[8];
  gather_fk(vol, scene, px, py, pz, q, fk, offs, coef);
  const bool is_value = q == 0;
  const float out = narrow_forward<true>(th, pf, fk, is_value ? px : (q == 1 ? 1.f : 0.f),
                                         is_value ? py : (q == 2 ? 1.f : 0.f),
                                         is_value ? pz : (q == 3 ? 1.f : 0.f), is_value ? 1.f : 0.f,
                                         is_value, nullptr);
  if (valid) {
    if (is_value) sdf[n] = out;
    else grad[n * 3 + (q - 1)] = out;
  }
}

// ------------------------------------------------------------------------------------------
// Per ray: NeuS alphas, transmittance, weights, [sum w t, sum w]; and the way back to
// d L / d sdf_k, d L / d grad_k, d L / d inv_s.  One thread per ray (S <= 191 steps of a few flops).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void alpha_terms(float sdf, float gx, float gy, float gz, float dx, float dy,
                                            float dz, float delta, float inv_s, float* c, float* half,
                                            float* e1, float* e2, float* raw) {
  *c = gx * dx + gy * dy + gz * dz;
  *half = fminf(*c, 0.f) * delta * 0.5f;                 // -relu(-c) * delta / 2
  *e1 = sigmoidf_((sdf - *half) * inv_s);
  *e2 = sigmoidf_((sdf + *half) * inv_s);
  *raw = (*e1 - *e2 + 1e-5f) / (*e1 + 1e-5f);
}

__global__ __launch_bounds__(64) void narrow_composite_fwd_kernel(
    const float* __restrict__ sdf, const float* __restrict__ grad, const float* __restrict__ dirs,
    const float* __restrict__ starts, const float* __restrict__ deltas,
    const float* __restrict__ inv_s_p, int64_t n_rays, int S, float* __restrict__ weights,
    float* __restrict__ trans, float* __restrict__ comp) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  const float inv_s = inv_s_p[0];
  const float dx = dirs[r * 3], dy = dirs[r * 3 + 1], dz = dirs[r * 3 + 2];
  float T = 1.f, st = 0.f, sw = 0.f;
  for (int k = 0; k < S; ++k) {
    const int64_t n = r * S + k;
    float c, half, e1, e2, raw;
    alpha_terms(sdf[n], grad[n * 3], grad[n * 3 + 1], grad[n * 3 + 2], dx, dy, dz, deltas[n], inv_s, &c,
                &half, &e1, &e2, &raw);
    const float alpha = fminf(fmaxf(raw, 0.f), 1.f);
    const float w = alpha * T;
    weights[n] = w;
    trans[n] = T;
    st += w * starts[n];
    sw += w;
    T *= 1.f - alpha + 1e-7f;
  }
  comp[r * 2] = st;
  comp[r * 2 + 1] = sw;
}

__global__ __launch_bounds__(64) void narrow_composite_bwd_kernel(
    const float* __restrict__ sdf, const float* __restrict__ grad, const float* __restrict__ dirs,
    const float* __restrict__ starts, const float* __restrict__ deltas,
    const float* __restrict__ inv_s_p, int64_t n_rays, int S, const float* __restrict__ weights,
    const float* __restrict__ trans, const float* __restrict__ g_comp,
    const float* __restrict__ g_weights, const float* __restrict__ g_sdf,
    const float* __restrict__ g_grad, float* __restrict__ ga, float* __restrict__ g_invs_part) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  const float inv_s = inv_s_p[0];
  const float dx = dirs[r * 3], dy = dirs[r * 3 + 1], dz = dirs[r * 3 + 2];
  const float gt = g_comp[r * 2], gs = g_comp[r * 2 + 1];
  float after = 0.f, ginv = 0.f;
  for (int k = S - 1; k >= 0; --k) {
    const int64_t n = r * S + k;
    const float sd = sdf[n];
    float c, half, e1, e2, raw;
    alpha_terms(sd, grad[n * 3], grad[n * 3 + 1], grad[n * 3 + 2], dx, dy, dz, deltas[n], inv_s, &c,
                &half, &e1, &e2, &raw);
    const float alpha = fminf(fmaxf(raw, 0.f), 1.f);
    const float w = weights[n];
    const float gw = gt * starts[n] + gs + (g_weights ? g_weights[n] : 0.f);
    const float g_alpha = gw * trans[n] - after / (1.f - alpha + 1e-7f);
    after += gw * w;
    const float g_raw = (raw >= 0.f && raw <= 1.f) ? g_alpha : 0.f;
    const float den = e1 + 1e-5f;
    const float gu1 = g_raw * e2 / (den * den) * e1 * (1.f - e1);
    const float gu2 = -g_raw / den * e2 * (1.f - e2);
    const float a = (g_sdf ? g_sdf[n] : 0.f) + inv_s * (gu1 + gu2);
    const float g_half = inv_s * (gu2 - gu1);
    ginv += gu1 * (sd - half) + gu2 * (sd + half);
    const float g_c = c < 0.f ? g_half * deltas[n] * 0.5f : 0.f;
    ga[n * 4 + 0] = a;
    ga[n * 4 + 1] = (g_grad ? g_grad[n * 3 + 0] : 0.f) + g_c * dx;
    ga[n * 4 + 2] = (g_grad ? g_grad[n * 3 + 1] : 0.f) + g_c * dy;
    ga[n * 4 + 3] = (g_grad ? g_grad[n * 3 + 2] : 0.f) + g_c * dz;
  }
  g_invs_part[r] = ginv;
}

// ------------------------------------------------------------------------------------------
// Main pass, backward.  One wave per workgroup, 16 samples (64 slots) per round, persistent.
// ------------------------------------------------------------------------------------------
constexpr int kLdS = 20;   // row stride of the staging tile (spreads the 16-byte row writes over 8 bank groups)

__device__ __forceinline__ int fk_phys(int row, int col) { return row * kC + (col ^ ((row & 1) << 4)); }

// dWc_l += stage^T fk,  dbc_l += column sums of the value rows of stage
__device__ __forceinline__ void accum_wc(const float* s_st, const float* s_fk, int lane, f32x4 (&acc)[2],
                                         float& bsum) {
  const int m16 = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int step = 0; step < 16; ++step) {
    const int row = 4 * step + kk;
    const float a = s_st[row * kLdS + m16];
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, s_fk[fk_phys(row, m16)], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, s_fk[fk_phys(row, 16 + m16)], acc[1], 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) bsum += s_st[(4 * (kk + 4 * i)) * kLdS + m16];   // value rows k = 4 s
}
// dW_l += stage^T u_l,  db_l += column sums of the value rows of stage
__device__ __forceinline__ void accum_w(const float* s_st, const float* s_ul, int lane, f32x4& acc,
                                        float& bsum) {
  const int m16 = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int step = 0; step < 16; ++step) {
    const int row = 4 * step + kk;
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(s_st[row * kLdS + m16], s_ul[row * kHd + m16], acc, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) bsum += s_st[(4 * (kk + 4 * i)) * kLdS + m16];
}

__device__ __forceinline__ void stage_row(float* s_st, int lane, const float (&v)[kHd]) {
#pragma unroll
  for (int m = 0; m < kHd; m += 4)
    *reinterpret_cast<float4*>(s_st + lane * kLdS + m) = make_float4(v[m], v[m + 1], v[m + 2], v[m + 3]);
}

__global__ __launch_bounds__(64) void narrow_field_bwd_kernel(
    Vol vol, const float* __restrict__ th, float pf, const float* __restrict__ origins,
    const float* __restrict__ dirs, const float* __restrict__ starts, int64_t n_samples, int S,
    const float* __restrict__ ga, float* __restrict__ g_vol, float* __restrict__ slabs,
    int64_t n_groups) {
  __shared__ __attribute__((aligned(16))) float s_u[kL * 64 * kHd];   // u_l rows, [l][k][16]
  __shared__ __attribute__((aligned(16))) float s_fk[64 * kC];        // fk rows, swizzled
  __shared__ __attribute__((aligned(16))) float s_st[64 * kLdS];      // bars of the current layer
  __shared__ float s_coef[64], s_p[16 * 3];
  const int lane = threadIdx.x, q = lane & 3, sidx = lane >> 2;
  const int m16 = lane & 15, kk = lane >> 4;
  const bool is_value = q == 0;
  const float a3 = is_value ? 1.f : 0.f;

  f32x4 accWc[kL][2], accW[kL - 1];
  float bcb[kL], bb[kL - 1], wpb[3] = {0.f, 0.f, 0.f}, bpb = 0.f, wlb = 0.f, blb = 0.f;
#pragma unroll
  for (int l = 0; l < kL; ++l) {
    accWc[l][0] = accWc[l][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    bcb[l] = 0.f;
  }
#pragma unroll
  for (int l = 0; l < kL - 1; ++l) {
    accW[l] = f32x4{0.f, 0.f, 0.f, 0.f};
    bb[l] = 0.f;
  }

  for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int64_t n = g * 16 + sidx;
    const bool valid = n < n_samples;
    const int64_t nn = valid ? n : n_samples - 1;
    const int64_t ray = nn / S;
    const int scene = (int)(ray / vol.rays_per_scene);
    const float t = starts[nn];
    const float px = origins[ray * 3 + 0] + dirs[ray * 3 + 0] * t;
    const float py = origins[ray * 3 + 1] + dirs[ray * 3 + 1] * t;
    const float pz = origins[ray * 3 + 2] + dirs[ray * 3 + 2] * t;
    float fk[kC], coef[8];
    int64_t offs[8];
    gather_fk(vol, scene, px, py, pz, q, fk, offs, coef);
    if (!valid) {
#pragma unroll
      for (int j = 0; j < kC; ++j) fk[j] = 0.f;
    }
    // upstream: a = dL/dsdf on the value lane, gamma_j = dL/dgrad_j on tangent lane j
    const float up = valid ? ga[nn * 4 + q] : 0.f;
    __syncthreads();   // the previous round's readers of the LDS tiles are done
#pragma unroll
    for (int j = 0; j < kC; j += 4)
      *reinterpret_cast<float4*>(&s_fk[fk_phys(lane, j)]) = make_float4(fk[j], fk[j + 1], fk[j + 2], fk[j + 3]);
    s_coef[lane] = up;
    if (is_value) {
      s_p[sidx * 3 + 0] = px;
      s_p[sidx * 3 + 1] = py;
      s_p[sidx * 3 + 2] = pz;
    }
    // forward again, u_l rows into LDS
    (void)narrow_forward<true>(th, pf, fk, is_value ? px : (q == 1 ? 1.f : 0.f),
                               is_value ? py : (q == 2 ? 1.f : 0.f), is_value ? pz : (q == 3 ? 1.f : 0.f),
                               a3, is_value, s_u + lane * kHd);
    __syncthreads();

    // last layer: sdf = w_last . u + b_last, grad_j = w_last . U_j
    {
      float v = 0.f;   // dw_last[m16] partial over slots k = 16 kk + i
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = 16 * kk + i;
        v = fmaf(s_coef[k], s_u[(kL - 1) * 64 * kHd + k * kHd + m16], v);
      }
      wlb += v;
      blb += is_value ? up : 0.f;
    }
    float ub[kHd], fkb[kC];
#pragma unroll
    for (int m = 0; m < kHd; ++m) ub[m] = up * th[oWl + m];
#pragma unroll
    for (int j = 0; j < kC; ++j) fkb[j] = 0.f;

#pragma unroll 1
    for (int l = kL - 1; l >= 0; --l) {
      const float* wc = th + oWc + l * sWc;
      __syncthreads();
      stage_row(s_st, lane, ub);
      __syncthreads();
      switch (l) {   // (uniform) - the accumulators are registers, so each layer has its own copy
        case 0: accum_wc(s_st, s_fk, lane, accWc[0], bcb[0]); break;
        case 1: accum_wc(s_st, s_fk, lane, accWc[1], bcb[1]); break;
        case 2: accum_wc(s_st, s_fk, lane, accWc[2], bcb[2]); break;
        case 3: accum_wc(s_st, s_fk, lane, accWc[3], bcb[3]); break;
        case 4: accum_wc(s_st, s_fk, lane, accWc[4], bcb[4]); break;
        default: accum_wc(s_st, s_fk, lane, accWc[5], bcb[5]); break;
      }
      // fkb += Wc_l^T ub
#pragma unroll
      for (int m = 0; m < kHd; ++m) {
#pragma unroll
        for (int c = 0; c < kC; ++c) fkb[c] = fmaf(wc[m * kC + c], ub[m], fkb[c]);
      }
      if (l == 0) break;
      // through x_l = softplus(z_{l-1}), X_l = softplus'(z_{l-1}) Z_{l-1}
      const float* w = th + oW + (l - 1) * sW;
      float u[kHd], zs[kHd], zb[kHd];
#pragma unroll
      for (int m = 0; m < kHd; m += 4) {
        const float4 v = *reinterpret_cast<const float4*>(&s_u[(l - 1) * 64 * kHd + lane * kHd + m]);
        u[m] = v.x;
        u[m + 1] = v.y;
        u[m + 2] = v.z;
        u[m + 3] = v.w;
      }
      layer_lin(w, u, a3, zs);   // z (value lane) / Z_j (tangent lanes)
#pragma unroll
      for (int m = 0; m < kHd; ++m) {
        float sp, d1, d2;
        softplus100(zs[m], &sp, &d1, &d2);
        const float s1 = PV2_QUAD_BCAST0(d1), s2 = PV2_QUAD_BCAST0(d2);
        float tz = is_value ? 0.f : ub[m] * zs[m];
        PV2_QUAD_SUM(tz);
        zb[m] = is_value ? ub[m] * s1 + tz * s2 : s1 * ub[m];
      }
      __syncthreads();
      stage_row(s_st, lane, zb);
      __syncthreads();
      switch (l - 1) {
        case 0: accum_w(s_st, s_u + 0 * 64 * kHd, lane, accW[0], bb[0]); break;
        case 1: accum_w(s_st, s_u + 1 * 64 * kHd, lane, accW[1], bb[1]); break;
        case 2: accum_w(s_st, s_u + 2 * 64 * kHd, lane, accW[2], bb[2]); break;
        case 3: accum_w(s_st, s_u + 3 * 64 * kHd, lane, accW[3], bb[3]); break;
        default: accum_w(s_st, s_u + 4 * 64 * kHd, lane, accW[4], bb[4]); break;
      }
      // ub = W_{l-1}^T zb
#pragma unroll
      for (int h = 0; h < kHd; ++h) ub[h] = 0.f;
#pragma unroll
      for (int m = 0; m < kHd; ++m) {
#pragma unroll
        for (int h = 0; h < kHd; ++h) ub[h] = fmaf(w[m * kHd + h], zb[m], ub[h]);
      }
    }
    // x_0 = pf (Wp p + bp), X_0 = pf Wp: the staging tile holds ub = bars of x_0 / X_0
    {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int s = kk + 4 * i;
        const float xb = s_st[(4 * s) * kLdS + m16];
        bpb += xb;
#pragma unroll
        for (int a = 0; a < 3; ++a) wpb[a] += xb * s_p[s * 3 + a] + s_st[(4 * s + 1 + a) * kLdS + m16];
      }
    }
    // volume: gV[corner] += sum over the quad of coef * fkb
    if (g_vol) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int64_t off = offs[c];   // (the corner rows are the sample's: same on the four lanes)
        float v[kC];
#pragma unroll
        for (int j = 0; j < kC; ++j) {
          v[j] = coef[c] * fkb[j];
          PV2_QUAD_SUM(v[j]);
        }
        if (valid && off >= 0) {   // lane q adds channels 8 q .. 8 q + 7
#pragma unroll
          for (int j = 0; j < kC / 4; ++j) {
            const float x = q == 0 ? v[j] : (q == 1 ? v[8 + j] : (q == 2 ? v[16 + j] : v[24 + j]));
            atomicAdd(g_vol + off + 8 * q + j, x);
          }
        }
      }
    }
  }

  // one slab per workgroup, theta layout
  float* slab = slabs + (int64_t)blockIdx.x * kTheta;
#pragma unroll
  for (int l = 0; l < kL; ++l) {
#pragma unroll
    for (int tcol = 0; tcol < 2; ++tcol)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        slab[oWc + l * sWc + (4 * kk + r) * kC + 16 * tcol + m16] = accWc[l][tcol][r];
    float v = bcb[l];
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (lane < 16) slab[oWc + l * sWc + kHd * kC + lane] = v;
  }
#pragma unroll
  for (int l = 0; l < kL - 1; ++l) {
#pragma unroll
    for (int r = 0; r < 4; ++r) slab[oW + l * sW + (4 * kk + r) * kHd + m16] = accW[l][r];
    float v = bb[l];
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (lane < 16) slab[oW + l * sW + kHd * kHd + lane] = v;
  }
  {
    float v = wlb;
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (lane < 16) slab[oWl + lane] = v;
    float b = blb;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) b += __shfl_xor(b, o);
    if (lane == 0) slab[oBl] = b;
    float bp = bpb;
    bp += __shfl_xor(bp, 16);
    bp += __shfl_xor(bp, 32);
    if (lane < 16) slab[oBp + lane] = pf * bp;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float w = wpb[a];
      w += __shfl_xor(w, 16);
      w += __shfl_xor(w, 32);
      if (lane < 16) slab[oWp + 3 * lane + a] = pf * w;
    }
  }
}

}  // namespace

extern "C" {

int pv2_narrow_head_dims(int* channels, int* hidden, int* layers, int* theta_len) {
  *channels = kC;
  *hidden = kHd;
  *layers = kL;
  *theta_len = kTheta;
  return PV2_OK;
}

#define PV2_NARROW_VOL_CHECK(name)                                                               \
  PV2_REQUIRE(vol_b >= 1 && vol_z >= 2 && vol_y >= 2 && vol_x >= 2, name ": bad volume shape");  \
  PV2_REQUIRE(vol_c == kC, name ": wrong channel count of the (channels-last) volume");          \
  PV2_REQUIRE(n_rays >= 0 && (n_rays % vol_b) == 0, name ": rays must split evenly over scenes")

int pv2_narrow_coarse_sample(const float* volume, int vol_b, int vol_z, int vol_y, int vol_x, int vol_c,
                             const float* origins, const float* dirs, const float* nears,
                             const float* fars, int64_t n_rays, int n_coarse, int n_importance,
                             const float* lin_bins, const float* t_rand, int t_rand_cols,
                             const float* lin_u, const float* u_rand, int u_rand_cols,
                             const float* theta, float points_factor, float base_inv_s, float* bins_out,
                             float* starts_out, float* deltas_out, int32_t* dbg_idx, float* dbg_sdf,
                             float* dbg_w, pv2_stream_t stream) {
  PV2_NARROW_VOL_CHECK("pv2_narrow_coarse_sample");
  PV2_REQUIRE(n_coarse >= 2 && n_coarse <= kMaxS0, "pv2_narrow_coarse_sample: 2 <= n_coarse <= 128");
  PV2_REQUIRE(n_importance >= 1 && n_importance <= kMaxImp,
              "pv2_narrow_coarse_sample: 1 <= n_importance <= 63");
  PV2_REQUIRE(t_rand == nullptr || t_rand_cols == 1 || t_rand_cols == n_coarse + 1,
              "pv2_narrow_coarse_sample: t_rand must have 1 or n_coarse+1 columns");
  PV2_REQUIRE(u_rand == nullptr || u_rand_cols == 1 || u_rand_cols == n_importance + 1,
              "pv2_narrow_coarse_sample: u_rand must have 1 or n_importance+1 columns");
  PV2_REQUIRE((dbg_sdf == nullptr) == (dbg_w == nullptr), "pv2_narrow_coarse_sample: debug outputs");
  if (n_rays == 0) return PV2_OK;
  PV2_REQUIRE(n_rays < 0x7fffffffLL, "pv2_narrow_coarse_sample: too many rays");
  Vol v{volume, vol_b, vol_z, vol_y, vol_x, n_rays / vol_b};
  hipLaunchKernelGGL(narrow_coarse_kernel, dim3((unsigned)n_rays), dim3(128), 0, (hipStream_t)stream, v,
                     theta, points_factor, origins, dirs, nears, fars, n_coarse, n_importance, lin_bins,
                     t_rand, t_rand_cols, lin_u, u_rand, u_rand_cols, base_inv_s, bins_out, starts_out,
                     deltas_out, dbg_idx, dbg_sdf, dbg_w);
  return pv2::check_launch("narrow_coarse_sample");
}

int pv2_narrow_field_forward(const float* volume, int vol_b, int vol_z, int vol_y, int vol_x, int vol_c,
                             const float* origins, const float* dirs, const float* starts,
                             const float* deltas, int64_t n_rays, int n_samples, const float* theta,
                             float points_factor, const float* inv_s, float* sdf, float* grad,
                             float* weights, float* trans, float* comp, pv2_stream_t stream) {
  PV2_NARROW_VOL_CHECK("pv2_narrow_field_forward");
  PV2_REQUIRE(n_samples >= 1, "pv2_narrow_field_forward: n_samples");
  if (n_rays == 0) return PV2_OK;
  const int64_t n = n_rays * n_samples;
  PV2_REQUIRE(n < 0x7fffffffLL * 32, "pv2_narrow_field_forward: too many samples");
  Vol v{volume, vol_b, vol_z, vol_y, vol_x, n_rays / vol_b};
  hipLaunchKernelGGL(narrow_field_fwd_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0,
                     (hipStream_t)stream, v, theta, points_factor, origins, dirs, starts, n, n_samples, sdf,
                     grad);
  hipLaunchKernelGGL(narrow_composite_fwd_kernel, dim3((unsigned)((n_rays + 63) / 64)), dim3(64), 0,
                     (hipStream_t)stream, sdf, grad, dirs, starts, deltas, inv_s, n_rays, n_samples,
                     weights, trans, comp);
  return pv2::check_launch("narrow_field_forward");
}

int64_t pv2_narrow_backward_slabs(int64_t n_rays, int n_samples) {
  const int64_t groups = (n_rays * n_samples + 15) / 16;
  return groups < kMaxSlabs ? (groups < 1 ? 1 : groups) : kMaxSlabs;
}

int pv2_narrow_field_backward(const float* volume, int vol_b, int vol_z, int vol_y, int vol_x, int vol_c,
                              const float* origins, const float* dirs, const float* starts,
                              const float* deltas, int64_t n_rays, int n_samples, const float* theta,
                              float points_factor, const float* inv_s, const float* sdf,
                              const float* grad, const float* weights, const float* trans,
                              const float* g_comp, const float* g_weights, const float* g_sdf,
                              const float* g_grad, float* work, float* g_volume, float* g_theta_slabs,
                              float* g_inv_s_part, pv2_stream_t stream) {
  PV2_NARROW_VOL_CHECK("pv2_narrow_field_backward");
  PV2_REQUIRE(n_samples >= 1, "pv2_narrow_field_backward: n_samples");
  if (n_rays == 0) return PV2_OK;
  const int64_t n = n_rays * n_samples;
  Vol v{volume, vol_b, vol_z, vol_y, vol_x, n_rays / vol_b};
  hipLaunchKernelGGL(narrow_composite_bwd_kernel, dim3((unsigned)((n_rays + 63) / 64)), dim3(64), 0,
                     (hipStream_t)stream, sdf, grad, dirs, starts, deltas, inv_s, n_rays, n_samples,
                     weights, trans, g_comp, g_weights, g_sdf, g_grad, work, g_inv_s_part);
  const int64_t groups = (n + 15) / 16;
  const int64_t slabs = pv2_narrow_backward_slabs(n_rays, n_samples);
  hipLaunchKernelGGL(narrow_field_bwd_kernel, dim3((unsigned)slabs), dim3(64), 0, (hipStream_t)stream, v,
                     theta, points_factor, origins, dirs, starts, n, n_samples, work, g_volume,
                     g_theta_slabs, groups);
  return pv2::check_launch("narrow_field_backward");
}

}  // extern "C"
