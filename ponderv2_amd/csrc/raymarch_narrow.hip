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
// The value and its three tangents obey the SAME linear recursion, so a sample is FOUR SLOTS - slot 0
// carries (p, 1 | f), slot j = 1..3 carries (e_j, 0 | F'[:, j]) - and a wave runs 64 slots (main pass) as the
// columns of 16 x 16 x 4 fp32 MFMA products: activations stay in the accumulator layout from layer to
// layer (the accumulator registers ARE the next product's B operand when the weight columns are read in
// the matching order, see below), the weights are the A operands - 16 bytes per lane per product, read
// from L1 (forward) or from an LDS copy (backward).  The four slots of a sample sit on four adjacent
// lanes, so the activation slope crosses from the value slot to its tangents by DPP (quad_perm).
// A first version ran the same recursion as scalar-weight FMAs (one lane = one slot, weights through
// s_load): 18 ms for the backward - every product waited on a scalar load with one wave per SIMD;
// the MFMA form with vector weight loads: 6.4 ms; exp / log / rcp on the hardware units instead of libm
// (the activation was longer than the products): forward 1.8 -> 0.95 ms; parameters in LDS: 4.8 ms.
// The weight gradients need a reduction ACROSS samples,
//     dWc_l = sum_k ub_l[k] (x) fk[k]         dW_l = sum_k zb_l[k] (x) u_l[k]      k = slot,
// which is one more MFMA product with both operands staged once per layer in LDS as [k][16];
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
constexpr int kMaxSlabs = 2048;
constexpr int kBwdNcb = 2;   // column blocks per round of the backward: 32 slots = 8 samples.  Measured: 4.8 ms against
                             // 8.1 ms with 4 - its LDS tiles (56 KB with the parameters) leave two waves per CU

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
#define PV2_QUAD_BCAST0(v) dpp_f<0x00>(v)                                         /* quad_perm [0,0,0,0] */
#define PV2_QUAD_SUM(v) ((v) = (v) + dpp_f<0xB1>(v), (v) = (v) + dpp_f<0x4E>(v))  /* [1,0,3,2], [2,3,0,1] */
#define PV2_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ------------------------------------------------------------------------------------------
// One wave = 64 SLOTS = four column blocks of 16.  Lane (n = lane & 15, g = lane >> 4) owns column n
// of every block.  Matrices [rows x slots] live in the MFMA accumulator layout: f32x4 D[cb], component
// r = row 4 g + r of slot 16 cb + n.  The same registers are the B operand of the NEXT product when
// component r is fed at reduction step r: the MFMA then reads it as B[k = 4 r + g], so the A operand of
// step r must carry the weight column of row-unit 4 kq + r - for lane (m, kq) that is W[m][4 kq + r],
// four CONSECUTIVE floats of row m.  No value ever moves between lanes on the way through the layers.
// Features use the same trick on 32 channels: lane (n, g) holds channels 8 g .. 8 g + 7 of its slots
// (two 16-byte loads per corner), step s of 8 pairs them with A = Wc[m][8 kq + s].
// In the main pass a slot is (sample, q): q = n & 3 = 0 the value, 1..3 the tangents d/dp_q, so the
// four slots of a sample are four adjacent lanes; in the coarse pass every slot is a value.
// ------------------------------------------------------------------------------------------
template <int NCB>
struct Feat {
  float v[NCB][8];   // [cb][s] = channel 8 g + s of slot 16 cb + n
};

// trilinear feature (q = 0) or its derivative along p_q of one slot, this lane's 8 channels
__device__ __forceinline__ void gather8(const Vol& vol, int scene, float px, float py, float pz, int q,
                                        int g, float (&f)[8]) {
  const Axes ax = make_axes(px, py, pz, vol);
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    int64_t off;
    float w, dx, dy, dz;
    if (corner(ax, vol, scene, c, kC, &off, &w, &dx, &dy, &dz)) {
      const float cf = q == 0 ? w : (q == 1 ? dx : (q == 2 ? dy : dz));
      const float4 a = ldg4(vol.p + off + 8 * g), b = ldg4(vol.p + off + 8 * g + 4);
      f[0] += cf * a.x;
      f[1] += cf * a.y;
      f[2] += cf * a.z;
      f[3] += cf * a.w;
      f[4] += cf * b.x;
      f[5] += cf * b.y;
      f[6] += cf * b.z;
      f[7] += cf * b.w;
    }
  }
}

// D[cb] += Wc F   (Wc [H, C] row-major)
template <int NCB>
__device__ __forceinline__ void mm_in(const float* __restrict__ wc, int n, int g, const Feat<NCB>& F,
                                      f32x4 (&D)[NCB]) {
  const float4 a0 = ldg4(wc + n * kC + 8 * g), a1 = ldg4(wc + n * kC + 8 * g + 4);
  const float A[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) D[cb] = PV2_MFMA(A[s], F.v[cb][s], D[cb]);
}
// D[cb] += W U   (W [H, H] row-major; U in accumulator layout)
template <int NCB>
__device__ __forceinline__ void mm_lin(const float* __restrict__ w, int n, int g, const f32x4 (&U)[NCB],
                                       f32x4 (&D)[NCB]) {
  const float4 a = ldg4(w + n * kHd + 4 * g);
  const float A[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) D[cb] = PV2_MFMA(A[r], U[cb][r], D[cb]);
}
// D[cb] += W^T Z
template <int NCB>
__device__ __forceinline__ void mm_lin_t(const float* __restrict__ w, int n, int g, const f32x4 (&Z)[NCB],
                                         f32x4 (&D)[NCB]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float a = w[(4 * g + r) * kHd + n];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) D[cb] = PV2_MFMA(a, Z[cb][r], D[cb]);
  }
}
// Fb[rb][cb] += (Wc^T U)[16 rb .. 16 rb + 15]   (rows = channels)
template <int NCB>
__device__ __forceinline__ void mm_in_t(const float* __restrict__ wc, int n, int g, const f32x4 (&U)[NCB],
                                        f32x4 (&Fb)[2][NCB]) {
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float a = wc[(4 * g + r) * kC + 16 * rb + n];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) Fb[rb][cb] = PV2_MFMA(a, U[cb][r], Fb[rb][cb]);
    }
}

__device__ __forceinline__ f32x4 bias4(const float* __restrict__ b, int g, float a3) {
  const float4 v = ldg4(b + 4 * g);
  return f32x4{v.x * a3, v.y * a3, v.z * a3, v.w * a3};
}

// x_0 = pf (Wp a + bp a3) as one K = 4 product: A = [Wp | bp], B = (a_0, a_1, a_2, a3) of the slot
template <int NCB>
__device__ __forceinline__ void first_layer(const float* __restrict__ th, float pf, int n, int g,
                                            const float (&avec)[NCB], f32x4 (&X)[NCB]) {
  const float a = g < 3 ? th[oWp + 3 * n + g] : th[oBp + n];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    X[cb] = PV2_MFMA(a, avec[cb], (f32x4{0.f, 0.f, 0.f, 0.f}));
    X[cb] *= pf;
  }
}

// softplus(beta = 100) on the value slots, slope * Z on the tangent slots (QUAD); value everywhere otherwise
template <bool QUAD, int NCB>
__device__ __forceinline__ void activate(const f32x4 (&Z)[NCB], bool is_value, f32x4 (&X)[NCB]) {
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float sp, d1, d2;
      softplus_fast(Z[cb][r], &sp, &d1, &d2);
      if (QUAD) {
        const float s = PV2_QUAD_BCAST0(d1);
        X[cb][r] = is_value ? sp : s * Z[cb][r];
      } else {
        X[cb][r] = sp;
      }
    }
}

template <int NCB>
__device__ __forceinline__ void store_rows16(float* base, int ld, int n, int g, const f32x4 (&D)[NCB]) {
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
    *reinterpret_cast<float4*>(base + (16 * cb + n) * ld + 4 * g) =
        make_float4(D[cb][0], D[cb][1], D[cb][2], D[cb][3]);
}

// The recursion of one wave.  Returns, per column block, w_last . u_{L-1} (+ b_last a3) summed over the
// lane's four rows ONLY - the caller adds the four row groups.  s_u != nullptr: every u_l is stored as
// [l][slot][16] (backward).
template <bool QUAD, int NCB>
__device__ __forceinline__ void narrow_forward(const float* __restrict__ th, float pf, int n, int g,
                                               const Feat<NCB>& F, const float (&avec)[NCB], float a3,
                                               bool is_value, float* s_u, float (&out)[NCB]) {
  f32x4 X[NCB], U[NCB], Z[NCB];
  first_layer(th, pf, n, g, avec, X);
#pragma unroll 1
  for (int l = 0; l < kL - 1; ++l) {
    const float* wc = th + oWc + l * sWc;
    const f32x4 bc = bias4(wc + kHd * kC, g, a3);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) U[cb] = X[cb] + bc;
    mm_in(wc, n, g, F, U);
    if (s_u) store_rows16(s_u + l * 16 * NCB * kHd, kHd, n, g, U);
    const float* w = th + oW + l * sW;
    const f32x4 bl = bias4(w + kHd * kHd, g, a3);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) Z[cb] = bl;
    mm_lin(w, n, g, U, Z);
    activate<QUAD, NCB>(Z, is_value, X);
  }
  const float* wc = th + oWc + (kL - 1) * sWc;
  const f32x4 bc = bias4(wc + kHd * kC, g, a3);
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) U[cb] = X[cb] + bc;
  mm_in(wc, n, g, F, U);
  if (s_u) store_rows16(s_u + (kL - 1) * 16 * NCB * kHd, kHd, n, g, U);
  const float4 wl = ldg4(th + oWl + 4 * g);
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
    out[cb] = wl.x * U[cb][0] + wl.y * U[cb][1] + wl.z * U[cb][2] + wl.w * U[cb][3] +
              (g == 0 ? th[oBl] * a3 : 0.f);
}

__device__ __forceinline__ float rows_sum(float v) {   // over the four row groups of a column
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// ------------------------------------------------------------------------------------------
// Coarse pass: one workgroup per ray, one slot per coarse sample (value only), then the shared
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
  const int n = lane & 15, g = lane >> 4;
  const int64_t ray = blockIdx.x;
  const float nearv = nears[ray], farv = fars[ray];
  const int nthreads = blockDim.x;
  coarse_bins(L, ray, nearv, farv, S0, lin_bins, t_rand, t_rand_cols, tid, nthreads);
  if (wave * 64 < S0) {   // SDF at the start positions (wave-uniform)
    const int scene = (int)(ray / vol.rays_per_scene);
    const float o0 = origins[ray * 3 + 0], o1 = origins[ray * 3 + 1], o2 = origins[ray * 3 + 2];
    const float d0 = dirs[ray * 3 + 0], d1 = dirs[ray * 3 + 1], d2 = dirs[ray * 3 + 2];
    Feat<4> F;
    float avec[4], out[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const int k = min(wave * 64 + 16 * cb + n, S0 - 1);
      const float t = L.e[k];
      const float px = o0 + d0 * t, py = o1 + d1 * t, pz = o2 + d2 * t;
      gather8(vol, scene, px, py, pz, 0, g, F.v[cb]);
      avec[cb] = g == 0 ? px : (g == 1 ? py : (g == 2 ? pz : 1.f));
    }
    narrow_forward<false, 4>(th, pf, n, g, F, avec, 1.f, true, nullptr, out);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const float v = rows_sum(out[cb]);
      const int k = wave * 64 + 16 * cb + n;
      if (g == 0 && k < S0) L.sdf[k] = v;
    }
  }
  __syncthreads();
  importance_merge(L, ray, nearv, farv, S0, n_imp, lin_u, u_rand, u_rand_cols, base_inv_s, bins_out,
                   starts_out, deltas_out, dbg_idx, dbg_sdf, dbg_w, tid, nthreads, wave, lane);
}

// ------------------------------------------------------------------------------------------
// Main pass, forward: sdf and grad sdf of every sample (16 samples = 64 slots per wave).
// ------------------------------------------------------------------------------------------
template <int NCB>
struct Pos {
  float x[NCB], y[NCB], z[NCB];
  int scene[NCB];
  bool valid[NCB];
};

// positions of this lane's slots (sample 4 cb + (n >> 2) of group `grp` of 4 NCB samples), their
// features and the first layer's input vector
template <int NCB>
__device__ __forceinline__ void load_slots(const Vol& vol, const float* __restrict__ origins,
                                           const float* __restrict__ dirs,
                                           const float* __restrict__ starts, int64_t n_samples, int S,
                                           int64_t grp, int n, int g, Pos<NCB>& P, Feat<NCB>& F,
                                           float (&avec)[NCB]) {
  const int q = n & 3;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int64_t smp = grp * (4 * NCB) + 4 * cb + (n >> 2);
    P.valid[cb] = smp < n_samples;
    const int64_t nn = P.valid[cb] ? smp : n_samples - 1;
    const int64_t ray = nn / S;
    P.scene[cb] = (int)(ray / vol.rays_per_scene);
    const float t = starts[nn];
    P.x[cb] = origins[ray * 3 + 0] + dirs[ray * 3 + 0] * t;
    P.y[cb] = origins[ray * 3 + 1] + dirs[ray * 3 + 1] * t;
    P.z[cb] = origins[ray * 3 + 2] + dirs[ray * 3 + 2] * t;
    gather8(vol, P.scene[cb], P.x[cb], P.y[cb], P.z[cb], q, g, F.v[cb]);
    if (q == 0) avec[cb] = g == 0 ? P.x[cb] : (g == 1 ? P.y[cb] : (g == 2 ? P.z[cb] : 1.f));
    else avec[cb] = g == q - 1 ? 1.f : 0.f;
  }
}

template <int NCB>
__global__ __launch_bounds__(256) void narrow_field_fwd_kernel(
    Vol vol, const float* __restrict__ th, float pf, const float* __restrict__ origins,
    const float* __restrict__ dirs, const float* __restrict__ starts, int64_t n_samples, int S,
    float* __restrict__ sdf, float* __restrict__ grad) {
  const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4, q = n & 3;
  const int64_t grp = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (grp * (4 * NCB) >= n_samples) return;   // (wave-uniform)
  Pos<NCB> P;
  Feat<NCB> F;
  float avec[NCB], out[NCB];
  load_slots<NCB>(vol, origins, dirs, starts, n_samples, S, grp, n, g, P, F, avec);
  narrow_forward<true, NCB>(th, pf, n, g, F, avec, q == 0 ? 1.f : 0.f, q == 0, nullptr, out);
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const float v = rows_sum(out[cb]);
    const int64_t smp = grp * (4 * NCB) + 4 * cb + (n >> 2);
    if (g == 0 && P.valid[cb]) {
      if (q == 0) sdf[smp] = v;
      else grad[smp * 3 + (q - 1)] = v;
    }
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
template <int NCB>
__device__ __forceinline__ void accum_wc(const float* s_st, const float* s_fk, int lane, f32x4 (&acc)[2],
                                         float& bsum) {
  const int m16 = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int step = 0; step < 4 * NCB; ++step) {
    const int row = 4 * step + kk;
    const float a = s_st[row * kLdS + m16];
    acc[0] = PV2_MFMA(a, s_fk[fk_phys(row, m16)], acc[0]);
    acc[1] = PV2_MFMA(a, s_fk[fk_phys(row, 16 + m16)], acc[1]);
  }
#pragma unroll
  for (int i = 0; i < NCB; ++i) bsum += s_st[(4 * (kk + 4 * i)) * kLdS + m16];   // value rows k = 4 s
}
// dW_l += stage^T u_l,  db_l += column sums of the value rows of stage
template <int NCB>
__device__ __forceinline__ void accum_w(const float* s_st, const float* s_ul, int lane, f32x4& acc,
                                        float& bsum) {
  const int m16 = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int step = 0; step < 4 * NCB; ++step) {
    const int row = 4 * step + kk;
    acc = PV2_MFMA(s_st[row * kLdS + m16], s_ul[row * kHd + m16], acc);
  }
#pragma unroll
  for (int i = 0; i < NCB; ++i) bsum += s_st[(4 * (kk + 4 * i)) * kLdS + m16];
}

template <int NCB>
__global__ __launch_bounds__(64) void narrow_field_bwd_kernel(
    Vol vol, const float* __restrict__ theta_g, float pf, const float* __restrict__ origins,
    const float* __restrict__ dirs, const float* __restrict__ starts, int64_t n_samples, int S,
    const float* __restrict__ ga, float* __restrict__ g_vol, float* __restrict__ slabs,
    int64_t n_groups) {
  constexpr int NS = 16 * NCB;   // slots per round
  __shared__ __attribute__((aligned(16))) float s_u[kL * NS * kHd];   // u_l rows, [l][slot][16]
  __shared__ __attribute__((aligned(16))) float s_fk[NS * kC];        // fk rows, swizzled
  __shared__ __attribute__((aligned(16))) float s_st[NS * kLdS];      // bars of the current layer
  __shared__ float s_coef[NS], s_p[4 * NCB * 3];
  // the parameters, once per workgroup: with one wave per SIMD nothing hides a trip to L2 in front of
  // every product, and the 18 KB do not survive in L1 next to the volume gathers
  __shared__ __attribute__((aligned(16))) float s_th[(kTheta + 3) & ~3];
  for (int i = threadIdx.x; i < kTheta; i += 64) s_th[i] = theta_g[i];
  __syncthreads();
  const float* th = s_th;
  const int lane = threadIdx.x, n = lane & 15, g = lane >> 4, q = n & 3;
  const int m16 = n, kk = g;
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

  for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    Pos<NCB> P;
    Feat<NCB> F;
    float avec[NCB], out[NCB], up[NCB];
    load_slots<NCB>(vol, origins, dirs, starts, n_samples, S, grp, n, g, P, F, avec);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int64_t smp = grp * (4 * NCB) + 4 * cb + (n >> 2);
      // upstream: a = dL/dsdf on the value slot, gamma_j = dL/dgrad_j on tangent slot j
      up[cb] = P.valid[cb] ? ga[smp * 4 + q] : 0.f;
      if (!P.valid[cb]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) F.v[cb][j] = 0.f;
      }
    }
    __syncthreads();   // the previous round's readers of the LDS tiles are done
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int slot = 16 * cb + n;
      *reinterpret_cast<float4*>(&s_fk[fk_phys(slot, 8 * g)]) =
          make_float4(F.v[cb][0], F.v[cb][1], F.v[cb][2], F.v[cb][3]);
      *reinterpret_cast<float4*>(&s_fk[fk_phys(slot, 8 * g + 4)]) =
          make_float4(F.v[cb][4], F.v[cb][5], F.v[cb][6], F.v[cb][7]);
      if (g == 0) s_coef[slot] = up[cb];
      if (g == 0 && is_value) {
        s_p[(slot >> 2) * 3 + 0] = P.x[cb];
        s_p[(slot >> 2) * 3 + 1] = P.y[cb];
        s_p[(slot >> 2) * 3 + 2] = P.z[cb];
      }
    }
    // forward again, u_l rows into LDS
    narrow_forward<true, NCB>(th, pf, n, g, F, avec, a3, is_value, s_u, out);
    __syncthreads();

    // last layer: sdf = w_last . u + b_last, grad_j = w_last . U_j
    {
      float v = 0.f;   // dw_last[m16] partial over slots k = 4 NCB kk + i
#pragma unroll
      for (int i = 0; i < 4 * NCB; ++i) {
        const int k = 4 * NCB * kk + i;
        v = fmaf(s_coef[k], s_u[(kL - 1) * NS * kHd + k * kHd + m16], v);
      }
      wlb += v;
      if (g == 0 && is_value) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) blb += up[cb];
      }
    }
    f32x4 UB[NCB], FB[2][NCB];
    {
      const float4 wl = ldg4(th + oWl + 4 * g);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        UB[cb] = f32x4{wl.x, wl.y, wl.z, wl.w} * up[cb];
        FB[0][cb] = FB[1][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }

#pragma unroll 1
    for (int l = kL - 1; l >= 0; --l) {
      const float* wc = th + oWc + l * sWc;
      __syncthreads();
      store_rows16(s_st, kLdS, n, g, UB);
      __syncthreads();
      switch (l) {   // (uniform) - the accumulators are registers, so each layer has its own copy
        case 0: accum_wc<NCB>(s_st, s_fk, lane, accWc[0], bcb[0]); break;
        case 1: accum_wc<NCB>(s_st, s_fk, lane, accWc[1], bcb[1]); break;
        case 2: accum_wc<NCB>(s_st, s_fk, lane, accWc[2], bcb[2]); break;
        case 3: accum_wc<NCB>(s_st, s_fk, lane, accWc[3], bcb[3]); break;
        case 4: accum_wc<NCB>(s_st, s_fk, lane, accWc[4], bcb[4]); break;
        default: accum_wc<NCB>(s_st, s_fk, lane, accWc[5], bcb[5]); break;
      }
      mm_in_t(wc, n, g, UB, FB);   // fkb += Wc_l^T ub
      if (l == 0) break;
      // through x_l = softplus(z_{l-1}), X_l = softplus'(z_{l-1}) Z_{l-1}
      const float* w = th + oW + (l - 1) * sW;
      f32x4 U[NCB], Z[NCB], ZB[NCB];
      const f32x4 bl = bias4(w + kHd * kHd, g, a3);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const float4 v = *reinterpret_cast<const float4*>(&s_u[(l - 1) * NS * kHd + (16 * cb + n) * kHd + 4 * g]);
        U[cb] = f32x4{v.x, v.y, v.z, v.w};
        Z[cb] = bl;
      }
      mm_lin(w, n, g, U, Z);   // z (value slots) / Z_j (tangent slots)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sp, d1, d2;
          softplus_fast(Z[cb][r], &sp, &d1, &d2);
          const float s1 = PV2_QUAD_BCAST0(d1), s2 = PV2_QUAD_BCAST0(d2);
          float tz = is_value ? 0.f : UB[cb][r] * Z[cb][r];
          PV2_QUAD_SUM(tz);
          ZB[cb][r] = is_value ? UB[cb][r] * s1 + tz * s2 : s1 * UB[cb][r];
        }
      __syncthreads();
      store_rows16(s_st, kLdS, n, g, ZB);
      __syncthreads();
      switch (l - 1) {
        case 0: accum_w<NCB>(s_st, s_u + 0 * NS * kHd, lane, accW[0], bb[0]); break;
        case 1: accum_w<NCB>(s_st, s_u + 1 * NS * kHd, lane, accW[1], bb[1]); break;
        case 2: accum_w<NCB>(s_st, s_u + 2 * NS * kHd, lane, accW[2], bb[2]); break;
        case 3: accum_w<NCB>(s_st, s_u + 3 * NS * kHd, lane, accW[3], bb[3]); break;
        default: accum_w<NCB>(s_st, s_u + 4 * NS * kHd, lane, accW[4], bb[4]); break;
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) UB[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
      mm_lin_t(w, n, g, ZB, UB);   // ub = W_{l-1}^T zb
    }
    // x_0 = pf (Wp p + bp), X_0 = pf Wp: the staging tile holds ub = bars of x_0 / X_0
    {
#pragma unroll
      for (int i = 0; i < NCB; ++i) {
        const int s = kk + 4 * i;
        const float xb = s_st[(4 * s) * kLdS + m16];
        bpb += xb;
#pragma unroll
        for (int a = 0; a < 3; ++a) wpb[a] += xb * s_p[s * 3 + a] + s_st[(4 * s + 1 + a) * kLdS + m16];
      }
    }
    // volume: gV[corner] += sum over the sample's four slots of coef * fkb; this lane holds channels
    // 16 rb + 4 g + r of its slots and adds two of the eight after the quad sum
    if (g_vol) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const Axes ax = make_axes(P.x[cb], P.y[cb], P.z[cb], vol);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          int64_t off;
          float w, dx, dy, dz;
          const bool ok = corner(ax, vol, P.scene[cb], c, kC, &off, &w, &dx, &dy, &dz);
          const float cf = q == 0 ? w : (q == 1 ? dx : (q == 2 ? dy : dz));
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            v[j] = cf * FB[j >> 2][cb][j & 3];
            PV2_QUAD_SUM(v[j]);
          }
          if (ok && P.valid[cb]) {   // (same for the four lanes of the sample) lane q: v[2 q], v[2 q + 1]
            const float x0 = q == 0 ? v[0] : (q == 1 ? v[2] : (q == 2 ? v[4] : v[6]));
            const float x1 = q == 0 ? v[1] : (q == 1 ? v[3] : (q == 2 ? v[5] : v[7]));
            float* dst = g_vol + off + 16 * (q >> 1) + 4 * g + 2 * (q & 1);
            atomicAdd(dst, x0);
            atomicAdd(dst + 1, x1);
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
    const float v = rows_sum(bcb[l]);
    if (lane < 16) slab[oWc + l * sWc + kHd * kC + lane] = v;
  }
#pragma unroll
  for (int l = 0; l < kL - 1; ++l) {
#pragma unroll
    for (int r = 0; r < 4; ++r) slab[oW + l * sW + (4 * kk + r) * kHd + m16] = accW[l][r];
    const float v = rows_sum(bb[l]);
    if (lane < 16) slab[oW + l * sW + kHd * kHd + lane] = v;
  }
  {
    const float v = rows_sum(wlb);
    if (lane < 16) slab[oWl + lane] = v;
    float b = blb;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) b += __shfl_xor(b, o);
    if (lane == 0) slab[oBl] = b;
    const float bp = rows_sum(bpb);
    if (lane < 16) slab[oBp + lane] = pf * bp;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float w = rows_sum(wpb[a]);
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
  hipLaunchKernelGGL(narrow_field_fwd_kernel<4>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0,
                     (hipStream_t)stream, v, theta, points_factor, origins, dirs, starts, n, n_samples, sdf,
                     grad);
  hipLaunchKernelGGL(narrow_composite_fwd_kernel, dim3((unsigned)((n_rays + 63) / 64)), dim3(64), 0,
                     (hipStream_t)stream, sdf, grad, dirs, starts, deltas, inv_s, n_rays, n_samples,
                     weights, trans, comp);
  return pv2::check_launch("narrow_field_forward");
}

int64_t pv2_narrow_backward_slabs(int64_t n_rays, int n_samples) {
  const int64_t groups = (n_rays * n_samples + 4 * kBwdNcb - 1) / (4 * kBwdNcb);
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
  const int64_t slabs = pv2_narrow_backward_slabs(n_rays, n_samples);
  hipLaunchKernelGGL(narrow_field_bwd_kernel<kBwdNcb>, dim3((unsigned)slabs), dim3(64), 0,
                     (hipStream_t)stream, v, theta, points_factor, origins, dirs, starts, n, n_samples, work,
                     g_volume, g_theta_slabs, (n + 4 * kBwdNcb - 1) / (4 * kBwdNcb));
  return pv2::check_launch("narrow_field_backward");
}

}  // extern "C"
