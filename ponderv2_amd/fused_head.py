"""The NeuS render head on the fused ray-march kernels of csrc/raymarch_fused.hip.

Per render pass the per-sample work of the reference's head (paths relative to the reference
checkout: render_utils/models/neus.py:16-36, ray_samplers.py:55-107,227-322,355-463,
rays.py:83-153, fields/sdf_field.py:122-284, decoders.py:6-109, renderers.py:5-75) becomes

    forward : coarse_sample (1 launch: stratified bins, coarse SDF, fixed-inv_s weights, inverse-CDF
              samples, merge)  ->  field forward (1 launch: feature gather, SDF MLP on MFMA, grad sdf,
              colour head, alpha)  ->  weights (1)  ->  composite of the 140-wide value row (1)
    backward: composite bwd (1) -> weights bwd (1) -> field backward (1) -> volume scatter (1)
              -> 4 weight-gradient GEMMs

``render_outputs`` is what ``SurfaceModel.get_outputs`` calls when ``usable`` says the model has the
head shape the kernels are compiled for; anything else keeps the modular path.  Layers without an
activation between them are collapsed HERE with torch ops so autograd carries the kernels'
gradients back to the nn.Linear parameters.  Device fp32 only; there is no CPU path in this module
(tests install host doubles from oracle/ over ``coarse_sample`` / ``field_render``).
"""
import ctypes
import os

import torch
import torch.nn.functional as F

from . import _lib, ray_epilogue
from .kernels import _ptr, _require_device, _stream, zeros_by_kernel

ENABLED = os.environ.get("PV2_FUSED_HEAD", "1") != "0"
# The projection network's final 1x1x1 convolution applied per sample inside the head instead of per
# grid cell before it (``FoldedVolume`` / ``field_render_folded`` below); PV2_FOLD_FINAL_CONV=0
# materialises the 128-channel volume as the reference does.
FOLD_ENABLED = os.environ.get("PV2_FOLD_FINAL_CONV", "1") != "0"
KX, KXP = 32, 40   # channels of the pre-convolution volume / row width [xt, s, 0 x 7] (pv2_neus_fold_dims)
# tests set this to a dict to receive the coarse pass's diagnostics (importance-sampling bin
# indices, coarse SDF and weights) of the next render
CAPTURE = None

_DIMS = None


def dims():
    """(hidden, f_sdf, f_rest, geo, value_row, sums_len) compiled into the kernels."""
    global _DIMS
    if _DIMS is None:
        vals = [ctypes.c_int() for _ in range(6)]
        _lib.check(_lib.lib().pv2_neus_head_dims(*[ctypes.byref(v) for v in vals]), "pv2_neus_head_dims")
        _DIMS = tuple(v.value for v in vals)
    return _DIMS


H, FS, F2, G, NV, NSUM = 128, 64, 64, 64, 140, 524  # checked against dims() on first device use
# value row columns
COL_F2, COL_GEO, COL_G, COL_N, COL_RGB, COL_T, COL_ONE = 0, 64, 128, 131, 134, 137, 138
# column-sum layout of the backward (csrc/raymarch_fused.hip kSum*)
SUM_C0, SUM_BC1, SUM_V1, SUM_B1, SUM_Q, SUM_RGB, SUM_INVS = 0, 128, 256, 384, 452, 516, 520


def device_ok(t):
    """Tensors the kernels take (overridden by the host doubles in tests).  Reduced-precision
    volumes (the projection network under autocast) are widened to fp32 on the way in."""
    return t.is_cuda and t.dtype in (torch.float32, torch.bfloat16, torch.float16)


class FoldedVolume:
    """What the projection network hands the render head when its last layer - a 1x1x1
    convolution (unet3d.py ``final_conv``) - is to be applied per SAMPLE: the activations in front
    of it (B, 32, Z, Y, X) and the convolution module.  ``materialize()`` is the 128-channel volume
    the reference builds, for every consumer other than the fused head."""

    def __init__(self, pre, conv):
        self.pre, self.conv = pre, conv

    @property
    def shape(self):
        b, _, z, y, x = self.pre.shape
        return torch.Size((b, self.conv.out_channels, z, y, x))

    def dim(self):
        return 5

    @property
    def dtype(self):
        return self.pre.dtype

    @property
    def is_cuda(self):
        return self.pre.is_cuda

    def materialize(self):
        from .ponder.models.ponder.unet3d import library_conv

        return library_conv(self.conv, self.pre)

    def x5(self):
        """The (B,Z,Y,X,32) channels-last fp32 view of the activations."""
        widen = lambda t: t.float() if t.dtype in (torch.bfloat16, torch.float16) else t
        x5 = widen(self.pre.permute(0, 2, 3, 4, 1))   # (float64 stays: the host doubles of the tests)
        return x5 if x5.is_contiguous() else x5.contiguous()

    def rows(self):
        """(x5, wfp): the (B,Z,Y,X,32) channels-last fp32 view of the activations and the
        [C_out, 40] matrix [Wf | bf | 0] (built by torch ops: gradients reach the module)."""
        widen = lambda t: t.float() if t.dtype in (torch.bfloat16, torch.float16) else t
        x5 = self.x5()
        conv = self.conv
        wf = widen(conv.weight.reshape(conv.out_channels, conv.in_channels)).to(x5.dtype)
        bf = (widen(conv.bias).to(x5.dtype) if conv.bias is not None
              else torch.zeros(conv.out_channels, dtype=x5.dtype, device=wf.device))
        pad = torch.zeros((conv.out_channels, KXP - KX - 1), dtype=x5.dtype, device=wf.device)
        return x5, torch.cat([wf, bf[:, None], pad], dim=1)


def fold_shape_ok(conv, x):
    """``conv`` is a 1x1x1 / stride 1 nn.Conv3d from KX to FS + F2 channels on a 5-D input: the
    shape the folded kernels are compiled for."""
    return (FOLD_ENABLED and ENABLED and isinstance(conv, torch.nn.Conv3d) and x.dim() == 5
            and tuple(conv.kernel_size) == (1, 1, 1) and tuple(conv.stride) == (1, 1, 1)
            and tuple(conv.padding) == (0, 0, 0) and conv.groups == 1
            and conv.in_channels == KX and conv.out_channels == FS + F2)


def fold_supported(conv, x):
    """``fold_shape_ok`` on a device volume in a type the kernels take (the tests' host doubles
    replace this by the shape check alone)."""
    return (fold_shape_ok(conv, x) and x.is_cuda and conv.weight.dtype == torch.float32
            and x.dtype in (torch.float32, torch.bfloat16, torch.float16))


def unfold(volume_feature):
    """The volume list with every ``FoldedVolume`` materialised (for the modular head)."""
    return [v.materialize() if isinstance(v, FoldedVolume) else v for v in volume_feature]


def _vol5(volume_feature, num_scenes):
    """The (B,Z,Y,X,C) channels-last view of the projected volume, or None."""
    if len(volume_feature) != 1:
        return None
    v = volume_feature[0]
    if v.dim() == 4:
        v = v.unsqueeze(0)
    if v.dim() != 5 or v.shape[0] != num_scenes:
        return None
    v = v.permute(0, 2, 3, 4, 1)
    return v if v.is_contiguous() else v.contiguous()


def usable(model, ray_bundle, volume_feature):
    """True when ``model`` (a NeuSModel) has exactly the head the fused kernels implement."""
    from .ponder.models.ponder.render_utils.ray_samplers import NeuSSampler, UniformSampler

    if not ENABLED:
        return False
    f = model.field
    smp = model.sampler
    sd, rd, md = f.sdf_decoder, f.rgb_decoder, f.semantic_decoder
    if not (isinstance(smp, NeuSSampler) and isinstance(smp.initial_sampler, UniformSampler)
            and smp.num_upsample_steps == 1 and 2 <= smp.num_samples <= 128
            and 1 <= smp.num_samples_importance <= 63):
        return False
    if not (f.volume_type == "default" and f.padding_mode == "zeros" and not f.share_volume
            and f.use_gradient and f._cos_anneal_ratio == 1.0 and rd is not None):
        return False
    if model.loss.weights.get("sparse_points_sdf_loss", 0.0) > 0:
        return False
    if not (sd.num_layers == 3 and rd.num_layers == 2 and sd.points_factor == 0.0
            and rd.points_factor == 0.0 and sd.fc_c[0].in_features == FS
            and sd.lin0.out_features == H and sd.lin1.out_features == 1 + G
            and rd.fc_c[0].in_features == 3 + F2 + G + 3 and rd.lin0.out_features == 3):
        return False
    if md is not None and not (md.num_layers == 2 and md.points_factor == 0.0
                               and md.out_activation is None and f.hoist_semantic
                               and md.fc_c[0].in_features == 3 + F2 + G):
        return False
    if len(volume_feature) != 1:
        return False
    v = volume_feature[0]
    c = v.shape[1] if v.dim() == 5 else v.shape[0]
    n_scenes = getattr(ray_bundle, "num_scenes", 1)
    if isinstance(v, FoldedVolume) and v.shape[0] != n_scenes:
        return False
    rays = ray_bundle.origins.shape[0]
    return bool(c == FS + F2 and device_ok(v) and device_ok(ray_bundle.origins)
                and rays % max(n_scenes, 1) == 0 and rays > 0)


# --------------------------------------------------------------------------------------------
# the two operations
# --------------------------------------------------------------------------------------------
def _check_dims():
    assert dims() == (H, FS, F2, G, NV, NSUM), dims()


def coarse_sample(vol5, origins, dirs, nears, fars, lin_bins, t_rand, lin_u, u_rand, n_importance,
                  MW, c0, bc1, W1, b1, base_inv_s, debug=False, wfs=None):
    """-> (bins (R,S+1), starts (R,S), deltas (R,S)[, debug dict]); nothing is differentiable.
    ``wfs`` [FS, 40]: ``vol5`` is the 32-channel pre-convolution volume (``FoldedVolume.rows``) and
    wfs = [Wf_sdf | bf_sdf | 0] maps its samples to the SDF features."""
    _require_device(vol5, origins, dirs, nears, fars)
    _check_dims()
    B, Z, Y, X, C = vol5.shape
    R = origins.shape[0]
    S0 = lin_bins.numel() - 1
    S = S0 + n_importance
    dev = vol5.device
    with torch.no_grad():
        f32 = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
        vol5, origins, dirs, nears, fars = map(f32, (vol5, origins, dirs, nears, fars))
        t_rand, u_rand = f32(t_rand), f32(u_rand)
        MW, c0, bc1, W1, b1, wfs = map(f32, (MW, c0, bc1, W1, b1, wfs))
        bins = torch.empty((R, S + 1), dtype=torch.float32, device=dev)
        starts = torch.empty((R, S), dtype=torch.float32, device=dev)
        deltas = torch.empty((R, S), dtype=torch.float32, device=dev)
        dbg = None
        if debug:
            dbg = dict(idx=torch.empty((R, n_importance + 1), dtype=torch.int32, device=dev),
                       sdf=torch.empty((R, S0), dtype=torch.float32, device=dev),
                       weights=torch.empty((R, S0), dtype=torch.float32, device=dev))
        fn = _lib.lib().pv2_neus_coarse_sample
        head = (_ptr(vol5), B, Z, Y, X, C)
        if wfs is not None:
            fn = _lib.lib().pv2_neus_coarse_sample_folded
            head += (_ptr(wfs),)
        _lib.check(fn(
            *head, _ptr(origins), _ptr(dirs), _ptr(nears.reshape(-1)),
            _ptr(fars.reshape(-1)), R, S0, n_importance, _ptr(f32(lin_bins)), _ptr(t_rand),
            0 if t_rand is None else t_rand.shape[-1], _ptr(f32(lin_u)), _ptr(u_rand),
            0 if u_rand is None else u_rand.shape[-1], _ptr(MW), _ptr(c0), _ptr(bc1), _ptr(W1),
            _ptr(b1), float(base_inv_s), _ptr(bins), _ptr(starts), _ptr(deltas),
            _ptr(dbg["idx"]) if debug else None, _ptr(dbg["sdf"]) if debug else None,
            _ptr(dbg["weights"]) if debug else None, _stream(vol5)), "pv2_neus_coarse_sample")
    return (bins, starts, deltas, dbg) if debug else (bins, starts, deltas)


def _gemm_tn_into(a, b, out):
    """out[I,J] += a[M,I]^T b[M,J]  (csrc/sparse_conv.hip, atomically accumulated)."""
    m, i = a.shape
    j = b.shape[1]
    assert out.shape == (i, j) and out.is_contiguous() and a.is_contiguous() and b.is_contiguous()
    _lib.check(_lib.lib().pv2_gemm_tn(_ptr(a), _ptr(b), m, i, j, _ptr(out), _stream(a)), "pv2_gemm_tn")


class _FieldRender(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vol5, origins, dirs, starts, deltas, MW, c0, bc1, W1, b1, A, b_rgb, inv_s,
                norm_pts, norm_div):
        _require_device(vol5, origins, dirs, starts, deltas, MW, W1, A)
        _check_dims()
        L = _lib.lib()
        B, Z, Y, X, C = vol5.shape
        R, S = starts.shape
        N = R * S
        dev = vol5.device
        c = lambda t: t.detach().contiguous()
        vol5, origins, dirs, starts, deltas = map(c, (vol5, origins, dirs, starts, deltas))
        MW, c0, bc1, W1, b1, A, b_rgb = map(c, (MW, c0, bc1, W1, b1, A, b_rgb))
        inv_s_c = c(inv_s).reshape(1)
        Mt = MW[:H].t().contiguous()
        q0 = MW[H:].t().mv(W1[0]).contiguous()
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        sdf, alpha, vals = new(R, S), new(R, S), new(R, S, NV)
        sf, sh0, sa1, sq = new(N, FS), new(N, H), new(N, H), new(N, FS)
        st = _stream(vol5)
        _lib.check(L.pv2_neus_field_forward(
            _ptr(vol5), B, Z, Y, X, C, _ptr(origins), _ptr(dirs), _ptr(starts), _ptr(deltas), R, S,
            _ptr(MW), _ptr(c0), _ptr(bc1), _ptr(W1), _ptr(b1), _ptr(Mt), _ptr(q0), _ptr(A),
            _ptr(b_rgb), _ptr(inv_s_c), int(norm_pts), float(norm_div), _ptr(sdf), _ptr(alpha),
            _ptr(vals), _ptr(sf), _ptr(sh0), _ptr(sa1), _ptr(sq), st), "pv2_neus_field_forward")
        weights = new(R, S)
        _lib.check(L.pv2_raymarch_weights_forward(_ptr(alpha), R, S, _ptr(weights), None, st),
                   "pv2_raymarch_weights_forward")
        comp = new(R, NV)
        _lib.check(L.pv2_raymarch_accumulate_forward(_ptr(weights), _ptr(vals), R, S, NV, _ptr(comp),
                                                     st), "pv2_raymarch_accumulate_forward")
        ctx.save_for_backward(vol5, origins, dirs, starts, deltas, MW, W1, A, inv_s_c, sdf, alpha,
                              vals, sf, sh0, sa1, sq, weights, Mt)
        ctx.norm = (int(norm_pts), float(norm_div))
        ctx.inv_s_shape = inv_s.shape
        grad = vals[:, :, COL_G:COL_G + 3].contiguous()
        ctx.mark_non_differentiable(weights)
        return sdf, grad, weights, comp

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_sdf, g_grad, _g_weights, g_comp):
        (vol5, origins, dirs, starts, deltas, MW, W1, A, inv_s, sdf, alpha, vals, sf, sh0, sa1, sq,
         weights, Mt) = ctx.saved_tensors
        L = _lib.lib()
        B, Z, Y, X, C = vol5.shape
        R, S = starts.shape
        N = R * S
        dev = vol5.device
        st = _stream(vol5)
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        g_comp = (torch.zeros((R, NV), dtype=torch.float32, device=dev) if g_comp is None
                  else g_comp.contiguous())
        g_sdf = None if g_sdf is None else g_sdf.contiguous()
        g_grad = None if g_grad is None else g_grad.contiguous()
        gw = new(R, S)
        _lib.check(L.pv2_raymarch_accumulate_backward(_ptr(weights), _ptr(vals), _ptr(g_comp), R, S,
                                                      NV, _ptr(gw), None, st),
                   "pv2_raymarch_accumulate_backward")
        g_alpha = new(R, S)
        _lib.check(L.pv2_raymarch_weights_backward(_ptr(alpha), _ptr(gw), R, S, _ptr(g_alpha), st),
                   "pv2_raymarch_weights_backward")
        gfeat, gvec, gz, tmat = new(N, C), new(N, 4), new(N, 2 * H), new(N, H)
        gq, gh, gy, sums = new(N, FS), new(N, 68), new(N, 4), new(NSUM)
        need_vol = ctx.needs_input_grad[0]
        gvol = zeros_by_kernel(tuple(vol5.shape), torch.float32, dev) if need_vol else None
        W1gt = W1[1:].t().contiguous()
        Wc1t = MW[H:].t().contiguous()
        _lib.check(L.pv2_neus_field_backward(
            _ptr(vol5), B, Z, Y, X, C, _ptr(origins), _ptr(dirs), _ptr(starts), _ptr(deltas), R, S,
            _ptr(MW), _ptr(W1), _ptr(Mt), _ptr(W1gt), _ptr(Wc1t), _ptr(A), _ptr(inv_s), ctx.norm[0],
            ctx.norm[1], _ptr(sdf), _ptr(vals), _ptr(sh0), _ptr(sq), _ptr(weights), _ptr(g_alpha),
            _ptr(g_sdf), _ptr(g_grad), _ptr(g_comp), _ptr(gfeat), _ptr(gvec), _ptr(gz), _ptr(tmat),
            _ptr(gq), _ptr(gh), _ptr(gy), _ptr(sums), _ptr(gvol), st), "pv2_neus_field_backward")
        # weight gradients: sums over all samples of outer products = four A^T B reductions
        g_MW = zeros_by_kernel((2 * H, FS), torch.float32, dev)
        _gemm_tn_into(gz, sf, g_MW)                 # [gh0 | ga1]^T f
        _gemm_tn_into(tmat, gq, g_MW[:H])           # + t^T gq on the M rows
        g_W1p = zeros_by_kernel((68, H), torch.float32, dev)
        _gemm_tn_into(gh, sa1, g_W1p)
        g_Ap = zeros_by_kernel((4, NV), torch.float32, dev)
        _gemm_tn_into(gy, vals.reshape(N, NV), g_Ap)
        qsum = sums[SUM_Q:SUM_Q + FS]
        v1 = W1[0]
        g_MW[H:] += torch.outer(v1, qsum)
        g_W1 = g_W1p[:1 + G].clone()
        g_W1[0] += sums[SUM_V1:SUM_V1 + H] + MW[H:].mv(qsum)
        g_A = torch.cat([g_Ap[:3, COL_G:COL_G + 3], g_Ap[:3, COL_F2:COL_F2 + F2],
                         g_Ap[:3, COL_GEO:COL_GEO + G],
                         gy.reshape(R, S, 4).sum(1)[:, :3].t().mm(dirs)], dim=1)
        return (gvol, None, None, None, None, g_MW, sums[SUM_C0:SUM_C0 + H].clone(),
                sums[SUM_BC1:SUM_BC1 + H].clone(), g_W1, sums[SUM_B1:SUM_B1 + 1 + G].clone(), g_A,
                sums[SUM_RGB:SUM_RGB + 3].clone(), sums[SUM_INVS].reshape(ctx.inv_s_shape), None,
                None)


def _gemm_nt(x, w):
    """x [M,K] . w[N,K]^T on pv2_gemm_nt (K % 8 == 0)."""
    m, k = x.shape
    n = w.shape[0]
    assert w.shape[1] == k and x.is_contiguous() and w.is_contiguous()
    y = torch.empty((m, n), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().pv2_gemm_nt(_ptr(x), m, k, _ptr(w), n, None, _ptr(y), _stream(x)),
               "pv2_gemm_nt")
    return y


class _FieldRenderFolded(torch.autograd.Function):
    """``_FieldRender`` on the 32-channel volume in front of the projection network's final 1x1x1
    convolution, with that convolution (``wfp`` = [Wf | bf | 0], [128, 40]) applied per sample:
    gather [xt, s] and its spatial derivatives (pv2_neus_fold_gather), two tall GEMMs to the
    feature rows f and the Jacobian rows d f_sdf / d p, the field kernels in their rows mode; the
    backward maps gfeat / q back through Wf and scatters 32 channels (csrc/raymarch_fused.hip,
    "Folded final convolution").  Same results as the convolution followed by ``_FieldRender`` up
    to fp32 re-association (tests/test_gpu_fused_head.py)."""

    @staticmethod
    def forward(ctx, x5, wfp, origins, dirs, starts, deltas, MW, c0, bc1, W1, b1, A, b_rgb, inv_s,
                norm_pts, norm_div):
        ctx.set_materialize_grads(False)   # (the non-differentiable weights output: no [R, S] of zeros per step)
        _require_device(x5, wfp, origins, dirs, starts, deltas, MW, W1, A)
        _check_dims()
        L = _lib.lib()
        B, Z, Y, X, C = x5.shape
        assert C == KX and tuple(wfp.shape) == (FS + F2, KXP), (x5.shape, wfp.shape)
        R, S = starts.shape
        N = R * S
        dev = x5.device
        c = lambda t: t.detach().contiguous()
        x5, wfp, origins, dirs, starts, deltas = map(c, (x5, wfp, origins, dirs, starts, deltas))
        MW, c0, bc1, W1, b1, A, b_rgb = map(c, (MW, c0, bc1, W1, b1, A, b_rgb))
        inv_s_c = c(inv_s).reshape(1)
        Mt = MW[:H].t().contiguous()
        q0 = MW[H:].t().mv(W1[0]).contiguous()
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        st = _stream(x5)
        gval, gder = new(N, KXP), new(N, 3, KXP)
        _lib.check(L.pv2_neus_fold_gather(
            _ptr(x5), B, Z, Y, X, C, _ptr(origins), _ptr(dirs), _ptr(starts), R, S, int(norm_pts),
            float(norm_div), _ptr(gval), _ptr(gder), st), "pv2_neus_fold_gather")
        frows = _gemm_nt(gval, wfp)                               # f            [N, 128]
        jrows = _gemm_nt(gder.view(3 * N, KXP), wfp[:FS].contiguous())   # d f_sdf / d p [N, 3, 64]
        sdf, alpha, vals = new(R, S), new(R, S), new(R, S, NV)
        sf, sh0, sa1, sq = new(N, FS), new(N, H), new(N, H), new(N, FS)
        _lib.check(L.pv2_neus_field_forward_rows(
            _ptr(frows), _ptr(jrows), _ptr(origins), _ptr(dirs), _ptr(starts), _ptr(deltas), R, S,
            _ptr(MW), _ptr(c0), _ptr(bc1), _ptr(W1), _ptr(b1), _ptr(Mt), _ptr(q0), _ptr(A),
            _ptr(b_rgb), _ptr(inv_s_c), int(norm_pts), float(norm_div), _ptr(sdf), _ptr(alpha),
            _ptr(vals), _ptr(sf), _ptr(sh0), _ptr(sa1), _ptr(sq), st), "pv2_neus_field_forward_rows")
        weights = new(R, S)
        _lib.check(L.pv2_raymarch_weights_forward(_ptr(alpha), R, S, _ptr(weights), None, st),
                   "pv2_raymarch_weights_forward")
        comp = new(R, NV)
        _lib.check(L.pv2_raymarch_accumulate_forward(_ptr(weights), _ptr(vals), R, S, NV, _ptr(comp),
                                                     st), "pv2_raymarch_accumulate_forward")
        ctx.save_for_backward(wfp, origins, dirs, starts, deltas, MW, W1, A, inv_s_c, sdf, alpha,
                              vals, sf, sh0, sa1, sq, weights, Mt, gval, gder, jrows)
        ctx.vol_shape = tuple(x5.shape)
        ctx.norm = (int(norm_pts), float(norm_div))
        ctx.inv_s_shape = inv_s.shape
        grad = vals[:, :, COL_G:COL_G + 3].contiguous()
        ctx.mark_non_differentiable(weights)
        return sdf, grad, weights, comp

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_sdf, g_grad, _g_weights, g_comp):
        g_vol, params_fn, g_inv_s = _folded_backward(ctx.saved_tensors, ctx.vol_shape, ctx.norm, ctx.inv_s_shape,
                                                     g_sdf, g_grad, g_comp, ctx.needs_input_grad[0],
                                                     ctx.needs_input_grad[1])
        g = params_fn()
        return (g_vol, g["wfp"], None, None, None, None, g["MW"], g["c0"], g["bc1"], g["W1"], g["b1"], g["A"],
                g["b_rgb"], g_inv_s, None, None)


def _folded_backward(saved, vol_shape, norm, inv_s_shape, g_sdf, g_grad, g_comp, need_vol, need_wfp):
    """The backward of the folded head in two parts: what the CRITICAL chain needs - the gradient of the
    32-channel volume, on the caller's stream - and a closure that computes every PARAMETER gradient (the
    A^T B reductions over the saved rows and what hangs off them), which may run on another stream.
    -> (g_vol or None, params_fn() -> dict of gradients of the collapsed parameters, g_inv_s)."""
    (wfp, origins, dirs, starts, deltas, MW, W1, A, inv_s, sdf, alpha, vals, sf, sh0, sa1, sq,
     weights, Mt, gval, gder, jrows) = saved
    L = _lib.lib()
    B, Z, Y, X, C = vol_shape
    R, S = starts.shape
    N = R * S
    dev = wfp.device
    st = _stream(wfp)
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    g_comp = (torch.zeros((R, NV), dtype=torch.float32, device=dev) if g_comp is None
              else g_comp.contiguous())
    g_sdf = None if g_sdf is None else g_sdf.contiguous()
    g_grad = None if g_grad is None else g_grad.contiguous()
    gw = new(R, S)
    _lib.check(L.pv2_raymarch_accumulate_backward(_ptr(weights), _ptr(vals), _ptr(g_comp), R, S,
                                                  NV, _ptr(gw), None, st),
               "pv2_raymarch_accumulate_backward")
    g_alpha = new(R, S)
    _lib.check(L.pv2_raymarch_weights_backward(_ptr(alpha), _ptr(gw), R, S, _ptr(g_alpha), st),
               "pv2_raymarch_weights_backward")
    gfeat, gvec, gz, tmat = new(N, FS + F2), new(N, 4), new(N, 2 * H), new(N, H)
    gq, gh, gy, sums = new(N, FS), new(N, 68), new(N, 4), new(NSUM)
    W1gt = W1[1:].t().contiguous()
    Wc1t = MW[H:].t().contiguous()
    _lib.check(L.pv2_neus_field_backward_rows(
        _ptr(jrows), _ptr(origins), _ptr(dirs), _ptr(starts), _ptr(deltas), R, S, _ptr(MW),
        _ptr(W1), _ptr(Mt), _ptr(W1gt), _ptr(Wc1t), _ptr(A), _ptr(inv_s), norm[0],
        norm[1], _ptr(sdf), _ptr(vals), _ptr(sh0), _ptr(weights), _ptr(g_alpha), _ptr(g_sdf),
        _ptr(g_grad), _ptr(g_comp), _ptr(gfeat), _ptr(gvec), _ptr(gz), _ptr(tmat), _ptr(gq),
        _ptr(gh), _ptr(gy), _ptr(sums), st), "pv2_neus_field_backward_rows")
    # the folded convolution: gradient of the 32-channel volume (the chain the rest of the backward waits for)
    g_vol = None
    if need_vol:
        wf = wfp[:, :KX]
        gx = _gemm_nt(gfeat, wf.t().contiguous())              # gfeat . Wf      [N, 32]
        qx = _gemm_nt(sq, wf[:FS].t().contiguous())            # q . Wf_sdf      [N, 32]
        g_vol = zeros_by_kernel(vol_shape, torch.float32, dev)
        _lib.check(L.pv2_neus_fold_scatter(
            B, Z, Y, X, C, _ptr(origins), _ptr(dirs), _ptr(starts), R, S, norm[0], norm[1],
            _ptr(gx), _ptr(gvec), _ptr(qx), _ptr(g_vol), st), "pv2_neus_fold_scatter")

    def params_fn():
        # head weight gradients: exactly as in _FieldRender.backward
        g_MW = zeros_by_kernel((2 * H, FS), torch.float32, dev)
        _gemm_tn_into(gz, sf, g_MW)
        _gemm_tn_into(tmat, gq, g_MW[:H])
        g_W1p = zeros_by_kernel((68, H), torch.float32, dev)
        _gemm_tn_into(gh, sa1, g_W1p)
        g_Ap = zeros_by_kernel((4, NV), torch.float32, dev)
        _gemm_tn_into(gy, vals.reshape(N, NV), g_Ap)
        qsum = sums[SUM_Q:SUM_Q + FS]
        v1 = W1[0]
        g_MW[H:] += torch.outer(v1, qsum)
        g_W1 = g_W1p[:1 + G].clone()
        g_W1[0] += sums[SUM_V1:SUM_V1 + H] + MW[H:].mv(qsum)
        g_A = torch.cat([g_Ap[:3, COL_G:COL_G + 3], g_Ap[:3, COL_F2:COL_F2 + F2],
                         g_Ap[:3, COL_GEO:COL_GEO + G],
                         gy.reshape(R, S, 4).sum(1)[:, :3].t().mm(dirs)], dim=1)
        g_wfp = None
        if need_wfp:
            g_wfp = zeros_by_kernel((FS + F2, KXP), torch.float32, dev)
            _gemm_tn_into(gfeat, gval, g_wfp)                      # gfeat^T [xt, s]
            y = (gvec[:, :3, None] * gder).sum(1)                  # sum_a gg_a d[xt, s]/dp_a  [N, 40]
            _gemm_tn_into(sq, y.contiguous(), g_wfp[:FS])          # + q^T (...) on the SDF rows
        return dict(wfp=g_wfp, MW=g_MW, c0=sums[SUM_C0:SUM_C0 + H].clone(),
                    bc1=sums[SUM_BC1:SUM_BC1 + H].clone(), W1=g_W1, b1=sums[SUM_B1:SUM_B1 + 1 + G].clone(),
                    A=g_A, b_rgb=sums[SUM_RGB:SUM_RGB + 3].clone())

    params_fn.reads = (gz, sf, tmat, gq, gh, sa1, gy, vals, sums, gfeat, gval, gvec, gder, sq, dirs, W1, MW)
    return g_vol, params_fn, sums[SUM_INVS].reshape(inv_s_shape)


class _SavedProxy:
    """Stands in for ``ctx`` when one Function's forward body is reused inside another: records what it
    would have saved / set."""

    def __init__(self, real):
        self._real, self.saved = real, ()

    def save_for_backward(self, *tensors):
        self.saved = tensors

    def mark_non_differentiable(self, *tensors):
        self._real.mark_non_differentiable(*tensors)

    def set_materialize_grads(self, value):
        self._real.set_materialize_grads(value)


LEAVES_ENABLED = os.environ.get("PV2_HEAD_LEAVES", "1") != "0"


def _head_leaves(field, conv):
    """The nn.Parameters behind the collapsed head and the folded convolution, in the order
    ``_FieldRenderFoldedLeaves`` takes them - or None when one of them is not a plain fp32 device leaf."""
    sd, rd = field.sdf_decoder, field.rgb_decoder
    ps = [conv.weight, conv.bias, sd.lin0.weight, sd.lin0.bias, sd.fc_c[0].weight, sd.fc_c[0].bias,
          sd.fc_c[1].weight, sd.fc_c[1].bias, sd.lin1.weight, sd.lin1.bias, rd.lin0.weight, rd.lin0.bias,
          rd.fc_c[0].weight, rd.fc_c[0].bias, sd.fc_p.weight, sd.fc_p.bias, rd.fc_p.weight, rd.fc_p.bias]
    if any(p is None or not isinstance(p, torch.nn.Parameter) or not p.is_cuda or p.dtype != torch.float32
           or not p.requires_grad for p in ps):
        return None
    return ps


def _collapse_values(ps):
    """``collapse`` without a graph, from the leaves of ``_head_leaves`` (+ the folded convolution's
    [Wf | bf | 0]): the same torch products in the same order."""
    (cw, cb, W0, b0, Wc0, bc0, Wc1, bc1, W1, b1, Wr1, br1, Wrc, brc) = [p.detach() for p in ps[:14]]
    with torch.no_grad():
        MW = torch.cat([W0.mm(Wc0), Wc1], dim=0)
        c0 = W0.mv(bc0) + b0
        A = Wr1.mm(Wrc)
        b_rgb = Wr1.mv(brc) + br1
        wf = cw.reshape(cw.shape[0], cw.shape[1])
        pad = torch.zeros((cw.shape[0], KXP - KX - 1), dtype=wf.dtype, device=wf.device)
        wfp = torch.cat([wf, cb[:, None], pad], dim=1)
    return dict(MW=MW, c0=c0, bc1=bc1, W1=W1, b1=b1, A=A, b_rgb=b_rgb, wfp=wfp)


class _FieldRenderFoldedLeaves(torch.autograd.Function):
    """``_FieldRenderFolded`` with the parameter COLLAPSE inside the node (round 6): its inputs are the
    nn.Parameters themselves, so (i) the ~25 small torch launches of ``collapse`` / ``FoldedVolume.rows`` and
    of their autograd backward are gone from the training stream, and (ii) every parameter gradient of the
    head - four A^T B reductions over the 135 k saved rows, the folded convolution's two, the products that
    carry them back through the collapse - is a LEAF gradient that feeds nothing until the optimizer: it
    runs on the backward side stream (sidestream.py), beside the chain that carries the volume's gradient
    on to the projection network.  Same kernels, same products, same numbers as the composite."""

    @staticmethod
    def forward(ctx, x5, inv_s, cp, geom, *leaves):
        origins, dirs, starts, deltas, norm_pts, norm_div = geom
        proxy = _SavedProxy(ctx)
        out = _FieldRenderFolded.forward(proxy, x5, cp["wfp"], origins, dirs, starts, deltas, cp["MW"], cp["c0"],
                                         cp["bc1"], cp["W1"], cp["b1"], cp["A"], cp["b_rgb"], inv_s, norm_pts,
                                         norm_div)
        ctx.n_inner = len(proxy.saved)
        ctx.save_for_backward(*proxy.saved, *leaves)
        ctx.vol_shape, ctx.norm, ctx.inv_s_shape = proxy.vol_shape, proxy.norm, proxy.inv_s_shape
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_sdf, g_grad, _g_weights, g_comp):
        from . import sidestream

        saved = ctx.saved_tensors
        inner, leaves = saved[:ctx.n_inner], saved[ctx.n_inner:]
        g_vol, params_fn, g_inv_s = _folded_backward(inner, ctx.vol_shape, ctx.norm, ctx.inv_s_shape, g_sdf, g_grad,
                                                     g_comp, ctx.needs_input_grad[0], True)
        (cw, cb, W0, b0, Wc0, bc0, Wc1, bc1, W1, b1, Wr1, br1, Wrc, brc, p0, p1, p2, p3) = leaves

        def leaf_gradients():
            g = params_fn()
            top = g["MW"][:H]
            g_W0 = torch.addmm(torch.outer(g["c0"], bc0), top, Wc0.t())
            g_Wr1 = torch.addmm(torch.outer(g["b_rgb"], brc), g["A"], Wrc.t())
            zeros = torch.zeros(p0.numel() + p1.numel() + p2.numel() + p3.numel(), dtype=torch.float32,
                                device=cw.device)
            zs, o = [], 0
            for p in (p0, p1, p2, p3):     # fc_p(points) * 0.0 of the reference: an exact zero gradient
                zs.append(zeros[o:o + p.numel()].view(p.shape))
                o += p.numel()
            return (g["wfp"][:, :KX].reshape(cw.shape).contiguous(), g["wfp"][:, KX].contiguous(),
                    g_W0, g["c0"], W0.t().mm(top), W0.t().mv(g["c0"]), g["MW"][H:], g["bc1"], g["W1"], g["b1"],
                    g_Wr1, g["b_rgb"], Wr1.t().mm(g["A"]), Wr1.t().mv(g["b_rgb"]), *zs)

        ref = g_comp if g_comp is not None else inner[0]
        if sidestream.active(ref) and all(sidestream.safe_leaf(p) for p in leaves):
            grads = sidestream.fork(leaf_gradients, params_fn.reads + tuple(leaves))
        else:
            grads = leaf_gradients()
        return (g_vol, g_inv_s, None, None) + tuple(grads)


def field_render_folded(x5, wfp, origins, dirs, starts, deltas, MW, c0, bc1, W1, b1, A, b_rgb, inv_s,
                        norm_pts, norm_div):
    """``field_render`` for a ``FoldedVolume`` (``x5, wfp = volume.rows()``)."""
    return _FieldRenderFolded.apply(x5, wfp, origins, dirs, starts, deltas, MW, c0, bc1, W1, b1, A,
                                    b_rgb, inv_s, norm_pts, norm_div)


def field_render_folded_leaves(x5, inv_s, cp, geom, *leaves):
    """``field_render_folded`` with the collapse inside the node (``cp = _collapse_values(leaves)``):
    the parameter gradients leave as the gradients of the modules' own parameters."""
    return _FieldRenderFoldedLeaves.apply(x5, inv_s, cp, geom, *leaves)


def field_render(vol5, origins, dirs, starts, deltas, MW, c0, bc1, W1, b1, A, b_rgb, inv_s,
                 norm_pts, norm_div):
    """-> sdf (R,S), grad (R,S,3), weights (R,S) [no gradient], comp (R,140).  Differentiable (once)
    in the volume, the collapsed parameters and inv_s."""
    return _FieldRender.apply(vol5, origins, dirs, starts, deltas, MW, c0, bc1, W1, b1, A, b_rgb,
                              inv_s, norm_pts, norm_div)


# --------------------------------------------------------------------------------------------
# model glue
# --------------------------------------------------------------------------------------------
class _ZeroOf(torch.autograd.Function):
    """``sum_i (p_i.sum() * 0.0)`` - the reference's ``fc_p(points) * 0.0`` terms (decoders.py:27,64,99): an
    exact zero whose graph hands every ``p_i`` a gradient of zeros (so the optimiser treats the
    parameter exactly as the reference's does: weight decay included).  No forward launch; one fill
    per parameter backward - the torch expression cost four launches per decoder each way."""

    @staticmethod
    def forward(ctx, *params):
        ctx.like = [(p.shape, p.dtype, p.device) for p in params]
        return _zero_scalar(params[0])

    @staticmethod
    def backward(ctx, g):
        return tuple(torch.zeros(shape, dtype=dtype, device=dev) for shape, dtype, dev in ctx.like)


_ZERO_SCALARS = {}


def _zero_scalar(like):
    """A fresh 0-dim view of a cached zero (never written to: the callers only add it)."""
    key = (like.device, like.dtype)
    z = _ZERO_SCALARS.get(key)
    if z is None:
        z = _ZERO_SCALARS[key] = torch.zeros((), dtype=like.dtype, device=like.device)
    return z.view(())


def collapse(field):
    """Collapsed parameters of the SDF and colour heads (differentiable torch ops)."""
    sd, rd = field.sdf_decoder, field.rgb_decoder
    # fc_p(points) * 0.0 of the reference: an exact zero with a graph
    zero = _ZeroOf.apply(sd.fc_p.weight, sd.fc_p.bias, rd.fc_p.weight, rd.fc_p.bias)
    W0, b0 = sd.lin0.weight, sd.lin0.bias
    Wc0, bc0 = sd.fc_c[0].weight, sd.fc_c[0].bias
    Wc1, bc1 = sd.fc_c[1].weight, sd.fc_c[1].bias
    MW = torch.cat([W0.mm(Wc0), Wc1], dim=0)
    c0 = W0.mv(bc0) + b0 + zero
    Wr1, br1 = rd.lin0.weight, rd.lin0.bias
    Wrc, brc = rd.fc_c[0].weight, rd.fc_c[0].bias
    return dict(MW=MW, c0=c0, bc1=bc1, W1=sd.lin1.weight, b1=sd.lin1.bias, A=Wr1.mm(Wrc),
                b_rgb=Wr1.mv(brc) + br1)


def render_outputs(model, ray_bundle, volume_feature):
    """``SurfaceModel.get_outputs`` on the fused kernels (same keys and values, except that the
    coarse pass's diagnostic point sets are not materialised)."""
    # the kernels are fp32 launches that autocast never touches; the small torch ops around them
    # (parameter collapse, per-ray epilogue) stay fp32 as well
    with torch.autocast(ray_bundle.origins.device.type, enabled=False):
        return _render_outputs(model, ray_bundle, volume_feature)


def _render_outputs(model, ray_bundle, volume_feature):
    from .ponder.models.ponder.render_utils.rays import device_constant, device_linspace

    field, smp = model.field, model.sampler
    B = getattr(ray_bundle, "num_scenes", 1)
    folded = isinstance(volume_feature[0], FoldedVolume)
    leaves = wfp = None
    if folded:
        vol5 = volume_feature[0].x5()
        if LEAVES_ENABLED and model.training and vol5.is_cuda and vol5.dtype == torch.float32:
            leaves = _head_leaves(field, volume_feature[0].conv)
        if leaves is None:
            vol5, wfp = volume_feature[0].rows()
    else:
        vol5 = _vol5(volume_feature, B).float()
    o, d = ray_bundle.origins.float(), ray_bundle.directions.float()
    R = o.shape[0]
    dev = o.device
    if leaves is not None:
        # the collapse inside the node, every parameter gradient a leaf gradient on the side stream
        cp = _collapse_values(leaves)
        wfp = cp["wfp"]
    else:
        cp = collapse(field)
    S0, n_imp = smp.num_samples, smp.num_samples_importance
    ini, pdf = smp.initial_sampler, smp.pdf_sampler
    t_rand = u_rand = None
    if ini.train_stratified and ini.training:
        t_rand = ini.rand((R, 1 if ini.single_jitter else S0 + 1), dtype=o.dtype, device=dev)
    if pdf.train_stratified and pdf.training:
        u_rand = pdf.rand((R, 1 if pdf.single_jitter else n_imp + 1), device=dev)
    nb = n_imp + 1
    lin_bins = device_linspace(0.0, 1.0, S0 + 1, dev)
    lin_u = device_linspace(0.0, 1.0 - 1.0 / nb, nb, dev)
    res = coarse_sample(
        vol5, o, d, ray_bundle.nears.reshape(-1), ray_bundle.fars.reshape(-1), lin_bins, t_rand,
        lin_u, u_rand, n_imp, cp["MW"], cp["c0"], cp["bc1"], cp["W1"], cp["b1"], smp.base_variance,
        debug=CAPTURE is not None, **({"wfs": wfp[:FS]} if folded else {}))
    bins, starts, deltas = res[:3]
    if CAPTURE is not None:
        CAPTURE.update(res[3], bins=bins)
    inv_s = field.deviation_network.get_variance()
    head = (o, d, starts, deltas, cp["MW"], cp["c0"], cp["bc1"], cp["W1"], cp["b1"], cp["A"],
            cp["b_rgb"], inv_s, field.norm_pts, 1.0 + field.norm_padding + 10e-4)
    if leaves is not None:
        sdf, grad, weights, comp = field_render_folded_leaves(
            vol5, inv_s, cp, (o, d, starts, deltas, field.norm_pts, 1.0 + field.norm_padding + 10e-4), *leaves)
    elif folded:
        sdf, grad, weights, comp = field_render_folded(vol5, wfp, *head)
    else:
        sdf, grad, weights, comp = field_render(vol5, *head)
    md = field.semantic_decoder
    bg_color = model.rgb_renderer.background_color

    def per_ray_outputs():
        """rgb / semantic / depth / normal with torch ops (renderers.py:5-75) - evaluation, and whoever
        reads these entries; the training step gets its losses from the composite rows directly
        (ray_epilogue.ray_losses)."""
        # ONE split of the composite row (its backward is one cat - six slices cost a zero-fill, a copy and
        # an add each): columns f'(64) geo(64) grad(3) normal(3) rgb(3) t 1 pad
        f2c, geoc, g3c, nrm, rgbc, tcol, wsum, _ = comp.split([F2, G, 3, 3, 3, 1, 1, NV - COL_ONE - 1], dim=1)
        res = {}
        bg = device_constant(bg_color, dev, comp.dtype)
        rgb = torch.addcmul(rgbc + bg, wsum, bg, value=-1.0)      # rgb + bg * (1 - wsum)
        res["rgb"] = rgb if model.training else rgb.clamp(0.0, 1.0)
        if md is not None:
            xbar = torch.cat([g3c, f2c, geoc], dim=1)
            zero = _ZeroOf.apply(md.fc_p.weight, md.fc_p.bias)
            hidden = F.linear(xbar, md.fc_c[0].weight) + (md.fc_c[0].bias + zero) * wsum
            lin = md.last_linear
            res["semantic"] = F.linear(hidden, lin.weight) + lin.bias * wsum
        depth = tcol / (wsum + 1e-10)
        lo, hi = torch.aminmax(starts.reshape(B, -1), dim=1)          # per scene: nearest / farthest sample
        lo = lo[:, None].expand(B, R // B).reshape(-1, 1)
        hi = hi[:, None].expand(B, R // B).reshape(-1, 1)
        res["depth"] = torch.clamp(depth, lo, hi)
        res["normal"] = nrm
        return res

    out = dict(weights=weights.unsqueeze(-1), sdf=sdf.unsqueeze(-1), gradients=grad,
               z_vals=starts.unsqueeze(-1))
    if model.training and ray_epilogue.ENABLED:
        # the per-ray entries on demand; the loss terms straight from the composite rows
        fused = dict(comp=comp, starts=starts, sdf=sdf, grad=grad, semantic=md, n_f2=F2, n_geo=G,
                     num_scenes=B, background=tuple(float(v) for v in bg_color))
        return ray_epilogue.RenderOutputs(out, fused, per_ray_outputs)
    out.update(per_ray_outputs())
    if not model.training:
        out["sampled_points"] = o[:, None, :] + d[:, None, :] * starts[..., None]
    return out
