"""The NeuS render head for NARROW SDF decoders on the fused kernels of csrc/raymarch_narrow.hip.

Serves the head the reference's nuScenes configuration builds (configs/nuscenes/pretrain-ponder-
spunet-v1m1-0-base.py: ``SDFField(sdf_decoder=dict(in_dim=32, out_dim=16 + 1, hidden_size=16,
n_blocks=5), share_volume=True)`` without colour / semantic decoders, depth loss only) - what
ponder/models/ponder/render_utils/{ray_samplers.py:355-463, fields/sdf_field.py:185-284, 122-146,
decoders.py:6-36, rays.py:83-105, renderers.py:33-45} run as ~500 small autograd ops per step,
including the double backward behind ``grad sdf``.  Here: one launch for the coarse pass +
importance sampling, two for the main pass forward (per-sample SDF and gradient; per-ray alphas,
weights, depth sums), two for its backward (hand-derived, second-order terms included), one sum
over the parameter-gradient slabs.

The decoder's parameters travel as ONE flat vector built by ``torch.cat`` (``pack_theta``), so
autograd splits the kernels' gradient back onto the nn.Linear parameters.  Every other head shape
keeps the modular path; there is no host fallback (tests install host doubles from oracle/ over
``coarse_sample`` / ``field_render``).
"""
import ctypes
import os

import torch

from . import _lib
from .fused_head import _vol5
from .kernels import _ptr, _require_device, _stream

ENABLED = os.environ.get("PV2_NARROW_HEAD", "1") != "0"
CALLS = 0        # renders served (tests / bench read it)
CAPTURE = None   # tests set this to a dict to receive the coarse pass's diagnostics

C, H, L, NTHETA = 32, 16, 6, 4609   # checked against the library on first device use
_DIMS = None


def device_ok(t):
    """Tensors the kernels take (overridden by the host doubles in tests)."""
    return t.is_cuda and t.dtype in (torch.float32, torch.bfloat16, torch.float16)


def dims():
    global _DIMS
    if _DIMS is None:
        vals = [ctypes.c_int() for _ in range(4)]
        _lib.check(_lib.lib().pv2_narrow_head_dims(*[ctypes.byref(v) for v in vals]),
                   "pv2_narrow_head_dims")
        _DIMS = tuple(v.value for v in vals)
    return _DIMS


def _check_dims():
    assert dims() == (C, H, L, NTHETA), dims()


def pack_theta(sd):
    """Flat parameter vector of an SDFDecoder in the kernels' layout (differentiable):
    Wp bp | Wc_l bc_l (l < L) | W_l b_l (l < L-1) | row 0 of the last linear layer, its bias."""
    n = sd.num_layers - 1
    parts = [sd.fc_p.weight.reshape(-1), sd.fc_p.bias]
    for l in range(n):
        parts += [sd.fc_c[l].weight.reshape(-1), sd.fc_c[l].bias]
    for l in range(n - 1):
        lin = getattr(sd, f"lin{l}")
        parts += [lin.weight.reshape(-1), lin.bias]
    last = sd.last_linear
    parts += [last.weight[0], last.bias[0:1]]
    return torch.cat(parts).float()


def usable(model, ray_bundle, volume_feature):
    """True when ``model`` (a NeuSModel) has exactly the head these kernels implement."""
    from .ponder.models.ponder.render_utils.ray_samplers import NeuSSampler, UniformSampler

    if not ENABLED:
        return False
    f, smp = model.field, model.sampler
    sd = f.sdf_decoder
    if not (isinstance(smp, NeuSSampler) and isinstance(smp.initial_sampler, UniformSampler)
            and smp.num_upsample_steps == 1 and 2 <= smp.num_samples <= 128
            and 1 <= smp.num_samples_importance <= 63):
        return False
    if not (f.volume_type == "default" and f.padding_mode == "zeros" and f.share_volume
            and f.use_gradient and f._cos_anneal_ratio == 1.0 and not f.norm_pts
            and f.rgb_decoder is None and f.semantic_decoder is None):
        return False
    lw = model.loss.weights
    # (losses on per-sample outputs other than through the composited depth are served too - the
    # backward takes upstream gradients of sdf / gradients / weights - but not the extra SDF query)
    if lw.get("sparse_points_sdf_loss", 0.0) > 0 or lw.get("rgb_loss", 0.0) > 0 \
            or lw.get("semantic_loss", 0.0) > 0:
        return False
    if not (sd.num_layers == L + 1 and sd.fc_p.out_features == H and sd.fc_c[0].in_features == C
            and all(getattr(sd, f"lin{l}").out_features == H for l in range(L - 1))
            and sd.last_linear.in_features == H):
        return False
    if len(volume_feature) != 1:
        return False
    v = volume_feature[0]
    if not torch.is_tensor(v):
        return False
    c = v.shape[1] if v.dim() == 5 else v.shape[0]
    n_scenes = getattr(ray_bundle, "num_scenes", 1)
    rays = ray_bundle.origins.shape[0]
    return bool(c == C and device_ok(v) and device_ok(ray_bundle.origins)
                and rays % max(n_scenes, 1) == 0 and rays > 0)


# --------------------------------------------------------------------------------------------
# the two operations
# --------------------------------------------------------------------------------------------
def coarse_sample(vol5, origins, dirs, nears, fars, lin_bins, t_rand, lin_u, u_rand, n_importance,
                  theta, points_factor, base_inv_s, debug=False):
    """-> (bins (R,S+1), starts (R,S), deltas (R,S)[, debug dict]); nothing is differentiable."""
    _require_device(vol5, origins, dirs, nears, fars)
    _check_dims()
    B, Z, Y, X, Cv = vol5.shape
    R = origins.shape[0]
    S0 = lin_bins.numel() - 1
    S = S0 + n_importance
    dev = vol5.device
    with torch.no_grad():
        f32 = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
        vol5, origins, dirs, nears, fars = map(f32, (vol5, origins, dirs, nears, fars))
        t_rand, u_rand, theta = f32(t_rand), f32(u_rand), f32(theta)
        bins = torch.empty((R, S + 1), dtype=torch.float32, device=dev)
        starts = torch.empty((R, S), dtype=torch.float32, device=dev)
        deltas = torch.empty((R, S), dtype=torch.float32, device=dev)
        dbg = None
        if debug:
            dbg = dict(idx=torch.empty((R, n_importance + 1), dtype=torch.int32, device=dev),
                       sdf=torch.empty((R, S0), dtype=torch.float32, device=dev),
                       weights=torch.empty((R, S0), dtype=torch.float32, device=dev))
        _lib.check(_lib.lib().pv2_narrow_coarse_sample(
            _ptr(vol5), B, Z, Y, X, Cv, _ptr(origins), _ptr(dirs), _ptr(nears.reshape(-1)),
            _ptr(fars.reshape(-1)), R, S0, n_importance, _ptr(f32(lin_bins)), _ptr(t_rand),
            0 if t_rand is None else t_rand.shape[-1], _ptr(f32(lin_u)), _ptr(u_rand),
            0 if u_rand is None else u_rand.shape[-1], _ptr(theta), float(points_factor),
            float(base_inv_s), _ptr(bins), _ptr(starts), _ptr(deltas),
            _ptr(dbg["idx"]) if debug else None, _ptr(dbg["sdf"]) if debug else None,
            _ptr(dbg["weights"]) if debug else None, _stream(vol5)), "pv2_narrow_coarse_sample")
    return (bins, starts, deltas, dbg) if debug else (bins, starts, deltas)


class _NarrowRender(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vol5, theta, inv_s, origins, dirs, starts, deltas, points_factor):
        _require_device(vol5, origins, dirs, starts, deltas, theta)
        _check_dims()
        B, Z, Y, X, Cv = vol5.shape
        R, S = starts.shape
        dev = vol5.device
        c = lambda t: t.detach().contiguous()
        vol5, theta, origins, dirs, starts, deltas = map(c, (vol5, theta, origins, dirs, starts, deltas))
        inv_s_c = c(inv_s).reshape(1)
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        sdf, grad, weights, trans, comp = new(R, S), new(R, S, 3), new(R, S), new(R, S), new(R, 2)
        _lib.check(_lib.lib().pv2_narrow_field_forward(
            _ptr(vol5), B, Z, Y, X, Cv, _ptr(origins), _ptr(dirs), _ptr(starts), _ptr(deltas), R, S,
            _ptr(theta), float(points_factor), _ptr(inv_s_c), _ptr(sdf), _ptr(grad), _ptr(weights),
            _ptr(trans), _ptr(comp), _stream(vol5)), "pv2_narrow_field_forward")
        ctx.save_for_backward(vol5, theta, inv_s_c, origins, dirs, starts, deltas, sdf, grad, weights,
                              trans)
        ctx.points_factor = float(points_factor)
        ctx.inv_s_shape = inv_s.shape
        ctx.mark_non_differentiable(trans)
        return sdf, grad, weights, comp, trans

    @staticmethod
    def backward(ctx, g_sdf, g_grad, g_weights, g_comp, _g_trans):
        vol5, theta, inv_s_c, origins, dirs, starts, deltas, sdf, grad, weights, trans = ctx.saved_tensors
        B, Z, Y, X, Cv = vol5.shape
        R, S = starts.shape
        dev = vol5.device
        lib = _lib.lib()
        opt = lambda t: None if t is None else t.contiguous()
        g_sdf, g_grad, g_weights = opt(g_sdf), opt(g_grad), opt(g_weights)
        g_comp = (torch.zeros((R, 2), dtype=torch.float32, device=dev) if g_comp is None
                  else g_comp.contiguous())
        n_slabs = int(lib.pv2_narrow_backward_slabs(R, S))
        work = torch.empty((R * S, 4), dtype=torch.float32, device=dev)
        slabs = torch.empty((n_slabs, NTHETA), dtype=torch.float32, device=dev)
        ginv = torch.empty((R,), dtype=torch.float32, device=dev)
        g_vol = torch.zeros_like(vol5) if ctx.needs_input_grad[0] else None
        _lib.check(lib.pv2_narrow_field_backward(
            _ptr(vol5), B, Z, Y, X, Cv, _ptr(origins), _ptr(dirs), _ptr(starts), _ptr(deltas), R, S,
            _ptr(theta), ctx.points_factor, _ptr(inv_s_c), _ptr(sdf), _ptr(grad), _ptr(weights),
            _ptr(trans), _ptr(g_comp), _ptr(g_weights), _ptr(g_sdf), _ptr(g_grad), _ptr(work),
            _ptr(g_vol), _ptr(slabs), _ptr(ginv), _stream(vol5)), "pv2_narrow_field_backward")
        g_theta = slabs.sum(0) if ctx.needs_input_grad[1] else None
        g_inv = ginv.sum().reshape(ctx.inv_s_shape) if ctx.needs_input_grad[2] else None
        return g_vol, g_theta, g_inv, None, None, None, None, None


def field_render(vol5, theta, inv_s, origins, dirs, starts, deltas, points_factor):
    """-> sdf (R,S), grad (R,S,3), weights (R,S), comp (R,2) = [sum w t, sum w]; differentiable in
    vol5, theta, inv_s (second-order terms through grad sdf included)."""
    return _NarrowRender.apply(vol5, theta, inv_s, origins, dirs, starts, deltas, points_factor)[:4]


def render_outputs(model, ray_bundle, volume_feature):
    """``SurfaceModel.get_outputs`` for this head (same keys and values, except that the coarse
    pass's diagnostic point sets are not materialised)."""
    with torch.autocast(ray_bundle.origins.device.type, enabled=False):
        return _render_outputs(model, ray_bundle, volume_feature)


def _render_outputs(model, ray_bundle, volume_feature):
    from .ponder.models.ponder.render_utils.rays import device_linspace

    global CALLS
    CALLS += 1
    field, smp = model.field, model.sampler
    B = getattr(ray_bundle, "num_scenes", 1)
    vol5 = _vol5(volume_feature, B).float()
    o, d = ray_bundle.origins.float(), ray_bundle.directions.float()
    R = o.shape[0]
    dev = o.device
    sd = field.sdf_decoder
    theta = pack_theta(sd)
    pf = float(sd.points_factor)
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
    res = coarse_sample(vol5, o, d, ray_bundle.nears.reshape(-1), ray_bundle.fars.reshape(-1),
                        lin_bins, t_rand, lin_u, u_rand, n_imp, theta, pf, smp.base_variance,
                        debug=CAPTURE is not None)
    bins, starts, deltas = res[:3]
    if CAPTURE is not None:
        CAPTURE.update(res[3], bins=bins)
    inv_s = field.deviation_network.get_variance()
    sdf, grad, weights, comp = field_render(vol5, theta, inv_s, o, d, starts, deltas, pf)
    depth = comp[:, 0:1] / (comp[:, 1:2] + 1e-10)
    per_scene = starts.reshape(B, -1)
    lo = per_scene.amin(1).repeat_interleave(R // B).reshape(-1, 1)
    hi = per_scene.amax(1).repeat_interleave(R // B).reshape(-1, 1)
    out = dict(depth=torch.maximum(torch.minimum(depth, hi), lo))
    out.update(weights=weights.unsqueeze(-1), sdf=sdf.unsqueeze(-1), gradients=grad,
               z_vals=starts.unsqueeze(-1))
    # (the composited normal feeds no loss of this head; kept for the callers that read it)
    out["normal"] = (weights.unsqueeze(-1) * torch.nn.functional.normalize(grad, dim=-1)).sum(1)
    if not model.training:
        out["sampled_points"] = o[:, None, :] + d[:, None, :] * starts[..., None]
    return out
