"""Alpha compositing along rays on the gfx950 kernels of csrc/raymarch.hip.

``composite_weights(alphas)`` and ``weighted_sum(weights, values)`` are the two operations the
NeuS head's compositing is made of (render_utils/rays.py ``alphas_to_weights``, the renderers'
``sum_s w_s * value_s``).  Each is ONE launch forward and one backward; they are once
differentiable, which is all the head needs (second-order terms enter through the SDF gradient
that feeds ``alphas``, upstream of here).  Device fp32 only - no fallback in this module; the
callers decide when to use it.
"""
import os

import torch

from . import _lib
from .kernels import _ptr, _require_device, _stream


# The modular head's compositing (rays.alphas_to_weights, renderers.*) goes through these kernels
# where `supported` (validated on MI355X in round 2); PV2_FUSED_COMPOSITE=0 restores the torch ops.
ENABLED = os.environ.get("PV2_FUSED_COMPOSITE", "1") != "0"


class _CompositeWeights(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alphas):  # (R, S, 1) -> weights (R, S, 1), transmittance (R, S + 1, 1)
        _require_device(alphas)
        a = alphas.contiguous()
        r, s = a.shape[0], a.shape[1]
        w = torch.empty_like(a)
        t = torch.empty((r, s + 1, 1), dtype=a.dtype, device=a.device)
        _lib.check(_lib.lib().pv2_raymarch_weights_forward(_ptr(a), r, s, _ptr(w), _ptr(t),
                                                           _stream(a)), "pv2_raymarch_weights_forward")
        ctx.save_for_backward(a)
        ctx.mark_non_differentiable(t)
        return w, t

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gw, _gt):
        (a,) = ctx.saved_tensors
        gw = gw.contiguous()
        ga = torch.empty_like(a)
        _lib.check(_lib.lib().pv2_raymarch_weights_backward(_ptr(a), _ptr(gw), a.shape[0], a.shape[1],
                                                            _ptr(ga), _stream(a)),
                   "pv2_raymarch_weights_backward")
        return ga


class _WeightedSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, values):  # (R, S, 1), (R, S, F) -> (R, F)
        _require_device(weights, values)
        w, x = weights.contiguous(), values.contiguous()
        r, s, f = x.shape
        out = torch.empty((r, f), dtype=x.dtype, device=x.device)
        _lib.check(_lib.lib().pv2_raymarch_accumulate_forward(_ptr(w), _ptr(x), r, s, f, _ptr(out),
                                                              _stream(x)),
                   "pv2_raymarch_accumulate_forward")
        ctx.save_for_backward(w, x)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        w, x = ctx.saved_tensors
        r, s, f = x.shape
        gout = gout.contiguous()
        gw = torch.empty_like(w) if ctx.needs_input_grad[0] else None
        gx = torch.empty_like(x) if ctx.needs_input_grad[1] else None
        _lib.check(_lib.lib().pv2_raymarch_accumulate_backward(
            _ptr(w), _ptr(x), _ptr(gout), r, s, f, _ptr(gw), _ptr(gx), _stream(x)),
            "pv2_raymarch_accumulate_backward")
        return gw, gx


def supported(alphas_or_weights, values=None):
    """fp32 device tensors with at most 256 samples per ray (and at most 512 features)."""
    t = alphas_or_weights
    ok = t.is_cuda and t.dtype == torch.float32 and t.dim() == 3 and t.shape[-1] == 1 and t.shape[1] <= 256
    if values is not None:
        ok = ok and values.is_cuda and values.dtype == torch.float32 and values.dim() == 3 \
            and values.shape[:2] == t.shape[:2] and values.shape[-1] <= 512
    return bool(ok) and not torch.is_autocast_enabled()


def composite_weights(alphas):
    """(weights, transmittance) of ``alphas`` (R, S, 1)."""
    return _CompositeWeights.apply(alphas)


def weighted_sum(weights, values):
    """sum over samples of weights (R, S, 1) * values (R, S, F) -> (R, F)."""
    return _WeightedSum.apply(weights, values)
