"""The per-ray / per-sample loss terms of the surface-rendering models on csrc/surface_loss.hip:
depth, colour (+ psnr), free-space, SDF and eikonal terms of
ponder/models/ponder/render_utils/models/base_surface_model.py:102-211 in two launches forward and one
backward instead of ~60 + ~100 small ATen launches (2 ms of host time on a host-bound step).  The
semantic term (a matrix product and a cross entropy) stays with the library; so does every input the
kernels do not take (host tensors, other dtypes): ``SurfaceModel.get_loss`` then evaluates the same
formulas with torch ops."""
import ctypes
import os

import torch

from . import _lib
from .kernels import _ptr, _stream

ENABLED = os.environ.get("PV2_FUSED_LOSS", "1") != "0"
CALLS = 0
TERMS = ("depth_loss", "rgb_loss", "psnr", "free_space_loss", "sdf_loss", "eikonal_loss")
_WEIGHTS = {}


def usable(preds, targets):
    if not ENABLED:
        return False
    need = [preds.get("depth"), preds.get("sdf"), preds.get("z_vals"), targets.get("depth")]
    if any(t is None for t in need):
        return False
    ts = need + [t for t in (preds.get("rgb"), preds.get("gradients"), targets.get("rgb")) if t is not None]
    return all(t.is_cuda and t.dtype == torch.float32 for t in ts) and preds["sdf"].dim() == 3


def _weights_on(device, w):
    key = (device, w)
    if key not in _WEIGHTS:
        _WEIGHTS[key] = torch.tensor(w, dtype=torch.float32, device=device)
    return _WEIGHTS[key]


class _SurfaceLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, rgb, sdf, grad, depth_gt, rgb_gt, z, trunc, weights):
        dev = depth.device
        c = lambda t: None if t is None else t.detach().contiguous()
        depth, rgb, sdf, grad, depth_gt, rgb_gt, z = map(c, (depth, rgb, sdf, grad, depth_gt, rgb_gt, z))
        R, S = sdf.shape[0], sdf.shape[1]
        lib = _lib.lib()
        w = _weights_on(dev, weights)
        ws = torch.empty(int(lib.pv2_surface_loss_workspace_floats()), dtype=torch.float32, device=dev)
        out = torch.empty(6, dtype=torch.float32, device=dev)
        sums = torch.empty(9, dtype=torch.float32, device=dev)
        _lib.check(lib.pv2_surface_loss_forward(
            _ptr(depth), _ptr(depth_gt), _ptr(rgb), _ptr(rgb_gt), _ptr(sdf), _ptr(z), _ptr(grad), R, S,
            float(trunc), _ptr(w), _ptr(ws), _ptr(out), _ptr(sums), _stream(depth)),
            "pv2_surface_loss_forward")
        ctx.save_for_backward(depth, depth_gt, sdf, z, w, sums, *[t for t in (rgb, rgb_gt, grad) if t is not None])
        ctx.has = (rgb is not None, grad is not None)
        ctx.trunc = float(trunc)
        return out.unbind(0)

    @staticmethod
    def backward(ctx, *ups):
        saved = list(ctx.saved_tensors)
        depth, depth_gt, sdf, z, w, sums = saved[:6]
        rest = saved[6:]
        rgb = rgb_gt = grad = None
        if ctx.has[0]:
            rgb, rgb_gt = rest[0], rest[1]
            rest = rest[2:]
        if ctx.has[1]:
            grad = rest[0]
        R, S = sdf.shape[0], sdf.shape[1]
        ups = [None if u is None else u.contiguous() for u in ups]
        arr = (ctypes.c_void_p * 6)(*[None if u is None else u.data_ptr() for u in ups])
        g_depth = torch.empty_like(depth)
        g_rgb = torch.empty_like(rgb) if rgb is not None else None
        g_sdf = torch.empty_like(sdf)
        g_grad = torch.empty_like(grad) if grad is not None else None
        _lib.check(_lib.lib().pv2_surface_loss_backward(
            _ptr(depth), _ptr(depth_gt), _ptr(rgb), _ptr(rgb_gt), _ptr(sdf), _ptr(z), _ptr(grad), R, S,
            ctx.trunc, _ptr(w), _ptr(sums), arr, _ptr(g_depth), _ptr(g_rgb), _ptr(g_sdf), _ptr(g_grad),
            _stream(depth)), "pv2_surface_loss_backward")
        return g_depth, g_rgb, g_sdf, g_grad, None, None, None, None, None


def surface_losses(preds, targets, loss_cfg):
    """dict of the fused terms (only those with a positive weight, plus ``psnr`` with the colour term),
    values equal to SurfaceModel.get_loss's up to fp32 summation order."""
    global CALLS
    CALLS += 1
    lw = loss_cfg.weights
    wt = tuple(float(lw.get(k, 0.0)) for k in ("depth_loss", "rgb_loss", "free_space_loss", "sdf_loss",
                                                "eikonal_loss"))
    rgb = preds.get("rgb") if wt[1] > 0 else None
    grad = preds.get("gradients") if wt[4] > 0 else None
    vals = _SurfaceLoss.apply(preds["depth"], rgb, preds["sdf"], grad, targets["depth"],
                              targets.get("rgb") if rgb is not None else None, preds["z_vals"],
                              loss_cfg.sensor_depth_truncation, wt)
    keep = dict(depth_loss=wt[0] > 0, rgb_loss=wt[1] > 0, psnr=wt[1] > 0, free_space_loss=wt[2] > 0,
                sdf_loss=wt[3] > 0, eikonal_loss=wt[4] > 0)
    return {k: v for k, v in zip(TERMS, vals) if keep[k]}
