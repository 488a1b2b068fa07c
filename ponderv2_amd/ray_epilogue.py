"""Per-ray epilogue of the fused ray march + every loss term of the surface model as ONE autograd node
(csrc/ray_epilogue.hip, csrc/surface_loss.hip).

The reference runs, after compositing, RGBRenderer / DepthRenderer / SemanticRenderer
(ponder/models/ponder/render_utils/renderers.py:5-75) and ``SurfaceModel.get_loss``
(render_utils/models/base_surface_model.py:102-211) as ~80 small torch ops forward and ~90 backward; on
MI355X each is a 5 - 10 us launch on the training stream.  Here ``fused_head._render_outputs`` hands
back a ``RenderOutputs`` dict that still knows the composite rows; ``SurfaceModel.get_loss`` gives it to
``ray_losses`` below, which produces every term (and their sum, in the model's order of addition) with
11 launches forward and ~13 backward, three + four of them library matrix products:

    bounds, rows      lo / hi of the scene's samples; xbar = [grad | f' | geo | sum w], rgb, depth
    sem = xbar M^T    the semantic head - linear, so composited before it - as ONE product:
                      M = [W_last W_c | W_last b_c + b_last], the biases ride on xbar's sum-w column
    raw = sem gt^T    the contrastive logits before the row scale 1 / (max(|sem_i|, 1e-12) T)
    cross entropy, surface terms (2 launches), finalize

Anything the node does not cover (other head shapes, host tensors, evaluation, a consumer that reads
``out["rgb"]``) takes the torch statement: ``RenderOutputs`` materialises the missing entries on demand.
"""
import ctypes
import os

import torch

from . import _lib, surface_loss
from .kernels import _ptr, _stream

ENABLED = os.environ.get("PV2_FUSED_RAY_LOSS", "1") != "0"
CALLS = 0
OUT_TERMS = ("depth_loss", "rgb_loss", "psnr", "semantic_loss", "free_space_loss", "sdf_loss",
             "eikonal_loss")


class RenderOutputs(dict):
    """The render head's output dict with the per-ray entries (rgb, depth, normal, semantic) computed on
    first use by ``materialize`` (the torch statement), and ``fused`` - what ``ray_losses`` needs to
    compute the losses without them."""

    def __init__(self, eager, fused, materialize):
        super().__init__(eager)
        self.fused = fused
        self._materialize = materialize

    def _fill(self):
        fn, self._materialize = self._materialize, None
        if fn is not None:
            for k, v in fn().items():
                dict.setdefault(self, k, v)

    def __missing__(self, key):
        self._fill()
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        if not dict.__contains__(self, key):
            self._fill()
        return dict.get(self, key, default)

    def __contains__(self, key):
        if not dict.__contains__(self, key):
            self._fill()
        return dict.__contains__(self, key)

    def keys(self):
        self._fill()
        return dict.keys(self)

    def items(self):
        self._fill()
        return dict.items(self)

    def values(self):
        self._fill()
        return dict.values(self)

    def __iter__(self):
        self._fill()
        return dict.__iter__(self)

    def __len__(self):
        self._fill()
        return dict.__len__(self)


class LossDict(dict):
    """Loss terms + ``total``: their sum, formed on the device in the model's order of addition."""
    total = None


def usable(preds, targets, loss_cfg):
    fused = getattr(preds, "fused", None)
    if not ENABLED or fused is None:
        return False
    lw = loss_cfg.weights
    if lw.get("sparse_points_sdf_loss", 0.0) > 0:
        return False
    ts = [fused["comp"], fused["starts"], fused["sdf"], fused["grad"], targets.get("depth"),
          targets.get("rgb")]
    if fused["semantic"] is not None and lw.get("semantic_loss", 0.0) > 0:
        ts.append(targets.get("semantic"))
    return all(t is not None and t.is_cuda and t.dtype == torch.float32 for t in ts)


class _RayLoss(torch.autograd.Function):
    """(comp, sdf, grad, W_c, b_c, W_last, b_last, fc_p.weight, fc_p.bias) -> the nine scalars of
    ``pv2_ray_loss_finalize``.  The semantic parameters may be None (no semantic head / zero weight)."""

    @staticmethod
    def forward(ctx, comp, sdf, grad, Wc, bc, Wl, bl, Wp, bp, starts, depth_gt, rgb_gt, sem_gt, cfg):
        ctx.set_materialize_grads(False)   # (eight outputs, usually only the total carries a gradient: no zero fills)
        L = _lib.lib()
        dev = comp.device
        st = _stream(comp)
        c = lambda t: None if t is None else t.detach().contiguous()
        comp, sdf, grad, starts = map(c, (comp, sdf, grad, starts))
        depth_gt, rgb_gt, sem_gt = c(depth_gt).reshape(-1), c(rgb_gt), c(sem_gt)
        R, S = starts.shape
        nv, n_f2, n_geo, B = comp.shape[1], cfg["n_f2"], cfg["n_geo"], cfg["num_scenes"]
        nx = 3 + n_f2 + n_geo + 1
        bg = cfg["background"]
        wt = cfg["weights"]           # depth, rgb, free_space, sdf, eikonal
        w_sem = cfg["w_sem"]
        has_sem = Wc is not None and w_sem > 0
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        lohi, xbar, rgb, depth = new(B, 2), new(R, nx), new(R, 3), new(R)
        _lib.check(L.pv2_ray_rows_forward(_ptr(comp), nv, n_f2, n_geo, _ptr(starts), R, S, B, bg[0], bg[1],
                                          bg[2], _ptr(lohi), _ptr(xbar), _ptr(rgb), _ptr(depth), st),
                   "pv2_ray_rows_forward")
        info = sem = raw = M_ext = Wc_ext = None
        if has_sem:
            Wc_ext = torch.cat([Wc.detach(), bc.detach()[:, None]], dim=1)        # [hidden, nx]
            M_ext = Wl.detach().mm(Wc_ext)                                         # [c_sem, nx]
            M_ext[:, nx - 1] += bl.detach()
            sem = xbar.mm(M_ext.t())                                               # [R, c_sem]
            raw = sem.mm(sem_gt.t())                                               # [R, R]
            info = new(R, int(L.pv2_ray_loss_info_floats()))
            _lib.check(L.pv2_semantic_ce_forward(_ptr(raw), _ptr(sem), _ptr(sem_gt), _ptr(depth_gt), R,
                                                 sem.shape[1], float(cfg["temperature"]), _ptr(info), st),
                       "pv2_semantic_ce_forward")
        use_rgb, use_grad = wt[1] > 0, wt[4] > 0 and grad is not None
        w = surface_loss._weights_on(dev, tuple(wt))
        ws = new(int(L.pv2_surface_loss_workspace_floats()))
        surf, sums, out = new(6), new(9), new(9)
        _lib.check(L.pv2_surface_loss_forward(
            _ptr(depth), _ptr(depth_gt), _ptr(rgb) if use_rgb else None, _ptr(rgb_gt) if use_rgb else None,
            _ptr(sdf), _ptr(starts), _ptr(grad) if use_grad else None, R, S, float(cfg["trunc"]), _ptr(w),
            _ptr(ws), _ptr(surf), _ptr(sums), st), "pv2_surface_loss_forward")
        _lib.check(L.pv2_ray_loss_finalize(_ptr(info), R, float(w_sem), _ptr(surf), _ptr(out), st),
                   "pv2_ray_loss_finalize")
        ctx.save_for_backward(comp, sdf, grad, starts, depth_gt, rgb_gt, sem_gt, lohi, xbar, rgb, depth,
                              info, sem, raw, M_ext, Wc_ext, Wl, w, sums, out)
        ctx.cfg = cfg
        ctx.flags = (has_sem, use_rgb, use_grad)
        ctx.param_like = [None if p is None else (p.shape, p.numel()) for p in (Wp, bp)]
        return tuple(out.unbind(0))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *ups):
        (comp, sdf, grad, starts, depth_gt, rgb_gt, sem_gt, lohi, xbar, rgb, depth, info, sem, raw, M_ext,
         Wc_ext, Wl, w, sums, out) = ctx.saved_tensors
        cfg = ctx.cfg
        has_sem, use_rgb, use_grad = ctx.flags
        L = _lib.lib()
        dev = comp.device
        st = _stream(comp)
        R, S = starts.shape
        nv, n_f2, n_geo, B = comp.shape[1], cfg["n_f2"], cfg["n_geo"], cfg["num_scenes"]
        nx = 3 + n_f2 + n_geo + 1
        bg = cfg["background"]
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        # upstream of term k = its own + the total's (the usual case: only the total carries one)
        g_tot = ups[7]
        if all(u is None for u in ups[:7]) and g_tot is not None:
            gs = [g_tot.reshape(1).contiguous()] * 7
        else:
            zero = torch.zeros(1, dtype=torch.float32, device=dev)
            base = zero if g_tot is None else g_tot.reshape(1)
            gs = [(base if u is None else base + u.reshape(1)).contiguous() for u in ups[:7]]
        # out order: depth, rgb, psnr, semantic, free_space, sdf, eikonal; the surface kernel's: without semantic
        surf_ups = [gs[0], gs[1], None if ups[2] is None else ups[2].reshape(1).contiguous(), gs[4], gs[5], gs[6]]
        arr = (ctypes.c_void_p * 6)(*[None if u is None else u.data_ptr() for u in surf_ups])
        g_depth, g_sdf = new(R), torch.empty_like(sdf)
        g_rgb = new(R, 3) if use_rgb else None
        g_grad = torch.empty_like(grad) if use_grad else None
        _lib.check(L.pv2_surface_loss_backward(
            _ptr(depth), _ptr(depth_gt), _ptr(rgb) if use_rgb else None, _ptr(rgb_gt) if use_rgb else None,
            _ptr(sdf), _ptr(starts), _ptr(grad) if use_grad else None, R, S, float(cfg["trunc"]), _ptr(w),
            _ptr(sums), arr, _ptr(g_depth), _ptr(g_rgb), _ptr(g_sdf), _ptr(g_grad), st),
            "pv2_surface_loss_backward")
        d_xbar = g_Wc = g_bc = g_Wl = g_bl = None
        if has_sem:
            C = sem.shape[1]
            d_raw, d_sem = new(R, R), new(R, C)
            _lib.check(L.pv2_semantic_ce_backward(_ptr(raw), _ptr(sem), _ptr(info), _ptr(gs[3]), _ptr(out), R,
                                                  C, float(cfg["w_sem"]), _ptr(d_raw), _ptr(d_sem), st),
                       "pv2_semantic_ce_backward")
            d_sem.addmm_(d_raw, sem_gt)                    # + d_raw . gt
            dM_ext = d_sem.t().mm(xbar)                    # [c_sem, nx]
            d_xbar = d_sem.mm(M_ext)                       # [R, nx]
            dWc_ext = Wl.t().mm(dM_ext)                    # [hidden, nx]
            g_Wl = dM_ext.mm(Wc_ext.t())
            g_Wc = dWc_ext[:, :nx - 1].contiguous()
            g_bc = dWc_ext[:, nx - 1].contiguous()
            g_bl = dM_ext[:, nx - 1].contiguous()
        d_comp = torch.empty_like(comp)
        _lib.check(L.pv2_ray_rows_backward(_ptr(comp), nv, n_f2, n_geo, R, B, bg[0], bg[1], bg[2], _ptr(lohi),
                                           _ptr(d_xbar), _ptr(g_rgb), _ptr(g_depth), _ptr(d_comp), st),
                   "pv2_ray_rows_backward")
        # fc_p of the semantic head: the reference's fc_p(points) * 0.0 - an exact zero gradient
        g_p = [None, None]
        like = [x for x in ctx.param_like if x is not None]
        if like:
            flat = torch.zeros(sum(n for _, n in like), dtype=torch.float32, device=dev)
            off = 0
            for i, x in enumerate(ctx.param_like):
                if x is not None:
                    g_p[i] = flat[off:off + x[1]].view(x[0])
                    off += x[1]
        return (d_comp, g_sdf, g_grad, g_Wc, g_bc, g_Wl, g_bl, g_p[0], g_p[1], None, None, None, None, None)


def ray_losses(preds, targets, loss_cfg):
    """``SurfaceModel.get_loss`` for a ``RenderOutputs`` of the fused head: LossDict of the terms with a
    positive weight (+ psnr with the colour term), ``.total`` = their sum."""
    global CALLS
    CALLS += 1
    f = preds.fused
    lw = loss_cfg.weights
    wt = tuple(float(lw.get(k, 0.0)) for k in ("depth_loss", "rgb_loss", "free_space_loss", "sdf_loss",
                                                "eikonal_loss"))
    w_sem = float(lw.get("semantic_loss", 0.0))
    md = f["semantic"]
    if md is not None and w_sem > 0:
        sem_p = (md.fc_c[0].weight, md.fc_c[0].bias, md.last_linear.weight, md.last_linear.bias,
                 md.fc_p.weight, md.fc_p.bias)
        sem_gt = targets["semantic"]
    else:
        sem_p, sem_gt, w_sem = (None,) * 6, None, 0.0
    cfg = dict(n_f2=f["n_f2"], n_geo=f["n_geo"], num_scenes=f["num_scenes"], background=f["background"],
               weights=wt, w_sem=w_sem, temperature=float(loss_cfg.get("temperature", 1.0)),
               trunc=float(loss_cfg.sensor_depth_truncation))
    vals = _RayLoss.apply(f["comp"], f["sdf"], f["grad"] if wt[4] > 0 else None, *sem_p, f["starts"],
                          targets["depth"], targets["rgb"], sem_gt, cfg)
    keep = dict(depth_loss=wt[0] > 0, rgb_loss=wt[1] > 0, psnr=wt[1] > 0, semantic_loss=w_sem > 0,
                free_space_loss=wt[2] > 0, sdf_loss=wt[3] > 0, eikonal_loss=wt[4] > 0)
    out = LossDict((k, v) for k, v in zip(OUT_TERMS, vals[:7]) if keep[k])
    out.total = vals[7]
    return out
