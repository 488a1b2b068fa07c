"""CPU stand-ins for the device kernels, built on the oracle restatements (checker only).

``install(monkeypatch)`` swaps the entry points of ponderv2_amd.kernels for oracle-backed
equivalents so that the host-side logic (spconv mirror, models, render head) can be exercised
without a GPU: by tests/ (pytest's monkeypatch) and by bench.py's cpu_baseline leg
(``installed()`` context manager).  The product never imports this file and has no CPU path of
its own.
"""
import contextlib
import numpy as np
import torch

from oracle import rulebook as orb
from oracle.sampler import SmoothSampler as OracleSampler
from oracle.scatter import scatter as oracle_scatter
from oracle.sparse_ops import sparse_conv


class CpuRulebook:
    def __init__(self, K, n_in, n_out, pin, pout, kstart):
        self.K, self.n_in, self.n_out = K, n_in, n_out
        self.pair_in, self.pair_out, self.kstart_host = pin, pout, np.asarray(kstart)

    @property
    def n_pairs(self):
        return int(self.kstart_host[-1])

    def transposed(self):
        return CpuRulebook(self.K, self.n_out, self.n_in, self.pair_out, self.pair_in,
                           self.kstart_host)


def build_subm_rulebook(coords, ksize):
    pin, pout, ks = orb.subm_rulebook(coords.numpy(), ksize)
    n = coords.shape[0]
    return CpuRulebook(ksize ** 3, n, n, torch.from_numpy(pin.astype(np.int64)),
                       torch.from_numpy(pout.astype(np.int64)), ks)


def build_downsample_rulebook(coords, stride, out_shape):
    oc, pin, pout, ks = orb.downsample_rulebook(coords.numpy(), stride, out_shape)
    rb = CpuRulebook(stride ** 3, coords.shape[0], len(oc), torch.from_numpy(pin.astype(np.int64)),
                     torch.from_numpy(pout.astype(np.int64)), ks)
    return rb, torch.from_numpy(oc)


def rulebook_from_table(tbl, K, n_in, n_out, n_rows_dev=None, bounded=False):
    t = tbl.numpy()
    pin, pout, ks = [], [], [0]
    for k in range(K):
        rows = np.nonzero(t[k] >= 0)[0]
        pin.append(rows)
        pout.append(t[k][rows])
        ks.append(ks[-1] + len(rows))
    return CpuRulebook(K, n_in, n_out, torch.from_numpy(np.concatenate(pin).astype(np.int64)),
                       torch.from_numpy(np.concatenate(pout).astype(np.int64)), ks)


def _no_autocast():
    """The device kernels are plain fp32 launches that autocast never touches; the doubles behave
    the same way."""
    return torch.autocast("cpu", enabled=False)


class _ConvIntoApply:
    @staticmethod
    def apply(feats, weight_okc, rb, init):
        with _no_autocast():
            return init + sparse_conv(feats, weight_okc, rb.pair_in, rb.pair_out, rb.kstart_host,
                                      rb.n_out)


class _ConvApply:
    @staticmethod
    def apply(feats, weight_okc, rb):
        with _no_autocast():
            return sparse_conv(feats, weight_okc, rb.pair_in, rb.pair_out, rb.kstart_host, rb.n_out)


class _ScatterApply:
    @staticmethod
    def apply(out, src, index, mean):
        with _no_autocast():
            return oracle_scatter(src, index, dim=0, out=out, reduce="mean" if mean else "sum")


class _HostCompositeWeights(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alphas):
        from oracle import raymarch as orm

        ctx.save_for_backward(alphas)
        w, t = orm.weights_from_alphas(alphas)
        ctx.mark_non_differentiable(t)
        return w, t

    @staticmethod
    def backward(ctx, gw, _gt):
        from oracle import raymarch as orm

        return orm.grad_alpha_closed_form(ctx.saved_tensors[0], gw)


class _HostWeightedSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, values):
        from oracle import raymarch as orm

        ctx.save_for_backward(weights, values)
        return orm.weighted_sum(weights, values)

    @staticmethod
    def backward(ctx, gout):
        from oracle import raymarch as orm

        return orm.weighted_sum_grads_closed_form(*ctx.saved_tensors, gout)


class _HostFieldRender(torch.autograd.Function):
    """Host double of ponderv2_amd.fused_head.field_render: forward from the restatement, backward
    from the HAND-DERIVED formulas the HIP kernels implement (oracle/fused_head.py)."""

    @staticmethod
    def forward(ctx, vol5, origins, dirs, starts, deltas, MW, c0, bc1, W1, b1, A, b_rgb, inv_s,
                norm_pts, norm_div):
        from oracle import fused_head as fh

        args = [t.detach() for t in (vol5, origins, dirs, starts, deltas, MW, c0, bc1, W1, b1, A,
                                     b_rgb, inv_s.reshape(()))]
        ctx.save_for_backward(*args)
        ctx.norm = (bool(norm_pts), float(norm_div) - 1.0 - 10e-4)
        ctx.inv_s_shape = inv_s.shape
        with torch.no_grad():
            out = fh.field_render(*args, norm_pts=ctx.norm[0], norm_padding=ctx.norm[1])
        ctx.mark_non_differentiable(out["weights"])
        return out["sdf"], out["grad"], out["weights"], out["comp"]

    @staticmethod
    def backward(ctx, g_sdf, g_grad, _gw, g_comp):
        from oracle import fused_head as fh

        args = ctx.saved_tensors
        R, S = args[3].shape
        z = lambda g, shape: torch.zeros(shape, dtype=args[0].dtype) if g is None else g
        g = fh.field_render_backward(*args, z(g_sdf, (R, S)), z(g_grad, (R, S, 3)),
                                     z(g_comp, (R, args[0].shape[-1] + 12)),
                                     norm_pts=ctx.norm[0], norm_padding=ctx.norm[1])
        return (g["vol"], None, None, None, None, g["MW"], g["c0"], g["bc1"], g["W1"], g["b1"], g["A"],
                g["b_rgb"], g["inv_s"].reshape(ctx.inv_s_shape), None, None)


def _materialise(x5, wfp):
    """V = Wf X + bf on every cell: what the folded head samples implicitly (wfp = [Wf | bf | 0])."""
    kx = x5.shape[-1]
    return x5 @ wfp[:, :kx].t() + wfp[:, kx]


def _host_field_render_folded(x5, wfp, *head):
    """Host double of ponderv2_amd.fused_head.field_render_folded: by definition the unfolded
    operation on the materialised volume (autograd carries the gradients to x5 and wfp)."""
    return _HostFieldRender.apply(_materialise(x5, wfp), *head)


def _host_coarse_sample(vol5, origins, dirs, nears, fars, lin_bins, t_rand, lin_u, u_rand,
                        n_importance, MW, c0, bc1, W1, b1, base_inv_s, debug=False, wfs=None):
    from oracle import fused_head as fh

    if wfs is not None:   # folded final convolution: the SDF half of the volume, materialised
        vol5 = _materialise(vol5.detach(), wfs.detach())
    with torch.no_grad():
        res = fh.coarse_sample(vol5, origins, dirs, nears, fars, lin_bins.to(vol5.dtype), t_rand,
                               u_rand, n_importance, MW, c0, bc1, W1[0], b1[0], base_inv_s,
                               return_debug=debug)
        bins, dbg = res if debug else (res, None)
        starts, deltas = fh.bins_to_samples(bins, nears, fars)
    return (bins, starts, deltas, dbg) if debug else (bins, starts, deltas)


class _HostNarrowRender(torch.autograd.Function):
    """Host double of ponderv2_amd.narrow_head.field_render: forward from the restatement, backward
    from the HAND-DERIVED formulas csrc/raymarch_narrow.hip implements (oracle/narrow_head.py)."""

    @staticmethod
    def forward(ctx, vol5, theta, inv_s, origins, dirs, starts, deltas, points_factor):
        from oracle import narrow_head as nh
        from ponderv2_amd import narrow_head as prod

        args = [t.detach() for t in (vol5, origins, dirs, starts, deltas, theta, inv_s.reshape(()))]
        ctx.save_for_backward(*args)
        ctx.pf = float(points_factor)
        ctx.hl = (prod.H, prod.L)
        ctx.inv_s_shape = inv_s.shape
        with torch.no_grad():
            out = nh.field_render(*args, *ctx.hl, ctx.pf)
        return out["sdf"], out["grad"], out["weights"], out["comp"]

    @staticmethod
    def backward(ctx, g_sdf, g_grad, g_w, g_comp):
        from oracle import narrow_head as nh

        args = ctx.saved_tensors
        R, S = args[3].shape
        z = lambda g, shape: torch.zeros(shape, dtype=args[0].dtype) if g is None else g
        g = nh.field_render_backward(*args, *ctx.hl, ctx.pf, z(g_sdf, (R, S)), z(g_grad, (R, S, 3)),
                                     z(g_w, (R, S)), z(g_comp, (R, 2)))
        return (g["vol"], g["theta"], g["inv_s"].reshape(ctx.inv_s_shape), None, None, None, None,
                None)


def _host_narrow_coarse_sample(vol5, origins, dirs, nears, fars, lin_bins, t_rand, lin_u, u_rand,
                               n_importance, theta, points_factor, base_inv_s, debug=False):
    from oracle import fused_head as fh, narrow_head as nh
    from ponderv2_amd import narrow_head as prod

    with torch.no_grad():
        res = nh.coarse_sample(vol5.detach(), origins, dirs, nears, fars, lin_bins.to(vol5.dtype),
                               t_rand, u_rand, n_importance, theta.detach().to(vol5.dtype), prod.H,
                               prod.L, float(points_factor), base_inv_s, return_debug=debug)
        bins, dbg = res if debug else (res, None)
        starts, deltas = fh.bins_to_samples(bins, nears, fars)
    return (bins, starts, deltas, dbg) if debug else (bins, starts, deltas)


class _Patcher:
    """Minimal monkeypatch look-alike for use outside pytest."""

    def __init__(self):
        self.undo = []

    def setattr(self, obj, name, value):
        self.undo.append((obj, name, getattr(obj, name)))
        setattr(obj, name, value)

    def restore(self):
        for obj, name, old in reversed(self.undo):
            setattr(obj, name, old)


@contextlib.contextmanager
def installed():
    p = _Patcher()
    install(p)
    try:
        yield
    finally:
        p.restore()


def install(monkeypatch):
    import ponderv2_amd.kernels as K
    import ponderv2_amd.torch_scatter as ts
    from ponderv2_amd.ponder.models.ponder.render_utils.fields import sdf_field

    monkeypatch.setattr(K, "build_subm_rulebook", build_subm_rulebook)
    monkeypatch.setattr(K, "build_downsample_rulebook", build_downsample_rulebook)
    monkeypatch.setattr(K, "SparseConvFunction", _ConvApply)
    monkeypatch.setattr(K, "SparseConvIntoFunction", _ConvIntoApply)
    monkeypatch.setattr(K, "rulebook_from_table", rulebook_from_table)
    monkeypatch.setattr(ts, "ScatterRowsFunction", _ScatterApply)
    monkeypatch.setattr(sdf_field, "SmoothSampler", OracleSampler)
    # compositing kernels (csrc/raymarch.hip): host doubles that use the kernels' closed-form
    # gradients, so the opt-in path (raymarch.ENABLED) can be exercised end to end on the host
    import ponderv2_amd.raymarch as rm

    monkeypatch.setattr(rm, "supported", lambda t, values=None: t.dim() == 3 and t.shape[-1] == 1)
    monkeypatch.setattr(rm, "composite_weights", _HostCompositeWeights.apply)
    monkeypatch.setattr(rm, "weighted_sum", _HostWeightedSum.apply)
    # fused ray march (csrc/raymarch_fused.hip): host doubles from oracle/fused_head.py
    import ponderv2_amd.fused_head as fhead

    monkeypatch.setattr(fhead, "device_ok", lambda t: True)
    monkeypatch.setattr(fhead, "coarse_sample", _host_coarse_sample)
    monkeypatch.setattr(fhead, "field_render", _HostFieldRender.apply)
    monkeypatch.setattr(fhead, "field_render_folded", _host_field_render_folded)
    monkeypatch.setattr(fhead, "fold_supported", fhead.fold_shape_ok)
    # the narrow-decoder head (csrc/raymarch_narrow.hip): host doubles from oracle/narrow_head.py
    import ponderv2_amd.narrow_head as nhead

    monkeypatch.setattr(nhead, "device_ok", lambda t: True)
    monkeypatch.setattr(nhead, "coarse_sample", _host_narrow_coarse_sample)
    monkeypatch.setattr(nhead, "field_render", _HostNarrowRender.apply)
