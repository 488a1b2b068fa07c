"""The first dense convolution of the projection network, evaluated from the occupied cells only.

``to_dense`` (ponder_indoor_base.py:177-342, ponder_outdoor_base.py:176-209) scatters the backbone
features into a dense grid that is >= 90 % empty (ScanNet: ~30 k occupied of 524 288 cells per
scene), and the projection network's first 3x3x3 convolution - 52 % of the dense U-Net's FLOPs at
96 input channels - then multiplies mostly zeros.  A dense conv IS a sparse conv with a full
rulebook, so the same layer is computed here with the sparse-conv kernels on a rulebook that only
lists (occupied cell, tap) pairs; the dense 96-channel grid is never materialised.

  out[p] = b + sum_k W_k . in[p + k - 1],      in[q] = y0 + [q occupied] * delta_q

  * ``delta`` lives on the Nc occupied cells; ``y0`` is the value of every EMPTY cell (zero for a
    plain conv; ``beta - mean * invstd * gamma`` when a BatchNorm3d precedes the conv, as in
    UNet3D's "bcr" levels - then ``delta = x * invstd * gamma`` because BN is affine per channel);
  * the constant part ``sum_k W_k . y0`` only depends on which taps fall inside the grid, i.e. on
    one of 27 border classes: it is expanded from a (3,3,3,Cout) table and initialises the output;
  * the occupied part is accumulated on top by ``spconv_forward`` (out rows = dense positions, so
    the result IS the channels-last dense tensor); backward reuses the sparse grad-input /
    grad-weight kernels, the constant part differentiates through plain einsums.
BatchNorm3d statistics over ALL cells follow from the occupied cells' sums (the rest are zeros).
"""
from dataclasses import dataclass, field
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from ponderv2_amd import kernels as K
from ponderv2_amd.torch_scatter import scatter


@dataclass
class DenseCells:
    """Occupied cells of a (batch, Z, Y, X) grid in ascending row order of its channels-last
    layout; ``feat`` carries gradients back to the sparse backbone."""

    feat: torch.Tensor          # (Nc, C)
    lin: torch.Tensor           # (Nc,) int64 row = ((b*Z + z)*Y + y)*X + x
    batch: int
    dims: Tuple[int, int, int]  # (Z, Y, X)
    _rulebook: Optional[object] = field(default=None, repr=False)

    @property
    def n_rows(self):
        z, y, x = self.dims
        return self.batch * z * y * x

    def rulebook(self):
        if self._rulebook is None:
            # (sized by upper bounds on device tensors, padding rows skipped: no host read)
            self._rulebook = K.rulebook_from_table(_tap_table(self), 27, self.lin.shape[0],
                                                   self.n_rows, bounded=self.lin.is_cuda)
        return self._rulebook


def cells_geometry(lin, batch, dims, build_rulebook=False):
    """The half of ``cells_from_voxels`` that only needs the voxels' dense rows: a flag grid and its
    prefix sum number the occupied cells.  No device->host read: the cell arrays are sized for the
    worst case (one cell per voxel); rows past the true cell count are padding (``lin`` -1), which
    every consumer skips.  ``build_rulebook``: also the (cell, tap) pair lists of the first
    convolution - all of it a function of the batch's coordinates, so a model's input-pipeline hook
    can run it a step ahead (PonderIndoor.prefetch)."""
    z, y, x = dims
    total = batch * z * y * x
    cap = lin.shape[0]
    flags = torch.zeros(total, dtype=torch.int32, device=lin.device)
    flags.index_fill_(0, lin, 1)
    cell_id = torch.cumsum(flags, 0, dtype=torch.int32)
    cell_of_voxel = cell_id[lin].long() - 1
    cell_lin = torch.full((cap,), -1, dtype=torch.int64, device=lin.device)
    cell_lin[cell_of_voxel] = lin  # duplicates write the same value
    geo = dict(cell_of_voxel=cell_of_voxel, cell_lin=cell_lin, batch=batch, dims=(z, y, x), rulebook=None)
    if build_rulebook:
        geo["rulebook"] = DenseCells(None, cell_lin, batch, (z, y, x)).rulebook()
    return geo


def cells_from_voxels(feat, lin, batch, dims, reduce="mean", geometry=None):
    """Pool per-voxel rows into their dense cells WITHOUT building the dense grid (``cells_geometry``
    - or its result from a step ahead - then one scatter-mean of the features; padding cells keep
    zero features)."""
    geo = geometry if geometry is not None else cells_geometry(lin, batch, dims)
    cap = geo["cell_lin"].shape[0]
    pooled = scatter(feat, geo["cell_of_voxel"][:, None], dim=0, reduce=reduce,
                     out=feat.new_zeros((cap, feat.shape[1])))
    return DenseCells(pooled, geo["cell_lin"], batch, tuple(dims), _rulebook=geo.get("rulebook"))


def _tap_table(cells):
    """int32 [27, Nc]: output row that cell i feeds through tap k = (kz*3 + ky)*3 + kx, or -1.
    Cross-correlation: out[p] += W_k . in[p + k - 1], so cell q reaches p = q - (k - 1)."""
    zs, ys, xs = cells.dims
    lin = cells.lin
    if lin.is_cuda and lin.dtype == torch.int64 and cells.n_rows < 2 ** 31:
        from ponderv2_amd import cells_level

        if cells_level.ENABLED:       # one launch (csrc/cells_level.hip) instead of ~35
            return cells_level.tap_table(cells)
    x = lin % xs
    y = torch.div(lin, xs, rounding_mode="floor") % ys
    z = torch.div(lin, xs * ys, rounding_mode="floor") % zs
    base = lin - ((z * ys + y) * xs + x)
    k = torch.arange(27, device=lin.device)[:, None]
    pz = z[None] - (torch.div(k, 9, rounding_mode="floor") - 1)
    py = y[None] - (torch.div(k, 3, rounding_mode="floor") % 3 - 1)
    px = x[None] - (k % 3 - 1)
    ok = (pz >= 0) & (pz < zs) & (py >= 0) & (py < ys) & (px >= 0) & (px < xs) & (lin >= 0)[None]
    row = base[None] + (pz * ys + py) * xs + px
    return torch.where(ok, row, torch.full_like(row, -1)).to(torch.int32).contiguous()


def _inside(size, device, dtype):
    """(3, size): 1 where tap k of a size-3 kernel reads inside [0, size) at position p."""
    q = torch.arange(size, device=device)[None] + torch.arange(3, device=device)[:, None] - 1
    return ((q >= 0) & (q < size)).to(dtype)


_CLASS_CACHE = {}


def _axis_classes(size):
    """Positions of an axis fall into classes by WHICH of the three taps of a size-3 kernel read
    inside [0, size): first / interior / last (plus "first and last" for size 1).  Returns
    (M (3, n_classes) float64 0/1 patterns, idx (size,) int64 class of every position)."""
    q = torch.arange(size)[None] + torch.arange(3)[:, None] - 1
    inside = ((q >= 0) & (q < size)).to(torch.float64)
    pats, idx = torch.unique(inside, dim=1, return_inverse=True)
    return pats, idx


def _border_classes(batch, dims, device, dtype):
    """Cached per grid: the class patterns of the three axes, the class id of every row of the
    (batch, Z, Y, X) grid, and the one-hot matrices the backward reduces with."""
    key = (batch, tuple(dims), str(device), dtype)
    hit = _CLASS_CACHE.get(key)
    if hit is None:
        (mz, iz), (my, iy), (mx, ix) = (_axis_classes(n) for n in dims)
        ny, nx = my.shape[1], mx.shape[1]
        rows = ((iz[:, None, None] * ny + iy[None, :, None]) * nx + ix[None, None, :]).reshape(-1)
        onehot = [torch.nn.functional.one_hot(i, m.shape[1]).t().to(dtype).to(device).contiguous()
                  for i, m in ((iz, mz), (iy, my), (ix, mx))]
        hit = _CLASS_CACHE[key] = dict(
            m=[m.to(dtype).to(device) for m in (mz, my, mx)], onehot=onehot,
            rows=rows.repeat(batch).to(device))
        if len(_CLASS_CACHE) > 8:
            _CLASS_CACHE.pop(next(iter(_CLASS_CACHE)))
    return hit


class _ExpandClasses(torch.autograd.Function):
    """table (nz, ny, nx, C) -> (batch * Z * Y * X, C): every grid row gets the table row of its
    border class.  One row-gather kernel forward (a pure write of the output); the backward sums
    the incoming gradient per class with three small GEMMs against one-hot matrices (one pass over
    the gradient) - instead of an index_add with a million rows landing on 27 addresses."""

    @staticmethod
    def forward(ctx, table, cls, batch, dims):
        ctx.cls, ctx.batch, ctx.dims = cls, batch, dims
        ctx.table_shape = table.shape
        flat = table.reshape(-1, table.shape[-1])
        if flat.is_cuda and flat.dtype == torch.float32 and flat.shape[1] % 4 == 0:
            from ponderv2_amd import _lib

            if "rows32" not in cls:
                cls["rows32"] = cls["rows"].to(torch.int32).contiguous()
            flat = flat.contiguous()
            out = torch.empty((cls["rows32"].numel(), flat.shape[1]), dtype=torch.float32, device=flat.device)
            _lib.check(_lib.lib().pv2_gather_rows(K._ptr(flat), K._ptr(cls["rows32"]), out.shape[0],
                                                  flat.shape[1], K._ptr(out), K._stream(flat)),
                       "pv2_gather_rows")
            return out
        return flat.index_select(0, cls["rows"])

    @staticmethod
    def backward(ctx, g):
        zs, ys, xs = ctx.dims
        oz, oy, ox = ctx.cls["onehot"]                   # (n_classes, size) each
        c = g.shape[-1]
        g = g.to(oz.dtype).reshape(ctx.batch * zs, ys * xs * c)   # (an autocast region may hand over 16 bits)
        gz = oz.repeat(1, ctx.batch).mm(g)               # sum over batch and z-class: (nz, Y*X*C)
        gy = torch.einsum("ayxo,by->abxo", gz.view(-1, ys, xs, c), oy)
        gt = torch.einsum("abxo,cx->abco", gy, ox)
        return gt.reshape(ctx.table_shape), None, None, None


def _constant_part(weight, y0, bias, batch, dims):
    """(batch*Z*Y*X, Cout): response of the zero-padded conv to the constant field y0 (+ bias).
    It only depends on which taps fall inside the grid, i.e. on the border class of the position
    per axis (first / interior / last): a (3,3,3,Cout) table of class responses, expanded by one
    row gather - building it through full-size einsums and ``repeat`` cost 0.9 ms per step at the
    ScanNet grid (a 67 MB intermediate, a non-vectorised broadcast copy to 134 MB, and their
    backward)."""
    zs, ys, xs = dims
    c_out = weight.shape[0]
    if y0 is None:
        if bias is None:
            return K.zeros_by_kernel((batch * zs * ys * xs, c_out), weight.dtype, weight.device) \
                if weight.is_cuda else weight.new_zeros((batch * zs * ys * xs, c_out))
        return bias.expand(batch * zs * ys * xs, c_out).clone(memory_format=torch.contiguous_format)
    cls = _border_classes(batch, dims, weight.device, weight.dtype)
    mz, my, mx = cls["m"]
    u = torch.einsum("ocijk,c->ijko", weight, y0)
    table = torch.einsum("ijko,ia,jb,kc->abco", u, mz, my, mx)
    if bias is not None:
        table = table + bias
    # (a fresh, non-view buffer: the sparse part is accumulated into it in place)
    return _ExpandClasses.apply(table, cls, batch, (zs, ys, xs))


def conv3d_on_cells(cells, delta, weight, y0=None, bias=None, relu=False):
    """3x3x3 / stride 1 / padding 1 convolution of the field ``y0 + occupied * delta``.
    weight (Cout, Cin, 3, 3, 3) as nn.Conv3d holds it -> (B, Cout, Z, Y, X), channels-last."""
    assert tuple(weight.shape[2:]) == (3, 3, 3)
    c_out, c_in = weight.shape[:2]
    init = _constant_part(weight, y0, bias, cells.batch, cells.dims)
    w_okc = weight.permute(0, 2, 3, 4, 1).reshape(c_out, 27, c_in).contiguous()
    out = K.SparseConvIntoFunction.apply(delta, w_okc, cells.rulebook(), init)
    if relu:   # on the rows, BEFORE the volume view: an in-place op on a view makes autograd clone the
        out = F.relu_(out)   # whole 134 MB gradient (CopySlices) and run the ReLU backward strided
    zs, ys, xs = cells.dims
    return out.view(cells.batch, zs, ys, xs, c_out).permute(0, 4, 1, 2, 3)


def bn_conv_relu_on_cells(bn, conv, cells):
    """relu(conv(batchnorm3d(dense))) - one "bcr" level of UNet3D (unet3d.py SingleConv) - from the
    occupied cells.  Training-mode statistics run over all batch*Z*Y*X positions."""
    from ponderv2_amd import cells_level

    if cells_level.supported(bn, conv, cells):   # the same level as one autograd node on HIP kernels
        return cells_level.bn_conv_relu(bn, conv, cells)
    x = cells.feat
    n_tot = float(cells.n_rows)
    if bn.training or not bn.track_running_stats:
        mean = x.sum(0) / n_tot
        var = ((x * x).sum(0) / n_tot - mean * mean).clamp_min(0.0)
        if bn.training and bn.track_running_stats:
            with torch.no_grad():
                if bn.momentum is not None:   # counted on the host, as every fused BatchNorm
                    from ponderv2_amd.rownorm import _bump_batches_tracked

                    _bump_batches_tracked(bn)
                    m = bn.momentum
                else:
                    bn.num_batches_tracked.add_(1)
                    m = 1.0 / float(bn.num_batches_tracked)
                bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
                bn.running_var.mul_(1 - m).add_(var * (n_tot / max(n_tot - 1.0, 1.0)), alpha=m)
    else:
        mean, var = bn.running_mean, bn.running_var
    scale = torch.rsqrt(var + bn.eps)
    if bn.affine:
        scale = scale * bn.weight
    y0 = -mean * scale
    if bn.affine:
        y0 = y0 + bn.bias
    return conv3d_on_cells(cells, x * scale, conv.weight, y0=y0, bias=conv.bias, relu=True)
