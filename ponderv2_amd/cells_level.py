"""UNet3D's first level (BatchNorm3d -> Conv3d 3x3x3 -> ReLU) from the occupied cells of the input
grid, as ONE autograd node over hand-written kernels (csrc/cells_level.hip, csrc/rownorm.hip,
csrc/sparse_conv.hip).

Same arithmetic as ``ponder/models/ponder/sparse_input.bn_conv_relu_on_cells`` (which stays the CPU
path and the fallback for shapes the kernels do not cover; tests/test_gpu_kernels.py compares both
with the dense float64 layer): the composite ran ~110 small launches forward and ~55 backward, among
them a 0.33 ms strided copy inside an einsum backward and a full-size ReLU backward pass; the node
launches 6 + the sparse convolution's own, forward, and 9 backward.  Reference: the dense grid of
ponder_indoor_base.py:177-342 (to_dense) through unet3d.py:292-318 (Encoder / SingleConv "bcr").

The node's output is a ReLU output.  A consumer that masks the gradient it returns with
``output > 0`` itself (dense_unet.py does, in the epilogue of its last kernel) says so through
``claim_premasked``; the node then takes the incoming gradient as it is instead of masking it in a
pass of its own over the full-size grid."""
import os

import torch

from . import _lib, rownorm, sidestream
from . import kernels as K
from .kernels import _ptr, _stream, workspace

ENABLED = os.environ.get("PV2_CELLS_NODE", "1") != "0"


class _Handle:
    """Shared between the node and whoever consumes its output (see ``claim_premasked``)."""

    __slots__ = ("premasked",)

    def __init__(self):
        self.premasked = False


def supported(bn, conv, cells):
    x = cells.feat
    c_in, c_out = conv.in_channels, conv.out_channels
    return (ENABLED and x.is_cuda and x.dtype == torch.float32 and cells.lin.is_cuda
            and bn.training and bn.affine and bn.momentum is not None
            and type(bn) in (torch.nn.BatchNorm3d, torch.nn.BatchNorm1d)
            and tuple(conv.kernel_size) == (3, 3, 3) and tuple(conv.stride) == (1, 1, 1)
            and tuple(conv.padding) == (1, 1, 1) and tuple(conv.dilation) == (1, 1, 1)
            and conv.groups == 1 and conv.padding_mode == "zeros"
            and conv.weight.dtype == torch.float32
            and c_in % 4 == 0 and c_in <= 1024 and c_out % 4 == 0 and c_out <= 256
            and 256 % (c_out // 4) == 0 and x.shape[0] > 0
            and cells.n_rows * max(c_out // 4, 27) < 2 ** 31)


def tap_table(cells):
    """int32 [27, cap] (sparse_input._tap_table in one launch)."""
    cap = cells.lin.shape[0]
    z, y, x = cells.dims
    tbl = torch.empty((27, cap), dtype=torch.int32, device=cells.lin.device)
    _lib.check(_lib.lib().pv2_cells_tap_table(_ptr(cells.lin), cap, z, y, x, _ptr(tbl), _stream(cells.lin)),
               "pv2_cells_tap_table")
    return tbl


class _CellsLevel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bn_w, bn_b, w, bias, cells, bn, handle):
        L = _lib.lib()
        x = x.contiguous()
        st = _stream(x)
        dev = x.device
        cap, c_in = x.shape
        c_out = w.shape[0]
        z, y, xx = cells.dims
        b, n_tot = cells.batch, cells.n_rows
        stats = torch.empty(4 * c_in, dtype=torch.float32, device=dev)   # mean, invstd | scale, y0
        base = stats.data_ptr()
        rm = bn.running_mean.data_ptr() if bn.track_running_stats else None
        rv = bn.running_var.data_ptr() if bn.track_running_stats else None
        _lib.check(L.pv2_bn_statistics_padded(
            x.data_ptr(), cap, n_tot - cap, c_in, bn_w.data_ptr(), bn_b.data_ptr(), float(bn.eps),
            float(bn.momentum), rm, rv, _ptr(rownorm._workspace(dev, c_in)), base, base + 8 * c_in, st),
            "pv2_bn_statistics_padded")
        if bn.track_running_stats and bn.num_batches_tracked is not None:
            rownorm._bump_batches_tracked(bn)
        packs = torch.empty((2, c_out, 27, c_in), dtype=torch.float32, device=dev)
        u = torch.empty((27, c_out), dtype=torch.float32, device=dev)
        sw = w.stride()
        _lib.check(L.pv2_cells_fold_weights(w.data_ptr(), sw[0], sw[1], sw[2], sw[3], sw[4], c_out, c_in,
                                            base + 8 * c_in, packs[0].data_ptr(), packs[1].data_ptr(),
                                            u.data_ptr(), st), "pv2_cells_fold_weights")
        out = torch.empty((n_tot, c_out), dtype=torch.float32, device=dev)
        _lib.check(L.pv2_cells_expand(u.data_ptr(), None if bias is None else bias.data_ptr(), b, z, y, xx,
                                      c_out, out.data_ptr(), st), "pv2_cells_expand")
        K.spconv_forward(x, packs[1], cells.rulebook(), out=out)
        torch.relu_(out)
        ctx.cells, ctx.handle, ctx.has_bias = cells, handle, bias is not None
        ctx.save_for_backward(x, stats, packs, bn_w, w, out)
        return out

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        x, stats, packs, bn_w, w, out = ctx.saved_tensors
        cells = ctx.cells
        g = g.contiguous()
        if not ctx.handle.premasked:
            g = torch.where(out > 0, g, torch.zeros_like(g))
        st = _stream(g)
        dev = g.device
        cap, c_in = x.shape
        c_out = w.shape[0]
        z, y, xx = cells.dims
        b, n_tot = cells.batch, cells.n_rows
        rb = cells.rulebook()
        base = stats.data_ptr()
        sw = w.stride()
        gu = torch.empty((27, c_out), dtype=torch.float32, device=dev)
        gy0 = torch.empty((27, c_in), dtype=torch.float32, device=dev)
        ws = workspace("cells_bwd", dev, int(L.pv2_cells_backward_workspace_floats(b, z, y, c_out)))
        _lib.check(L.pv2_cells_backward_table(g.data_ptr(), b, z, y, xx, c_out, w.data_ptr(), sw[0], sw[1],
                                              sw[2], sw[3], sw[4], c_in, _ptr(ws), gu.data_ptr(),
                                              gy0.data_ptr(), st), "pv2_cells_backward_table")
        dx = dgamma = dbeta = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            gd = K.spconv_grad_input(g, packs[0], rb)          # d / d(normalised cell rows)
            gsum = torch.empty(2 * c_in, dtype=torch.float32, device=dev)
            dx = torch.empty_like(x)
            _lib.check(L.pv2_bn_backward_padded(gd.data_ptr(), x.data_ptr(), cap, n_tot - cap, c_in, base,
                                                bn_w.data_ptr(), gy0.data_ptr(), 27,
                                                _ptr(rownorm._workspace(dev, c_in)), gsum.data_ptr(),
                                                dx.data_ptr(), st), "pv2_bn_backward_padded")
            dbeta, dgamma = gsum[:c_in], gsum[c_in:]
        dbias = g.sum(0) if ctx.has_bias else None

        def weight_gradient():
            dws = K.spconv_backward_weight(x, g, rb, c_out)       # for W * scale, [c_out, 27, c_in]
            dw = torch.empty_strided(w.shape, w.stride(), dtype=torch.float32, device=dev)
            _lib.check(L.pv2_cells_dw_finish(dws.data_ptr(), gu.data_ptr(), base + 8 * c_in, c_out, c_in,
                                             dw.data_ptr(), sw[0], sw[1], sw[2], sw[3], sw[4], _stream(g)),
                       "pv2_cells_dw_finish")
            return dw

        dw = None
        if ctx.needs_input_grad[3]:
            if sidestream.active(g) and sidestream.safe_leaf(w):
                # (what the side stream reads must outlive this node: operands AND the rulebook's pair lists /
                # tile prefixes - they are released with ctx as soon as backward() returns, and the
                # allocator would hand their memory to the next request on the main stream)
                dw = sidestream.fork(weight_gradient, (x, g, gu, stats, w, packs, cells, rb))
            else:
                dw = weight_gradient()
        return dx, dgamma, dbeta, dw, dbias, None, None, None


def bn_conv_relu(bn, conv, cells):
    """relu(conv(batchnorm3d(dense grid))) from the occupied cells -> (B, C_out, Z, Y, X) channels-last
    volume (a view of the node's (rows, C_out) output).  Training-mode statistics over all cells."""
    handle = _Handle()
    rows = _CellsLevel.apply(cells.feat, bn.weight, bn.bias, conv.weight, conv.bias, cells, bn, handle)
    z, y, x = cells.dims
    vol = rows.view(cells.batch, z, y, x, conv.out_channels).permute(0, 4, 1, 2, 3)
    vol._pv2_relu_handle = handle
    return vol


def claim_premasked(volume):
    """Called by the ONE consumer of ``volume`` that returns its gradient already multiplied by
    ``volume > 0``: True when ``volume`` is this node's output (which then skips its own mask)."""
    handle = getattr(volume, "_pv2_relu_handle", None)
    if handle is None:
        return False
    handle.premasked = True
    return True
