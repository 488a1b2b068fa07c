"""Dense 3x3x3 convolutions of the projection network on the gfx950 kernels of csrc/dense_conv.hip.

Tensor-level wrappers over ``pv2_dconv3_*`` and the two autograd units the dense U-Net is built
from (ponder/models/ponder/unet3d.py of the reference):

``bn_conv_relu(bn, conv, x)``      SingleConv in "bcr" order (:125-156): training-mode BatchNorm3d ->
                                    Conv3d(k3, p1, no bias) -> ReLU.  The statistics come from the
                                    row kernels of csrc/rownorm.hip; the normalised tensor is never
                                    written - the conv applies ``x * scale + shift`` while it stages
                                    its input tile (zero padding after the map), the ReLU sits in its
                                    epilogue.  Backward: ReLU mask while staging the gradient, the
                                    conv on flipped weights, the deterministic weight gradient, then
                                    the BatchNorm backward on the rows.
``upsample_add(convT, skip, x)``   Decoder joining (:401-411, :474-489): ``skip +
                                    ConvTranspose3d(k3, s2, p1)(x)`` with the bias and the sum in the
                                    epilogue of the transposed conv.

All volumes are float32 ``(B, C, Z, Y, X)`` tensors in channels_last_3d storage.
"""
import os

import torch

from . import _lib, rownorm, sidestream
from .kernels import _ptr, _require_device, _stream, workspace

ENABLED = os.environ.get("PV2_DENSE_CONV", "1") != "0"


def _cl(x):
    """(B, C, Z, Y, X) -> channels-last storage (no copy when it already is)."""
    return x.contiguous(memory_format=torch.channels_last_3d)


def _rows(x):
    """The (cells, C) row matrix of a channels-last volume (a view)."""
    return x.permute(0, 2, 3, 4, 1).reshape(-1, x.shape[1])


def _new_volume(b, c, z, y, x, like):
    return torch.empty((b, z, y, x, c), dtype=torch.float32, device=like.device).permute(0, 4, 1, 2, 3)


def pack_weights(weight, out_dim, flip, mode=0):
    """The (transposed) conv weight ``[d0, d1, 3, 3, 3]`` in MFMA fragment order; ``out_dim`` names the
    dimension that plays the launch's output channels, the other one is reduced over.  ``mode``: the
    ``conv3_forward`` mode the result is for (the kernels' weight formats differ: mode 0 takes three
    bf16 pieces per value, csrc/mfma_split.h)."""
    _require_device(weight)
    assert weight.dim() == 5 and tuple(weight.shape[2:]) == (3, 3, 3) and weight.dtype == torch.float32
    red_dim = 1 - out_dim
    n_out, n_red = weight.shape[out_dim], weight.shape[red_dim]
    L = _lib.lib()
    packed = torch.empty(int(L.pv2_dconv3_packed_floats(n_out, n_red, mode)), dtype=torch.float32,
                         device=weight.device)
    st = weight.stride()
    _lib.check(L.pv2_dconv3_pack_weights(
        _ptr(weight), n_out, n_red, st[out_dim], st[red_dim], st[2], st[3], st[4], int(flip), mode,
        _ptr(packed), _stream(weight)), "pv2_dconv3_pack_weights")
    return packed


def conv3_forward(x, packed, c_out, mode=0, in_scale=None, in_shift=None, mask_src=None, bias=None,
                  addend=None, relu=False, out_mask_src=None):
    """``[relu](conv(mask(x * in_scale + in_shift)) + bias + addend)``; mode 0: k3 s1 p1, 1: transposed
    k3 s2 p1 (2x), 2: strided k3 s2 p1 (1/2)."""
    _require_device(x, packed)
    x = _cl(x)
    b, c_in, z, y, xx = x.shape
    if mode == 0:
        oz, oy, ox = z, y, xx
    elif mode == 1:
        oz, oy, ox = 2 * z, 2 * y, 2 * xx
    else:
        oz, oy, ox = z // 2, y // 2, xx // 2
    out = _new_volume(b, c_out, oz, oy, ox, x)
    if mask_src is not None:
        mask_src = _cl(mask_src)
        assert mask_src.shape == x.shape
    if addend is not None:
        addend = _cl(addend)
        assert addend.shape == out.shape
    _lib.check(_lib.lib().pv2_dconv3_forward(
        _ptr(x), b, z, y, xx, c_in, _ptr(packed), c_out, mode, _ptr(in_scale), _ptr(in_shift),
        _ptr(mask_src), _ptr(bias), _ptr(addend), int(relu),
        _ptr(None if out_mask_src is None else _cl(out_mask_src)), _ptr(out), _stream(x)),
        "pv2_dconv3_forward")
    return out


def conv3_backward_weight(x, gy, weight_like, mode=0, in_scale=None, in_shift=None, mask_src=None,
                          n_dim=0):
    """The weight gradient in the layout (shape, strides) of ``weight_like``; ``n_dim``: which of its
    first two dimensions indexes gy's channels (0 for Conv3d, 1 for ConvTranspose3d)."""
    _require_device(x, gy)
    x, gy = _cl(x), _cl(gy)
    b, c_x, z, y, xx = x.shape
    c_g = gy.shape[1]
    if mask_src is not None:
        mask_src = _cl(mask_src)
    L = _lib.lib()
    floats = int(L.pv2_dconv3_wgrad_partial_floats(b, z, y, xx, c_x, c_g, mode))
    part = workspace("dconv_wgrad", x.device, floats)   # (one per stream: keyed by the current one)
    dw = torch.empty_like(weight_like)
    st = dw.stride()
    _lib.check(L.pv2_dconv3_backward_weight(
        _ptr(x), b, z, y, xx, c_x, _ptr(in_scale), _ptr(in_shift), _ptr(gy), c_g, _ptr(mask_src),
        mode, _ptr(part), _ptr(dw), st[n_dim], st[1 - n_dim], st[2], st[3], st[4], _stream(x)),
        "pv2_dconv3_backward_weight")
    return dw


def conv_supported(conv, x):
    """Conv3d(k3, s1, p1, zero padding, no groups / dilation) on a float32 device volume whose channel
    counts the kernels tile (C_in % 32 == 0 for the weight gradient, C_out % 32 == 0)."""
    return (ENABLED and isinstance(conv, torch.nn.Conv3d) and x.is_cuda and x.dim() == 5
            and x.dtype == torch.float32 and conv.weight.dtype == torch.float32
            and tuple(conv.kernel_size) == (3, 3, 3) and tuple(conv.stride) == (1, 1, 1)
            and tuple(conv.padding) == (1, 1, 1) and tuple(conv.dilation) == (1, 1, 1)
            and conv.groups == 1 and conv.padding_mode == "zeros"
            and conv.in_channels % 32 == 0 and conv.out_channels % 32 == 0)


def upsample_supported(convT, x, output_size):
    return (ENABLED and isinstance(convT, torch.nn.ConvTranspose3d) and x.is_cuda and x.dim() == 5
            and x.dtype == torch.float32 and convT.weight.dtype == torch.float32
            and tuple(convT.kernel_size) == (3, 3, 3) and tuple(convT.stride) == (2, 2, 2)
            and tuple(convT.padding) == (1, 1, 1) and tuple(convT.dilation) == (1, 1, 1)
            and convT.groups == 1 and convT.in_channels % 32 == 0 and convT.out_channels % 32 == 0
            and [int(s) for s in output_size] == [2 * int(s) for s in x.shape[2:]])


def _on_side_stream(fn, leaf, keep):
    """Run ``fn`` (a weight-gradient launch) on the backward side stream when that applies."""
    if sidestream.active(keep[0]) and sidestream.safe_leaf(leaf):
        return sidestream.fork(fn, keep)
    return fn()


class _BnConvRelu(torch.autograd.Function):
    """y = relu(conv(bn(x))): x (B, C_in, Z, Y, X) channels-last, training-mode batch statistics."""

    @staticmethod
    def forward(ctx, x, bn_weight, bn_bias, weight, running_mean, running_var, eps, momentum, relu):
        x = _cl(x)
        b, c_in, z, y, xx = x.shape
        rows = _rows(x)
        n = rows.shape[0]
        L = _lib.lib()
        mean_invstd = torch.empty(2 * c_in, dtype=torch.float32, device=x.device)
        affine = torch.empty(2 * c_in, dtype=torch.float32, device=x.device)
        _lib.check(L.pv2_bn_statistics(
            _ptr(rows), n, c_in, _ptr(bn_weight), _ptr(bn_bias), float(eps), float(momentum),
            _ptr(running_mean), _ptr(running_var), _ptr(rownorm._workspace(x.device, c_in)),
            _ptr(mean_invstd), _ptr(affine), _stream(x)), "pv2_bn_statistics")
        packed = pack_weights(weight, 0, False)
        out = conv3_forward(x, packed, weight.shape[0], 0, in_scale=affine[:c_in],
                            in_shift=affine[c_in:], relu=relu)
        ctx.save_for_backward(x, out if relu else None, mean_invstd, affine, bn_weight, weight)
        ctx.relu = relu
        return out

    @staticmethod
    def backward(ctx, gy):
        x, out, mean_invstd, affine, bn_weight, weight = ctx.saved_tensors
        gy = _cl(gy)
        b, c_in, z, y, xx = x.shape
        gw = gbw = gbb = gx = None
        if ctx.needs_input_grad[3]:
            gw = _on_side_stream(
                lambda: conv3_backward_weight(x, gy, weight, 0, in_scale=affine[:c_in],
                                              in_shift=affine[c_in:], mask_src=out),
                weight, (gy, x, out, affine))
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            packed_t = pack_weights(weight, 1, True)
            gxn = conv3_forward(gy, packed_t, c_in, 0, mask_src=out)   # d loss / d bn(x)
            rows_g, rows_x = _rows(gxn), _rows(x)
            n = rows_x.shape[0]
            gsum = torch.empty(2 * c_in, dtype=torch.float32, device=x.device)
            gx = _new_volume(b, c_in, z, y, xx, x)
            _lib.check(_lib.lib().pv2_bn_backward(
                _ptr(rows_g), _ptr(rows_x), None, _ptr(mean_invstd), _ptr(bn_weight), n, c_in,
                _ptr(rownorm._workspace(x.device, c_in)), _ptr(gsum), _ptr(_rows(gx)), None,
                _stream(x)), "pv2_bn_backward")
            gbb, gbw = gsum[:c_in], gsum[c_in:]
        return gx, gbw, gbb, gw, None, None, None, None, None


def bn_conv_relu(bn, conv, x, relu=True):
    """``relu(conv(bn(x)))`` for a training-mode BatchNorm3d with local statistics."""
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        rownorm._bump_batches_tracked(bn)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    return _BnConvRelu.apply(x, bn.weight, bn.bias, conv.weight, rm, rv, bn.eps, bn.momentum, relu)


def bn_conv_supported(bn, conv, x):
    return (conv_supported(conv, x) and conv.bias is None and isinstance(bn, torch.nn.BatchNorm3d)
            and not isinstance(bn, torch.nn.SyncBatchNorm) and bn.training and bn.affine
            and bn.momentum is not None and bn.weight.dtype == torch.float32
            and x.shape[0] * x.shape[2] * x.shape[3] * x.shape[4] > 1)


class _UpsampleAdd(torch.autograd.Function):
    """out = skip + conv_transpose3d(x, weight, bias; k3 s2 p1, output = 2 x input)."""

    @staticmethod
    def forward(ctx, skip, x, weight, bias):
        x = _cl(x)
        packed = pack_weights(weight, 1, False, mode=1)
        out = conv3_forward(x, packed, weight.shape[1], 1, bias=bias, addend=skip)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = _cl(g)
        gskip = g if ctx.needs_input_grad[0] else None
        gx = gw = gb = None
        if ctx.needs_input_grad[2] or (ctx.has_bias and ctx.needs_input_grad[3]):
            def weight_half():
                g_w = g_b = None
                if ctx.needs_input_grad[2]:
                    g_w = conv3_backward_weight(x, g, weight, 1, n_dim=1)
                if ctx.has_bias and ctx.needs_input_grad[3]:
                    g_b = rownorm.col_sum(_rows(g))
                return g_w, g_b
            gw, gb = _on_side_stream(weight_half, weight, (g, x))
        if ctx.needs_input_grad[1]:
            packed_s = pack_weights(weight, 0, False, mode=2)
            gx = conv3_forward(g, packed_s, weight.shape[0], 2)
        return gskip, gx, gw, gb


def upsample_add(convT, skip, x):
    return _UpsampleAdd.apply(skip, x, convT.weight, convT.bias)
