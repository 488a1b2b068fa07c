"""Dense 3-D projection networks that follow the sparse backbone: ``UNet3D-v1m2`` (indoor) and
``SimpleConv3D-v1m1`` (outdoor).

Mirror of ponder/models/ponder/unet3d.py (SimpleConv3D :16-34, create_conv :45-122, SingleConv
:125-156, Encoder :292-356, Decoder :359-444, Upsampling :447-493, Abstract3DUNet :530-671,
UNet3Dv1m2 :710-743) restricted to what the v1m2 variant instantiates: SingleConv levels in
"bcr" order (BatchNorm3d -> Conv3d(no bias) -> ReLU), MaxPool3d(2) between encoder levels,
ConvTranspose3d(k3, s2, p1, output_size=skip size) + summation joining in the decoder, and a
final 1x1 conv.  On the device in fp32 the 3x3x3 levels and the transposed convolutions run on the
hand-written MFMA kernels of csrc/dense_conv.hip (ponderv2_amd/dense_conv.py: BatchNorm affine, ReLU,
ReLU mask, bias and the skip sum fused into them); anything else (autocast, eval mode, other orders,
host tensors) goes to the convolution library through PyTorch-ROCm; parameter names match the
reference.
"""
import os

import torch
import torch.nn as nn

from ponderv2_amd import dense_conv, dense_unet, sidestream
from ..builder import MODELS


class _SplitBackwardConv(torch.autograd.Function):
    """``nn.Conv3d`` / ``nn.ConvTranspose3d`` (the library convolution, forward unchanged) whose
    backward is issued as its two halves: grad-input on the current stream - the next layer down
    waits for it - and grad-weight (+ grad-bias) on the backward side stream (sidestream.py),
    where it overlaps with the rest of the chain; stock autograd runs the two back to back on one
    stream.  Under autocast the operands are cast here exactly as the autocast dispatch would
    (16-bit input, weight and output; the weight gradient returns to the master weight's type)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation, transposed, output_padding, groups):
        amp = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else None
        w, b = weight, bias
        if amp is not None:
            x, w = x.to(amp), weight.to(amp)
            b = None if bias is None else bias.to(amp)
        ctx.geom = (stride, padding, dilation, transposed, output_padding, groups)
        ctx.bias_sizes = None if bias is None else [bias.shape[0]]
        ctx.leaf = weight
        ctx.save_for_backward(x, w)
        with torch.autocast("cuda", enabled=False):
            return torch.ops.aten.convolution(x, w, b, stride, padding, dilation, transposed,
                                              output_padding, groups)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        leaf, bias_sizes = ctx.leaf, ctx.bias_sizes
        want_b = bias_sizes is not None and ctx.needs_input_grad[2]
        gx = gw = gb = None

        def weight_half():
            _, g_w, g_b = torch.ops.aten.convolution_backward(
                gy, x, w, bias_sizes, *ctx.geom, [False, ctx.needs_input_grad[1], want_b])
            if g_w is not None and g_w.dtype != leaf.dtype:
                g_w = g_w.to(leaf.dtype)
            if g_b is not None and g_b.dtype != leaf.dtype:
                g_b = g_b.to(leaf.dtype)
            return g_w, g_b

        if ctx.needs_input_grad[1] or want_b:
            if sidestream.active(gy) and sidestream.safe_leaf(leaf):
                gw, gb = sidestream.fork(weight_half, (gy, x, w))
            else:
                gw, gb = weight_half()
        if ctx.needs_input_grad[0]:
            gx = torch.ops.aten.convolution_backward(gy, x, w, None, *ctx.geom,
                                                     [True, False, False])[0]
        return gx, gw, gb, None, None, None, None, None, None


class _PointwiseConv(torch.autograd.Function):
    """1x1x1 convolution of a channels-last volume = ONE tall GEMM over its rows (cells x C_in) .
    W^T + bias, on the MFMA kernels of ponderv2_amd.linear (``pv2_gemm_nt`` / ``pv2_gemm_tn``).
    The U-Net's ``final_conv`` maps 32 -> 128 channels on every one of the 128x128x32 cells: a
    bandwidth-bound pass over 134 MB in and 537 MB out.  Through the convolution library it was a
    GEMM in the other layout plus a separate bias pass plus a 537 MB layout conversion on the way
    to the channels-last sampler, and a 0.64 ms skinny BLAS reduction for the weight gradient
    (1.9 ms per step in all, profiles/r02_rocprofv3_kernel_stats_v7_f32.csv); here the result is
    written once, bias included, in the layout the ray march gathers from.  fp32 volumes only."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from ponderv2_amd import _lib
        from ponderv2_amd.kernels import _ptr, _stream

        b, c, z, y, xx = x.shape
        n = weight.shape[0]
        rows = x.permute(0, 2, 3, 4, 1).reshape(-1, c)          # a view of a channels-last volume
        rows = rows.float().contiguous()
        w2 = weight.reshape(n, c).float().contiguous()
        bias32 = None if bias is None else bias.float().contiguous()
        out = torch.empty((rows.shape[0], n), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().pv2_gemm_nt(_ptr(rows), rows.shape[0], c, _ptr(w2), n, _ptr(bias32),
                                          _ptr(out), _stream(rows)), "pv2_gemm_nt")
        ctx.save_for_backward(rows, w2)
        ctx.leaf, ctx.has_bias, ctx.x_dtype, ctx.w_shape = weight, bias is not None, x.dtype, weight.shape
        return out.view(b, z, y, xx, n).permute(0, 4, 1, 2, 3)

    @staticmethod
    def backward(ctx, gy):
        from ponderv2_amd import _lib
        from ponderv2_amd.kernels import _ptr, _stream

        rows, w2 = ctx.saved_tensors
        n, c = w2.shape
        b, _, z, y, xx = gy.shape
        g = gy.permute(0, 2, 3, 4, 1).reshape(-1, n).float().contiguous()
        m = g.shape[0]
        L = _lib.lib()
        gx = gw = gb = None

        def weight_half():
            g_w = g_b = None
            if ctx.needs_input_grad[1]:
                g_w = torch.empty((n, c), dtype=torch.float32, device=g.device)
                _lib.check(L.pv2_zero_fill(_ptr(g_w), g_w.numel() * 4, _stream(g)), "pv2_zero_fill")
                _lib.check(L.pv2_gemm_tn(_ptr(g), _ptr(rows), m, n, c, _ptr(g_w), _stream(g)),
                           "pv2_gemm_tn")
                g_w = g_w.view(ctx.w_shape)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                g_b = torch.empty(n, dtype=torch.float32, device=g.device)
                _lib.check(L.pv2_col_sum(_ptr(g), m, n, _ptr(g_b), _stream(g)), "pv2_col_sum")
            return g_w, g_b

        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            if sidestream.active(g) and sidestream.safe_leaf(ctx.leaf):
                gw, gb = sidestream.fork(weight_half, (g, rows))
            else:
                gw, gb = weight_half()
        if ctx.needs_input_grad[0]:
            wt = w2.t().contiguous()                               # [c, n]: grad-input = g . W
            gx = torch.empty((m, c), dtype=torch.float32, device=g.device)
            _lib.check(L.pv2_gemm_nt(_ptr(g), m, n, _ptr(wt), c, None, _ptr(gx), _stream(g)),
                       "pv2_gemm_nt")
            gx = gx.view(b, z, y, xx, c).permute(0, 4, 1, 2, 3)
            if ctx.x_dtype != torch.float32:
                gx = gx.to(ctx.x_dtype)
        return gx, gw, gb


def pointwise_conv_supported(module, x):
    """``module`` is a 1x1x1 / stride 1 / unpadded ``nn.Conv3d`` and ``x`` a device volume whose
    channel counts the tall GEMM kernels take (both multiples of 8: each is the reduction width of
    one of the two passes)."""
    return (isinstance(module, nn.Conv3d) and x.is_cuda and x.dim() == 5
            and tuple(module.kernel_size) == (1, 1, 1) and tuple(module.stride) == (1, 1, 1)
            and tuple(module.padding) == (0, 0, 0) and tuple(module.dilation) == (1, 1, 1)
            and module.groups == 1 and module.weight.dtype == torch.float32
            and module.in_channels % 8 == 0 and module.out_channels % 8 == 0
            and x.dtype == torch.float32    # (16-bit volumes: the library's 16-bit GEMM moves half
                                             #  the bytes - measured 0.45 ms per step faster there)
            and os.environ.get("PV2_POINTWISE_CONV", "1") != "0")


def library_conv(module, x, output_size=None):
    """``module(x)`` for an ``nn.Conv3d`` / ``nn.ConvTranspose3d`` with its weight gradient routed
    to the backward side stream when that applies (device tensors, training, zero padding)."""
    if output_size is None and pointwise_conv_supported(module, x):
        return _PointwiseConv.apply(x, module.weight, module.bias)
    if not (sidestream.ENABLED and x.is_cuda and torch.is_grad_enabled()
            and module.weight.requires_grad and getattr(module, "padding_mode", "zeros") == "zeros"):
        return module(x) if output_size is None else module(x, output_size)
    transposed = isinstance(module, nn.ConvTranspose3d)
    out_pad = (0, 0, 0)
    if transposed:
        out_pad = tuple(module._output_padding(x, output_size, module.stride, module.padding,
                                               module.kernel_size, 3, module.dilation))
    return _SplitBackwardConv.apply(x, module.weight, module.bias, tuple(module.stride),
                                    tuple(module.padding), tuple(module.dilation), transposed,
                                    out_pad, module.groups)


@MODELS.register_module("SimpleConv3D-v1m1")
class SimpleConv3D(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, padding=1, stride=1):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv3d(in_channels, out_channels, kernel_size=kernel_size, padding=padding,
                      stride=stride),
            nn.BatchNorm3d(out_channels), nn.ReLU(inplace=True))

    def forward(self, x):
        return self.conv(x)

    def forward_cells(self, cells):
        """Same result from the occupied cells of the (mostly empty) input grid: the conv runs on
        the sparse-conv kernels (sparse_input.py), BatchNorm3d + ReLU on its dense 32-channel
        output as usual."""
        from .sparse_input import conv3d_on_cells

        conv, bn, relu = self.conv[0], self.conv[1], self.conv[2]
        assert conv.kernel_size == (3, 3, 3) and conv.stride == (1, 1, 1) and conv.padding == (1, 1, 1)
        return relu(bn(conv3d_on_cells(cells, cells.feat, conv.weight, bias=conv.bias)))


class SingleConv(nn.Sequential):
    """One conv level; ``order`` spells the op sequence: b(atchnorm) g(roupnorm) c(onv) r(elu)
    l(eaky relu) e(lu).  The conv carries a bias only when no norm is present."""

    def __init__(self, in_channels, out_channels, kernel_size=3, order="bcr", num_groups=8,
                 padding=1):
        super().__init__()
        assert "c" in order and order[0] not in "rle"
        self._order = order
        has_norm = "b" in order or "g" in order
        for pos, op in enumerate(order):
            before_conv = pos < order.index("c")
            width = in_channels if before_conv else out_channels
            if op == "c":
                self.add_module("conv", nn.Conv3d(in_channels, out_channels, kernel_size,
                                                  padding=padding, bias=not has_norm))
            elif op == "b":
                self.add_module("batchnorm", nn.BatchNorm3d(width))
            elif op == "g":
                groups = num_groups if width >= num_groups else 1
                assert width % groups == 0
                self.add_module("groupnorm", nn.GroupNorm(groups, width))
            elif op == "r":
                self.add_module("ReLU", nn.ReLU(inplace=True))
            elif op == "l":
                self.add_module("LeakyReLU", nn.LeakyReLU(0.1, inplace=True))
            elif op == "e":
                self.add_module("ELU", nn.ELU(inplace=True))
            else:
                raise ValueError(f"unsupported layer type {op!r} in order {order!r}")

    def forward(self, x):
        # "bcr" on a device volume: BatchNorm statistics on the row kernels, the affine map, the ReLU
        # and (backward) the ReLU mask inside the hand-written 3x3x3 convolution (dense_conv.py)
        if (self._order == "bcr" and dense_conv.bn_conv_supported(self.batchnorm, self.conv, x)
                and torch.is_grad_enabled()):
            return dense_conv.bn_conv_relu(self.batchnorm, self.conv, x)
        for module in self:
            x = library_conv(module, x) if isinstance(module, nn.Conv3d) else module(x)
        return x


# ATen has no channels-last kernel for max_pool3d: it transposes the input to (N, T, C, H, W) and the
# gradient back - at the full-resolution level one 0.50 ms strided copy forward and a 0.20 ms strided
# ReLU backward behind it per step.  The HIP kernels of csrc/dense_pool.hip pool the channels-last
# grid where it lies (one launch per direction, every gradient element written once).
# PV2_CL_MAXPOOL=0 restores nn.MaxPool3d.
CL_MAXPOOL = os.environ.get("PV2_CL_MAXPOOL", "1") != "0"


class _MaxPoolCL(torch.autograd.Function):
    """F.max_pool3d(x, 2) for a float32 channels-last-3d device tensor (B, C, Z, Y, X)."""

    @staticmethod
    def forward(ctx, x):
        from ponderv2_amd import _lib
        from ponderv2_amd.kernels import _ptr, _stream

        b, c, z, y, xx = x.shape
        out = torch.empty((b, z // 2, y // 2, xx // 2, c), dtype=torch.float32, device=x.device)
        idx = torch.empty((b, z // 2, y // 2, xx // 2, c // 4), dtype=torch.int32, device=x.device)
        _lib.check(_lib.lib().pv2_maxpool3d_cl_forward(_ptr(x), b, z, y, xx, c, _ptr(out), _ptr(idx),
                                                       _stream(x)), "pv2_maxpool3d_cl_forward")
        ctx.save_for_backward(idx)
        ctx.shape = (b, c, z, y, xx)
        return out.permute(0, 4, 1, 2, 3)

    @staticmethod
    def backward(ctx, grad_out):
        from ponderv2_amd import _lib
        from ponderv2_amd.kernels import _ptr, _stream

        (idx,) = ctx.saved_tensors
        b, c, z, y, xx = ctx.shape
        g = grad_out.permute(0, 2, 3, 4, 1).contiguous()   # channels-last storage: a no-op view
        gx = torch.empty((b, z, y, xx, c), dtype=torch.float32, device=g.device)
        _lib.check(_lib.lib().pv2_maxpool3d_cl_backward(_ptr(g), _ptr(idx), b, z, y, xx, c, _ptr(gx),
                                                        _stream(g)), "pv2_maxpool3d_cl_backward")
        return gx.permute(0, 4, 1, 2, 3)


def channels_last_max_pool3d(x):
    """``F.max_pool3d(x, 2)`` for a channels-last-3d ``x`` (B, C, Z, Y, X) without leaving the
    layout.  Device float32 tensors with C % 4 == 0 run the kernels of csrc/dense_pool.hip (same
    values, same choice among tied maxima as max_pool3d_with_indices: the first in window order);
    anything else takes a layout-aware composite of stock ops - a 2x2 ``max_pool2d`` over (Y, X) on
    the ``(B*Z, C, Y, X)`` channels-last-2d view of the same memory, then the maximum over pairs of
    z-slices (same values; ties may resolve to another member of the window)."""
    b, c, z, y, xx = x.shape
    if (x.is_cuda and x.dtype == torch.float32 and c % 4 == 0 and min(z, y, xx) >= 2
            and x.is_contiguous(memory_format=torch.channels_last_3d)):
        return _MaxPoolCL.apply(x)
    z2 = z // 2
    rows = x.permute(0, 2, 3, 4, 1)[:, :2 * z2]                       # (B, 2*z2, Y, X, C), a view
    planes = rows.reshape(b * 2 * z2, y, xx, c).permute(0, 3, 1, 2)    # (B*Z, C, Y, X) channels-last
    pooled = torch.nn.functional.max_pool2d(planes, 2)                 # (B*Z, C, Y/2, X/2), NHWC
    y2, x2 = pooled.shape[2], pooled.shape[3]
    pairs = pooled.permute(0, 2, 3, 1).reshape(b, z2, 2, y2, x2, c)
    return pairs.max(dim=2).values.permute(0, 4, 1, 2, 3)              # (B, C, Z/2, Y/2, X/2), channels-last


class Encoder(nn.Module):
    def __init__(self, in_channels, out_channels, apply_pooling=True, order="bcr", num_groups=8):
        super().__init__()
        self.pooling = nn.MaxPool3d(kernel_size=(2, 2, 2)) if apply_pooling else None
        self.basic_module = SingleConv(in_channels, out_channels, order=order,
                                       num_groups=num_groups)

    def forward(self, x):
        if self.pooling is not None:
            if (CL_MAXPOOL and x.dim() == 5 and x.is_contiguous(memory_format=torch.channels_last_3d)
                    and not x.is_contiguous()):
                x = channels_last_max_pool3d(x)
            else:
                x = self.pooling(x)
        return self.basic_module(x)


class Upsampling(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, scale_factor=(2, 2, 2)):
        super().__init__()
        self.upsample = nn.ConvTranspose3d(in_channels, out_channels, kernel_size=kernel_size,
                                           stride=scale_factor, padding=1)

    def forward(self, encoder_features, x):
        return library_conv(self.upsample, x, list(encoder_features.size()[2:]))


class Decoder(nn.Module):
    """Learned 2x upsampling, sum with the skip features, one conv level."""

    def __init__(self, in_channels, out_channels, order="bcr", num_groups=8):
        super().__init__()
        self.upsampling = Upsampling(in_channels, out_channels)
        self.basic_module = SingleConv(out_channels, out_channels, order=order,
                                       num_groups=num_groups)

    def forward(self, encoder_features, x):
        up = self.upsampling.upsample
        if (dense_conv.upsample_supported(up, x, encoder_features.shape[2:])
                and encoder_features.dtype == torch.float32 and torch.is_grad_enabled()):
            # skip + ConvTranspose3d(x) (+ bias) in one launch: the sum is the conv's epilogue
            return self.basic_module(dense_conv.upsample_add(up, encoder_features, x))
        return self.basic_module(encoder_features + self.upsampling(encoder_features, x))


@MODELS.register_module("UNet3D-v1m2")
class UNet3Dv1m2(nn.Module):
    def __init__(self, in_channels, out_channels, final_sigmoid=False, f_maps=32,
                 layer_order="bcr", num_groups=1, num_levels=4, is_segmentation=False,
                 testing=False, **kwargs):
        super().__init__()
        if isinstance(f_maps, int):
            f_maps = [f_maps * 2 ** k for k in range(num_levels)]
        self.testing = testing
        widths = [in_channels] + list(f_maps)
        self.encoders = nn.ModuleList(
            Encoder(widths[i], widths[i + 1], apply_pooling=i > 0, order=layer_order,
                    num_groups=num_groups) for i in range(len(f_maps)))
        rev = list(reversed(f_maps))
        self.decoders = nn.ModuleList(
            Decoder(rev[i], rev[i + 1], order=layer_order, num_groups=num_groups)
            for i in range(len(rev) - 1))
        self.final_conv = nn.Conv3d(f_maps[0], out_channels, 1)
        self.final_activation = None
        if is_segmentation:
            self.final_activation = nn.Sigmoid() if final_sigmoid else nn.Softmax(dim=1)

    def cells_supported(self):
        """The sparse first level needs "bcr" (BatchNorm3d -> Conv3d 3x3x3 -> ReLU) without pooling,
        and a BatchNorm whose statistics are local to this rank (not SyncBatchNorm)."""
        first = self.encoders[0]
        names = [n for n, _ in first.basic_module.named_children()]
        return (first.pooling is None and names == ["batchnorm", "conv", "ReLU"]
                and not isinstance(first.basic_module.batchnorm, nn.SyncBatchNorm)
                and tuple(first.basic_module.conv.kernel_size) == (3, 3, 3))

    def forward_cells(self, cells, fold_final=False):
        """Forward from the occupied cells of the input grid: level 0 ("bcr": BatchNorm3d -> conv
        -> ReLU on the 96-channel grid, half of this network's FLOPs) is computed sparsely
        (sparse_input.py), the rest of the U-Net runs on its dense output."""
        from .sparse_input import bn_conv_relu_on_cells

        if not self.cells_supported():
            raise NotImplementedError("forward_cells needs a local BatchNorm3d -> Conv3d(3) -> ReLU first level")
        first = self.encoders[0].basic_module
        with torch.autocast("cuda", enabled=False):  # the sparse kernels are fp32
            x = bn_conv_relu_on_cells(first.batchnorm, first.conv, cells)
        return self.forward(None, first=x, fold_final=fold_final)

    def forward(self, x, first=None, fold_final=False):
        """``fold_final``: stop in front of ``final_conv`` when the fused render head can apply it
        per sample (fused_head.FoldedVolume) - the 128-channel volume is then never built."""
        x0 = first if first is not None else self.encoders[0](x)
        if dense_unet.supported(self, x0):
            # every level behind the first as ONE autograd node over the dense kernels (dense_unet.py)
            from ponderv2_amd import cells_level
            x = dense_unet.forward(self, x0, premask_input=cells_level.claim_premasked(x0))
        else:
            skips, x = [x0], x0
            for encoder in self.encoders[1:]:
                x = encoder(x)
                skips.insert(0, x)
            for decoder, skip in zip(self.decoders, skips[1:]):
                x = decoder(skip, x)
        if fold_final and not (self.testing and self.final_activation is not None):
            from ponderv2_amd import fused_head
            if fused_head.fold_supported(self.final_conv, x):
                return fused_head.FoldedVolume(x, self.final_conv)
        x = library_conv(self.final_conv, x)
        if self.testing and self.final_activation is not None:
            x = self.final_activation(x)
        return x
