"""conv -> BatchNorm1d -> (+ shortcut) -> ReLU of the sparse backbone as ONE native call per direction.

The reference runs these as separate modules (``BasicBlock.forward``,
ponder/models/sparse_unet/spconv_unet_v1m1_base.py:70-83; the ``SparseConv3d / SubMConv3d +
norm_fn + ReLU`` stages of ``SparseSequential`` :108,120-121,143-145) - ~10 kernel launches and as
many autograd nodes per unit and direction.  Here a unit is one autograd node over
``pv2_convbn_forward`` / ``pv2_convbn_backward`` (csrc/sparse_conv_pr.hip):

    forward   products -> row reduce (+ BatchNorm partial statistics) -> combine -> apply(+res, ReLU)
    backward  BatchNorm backward (3 launches) -> weight gradient (2, on the side stream)
              -> grad-input products -> row reduce

No atomics and no zero-fills anywhere: every result is bitwise reproducible.  The unit uses the
parameters and buffers of the stock modules (state_dict compatible); anything the kernels do not
cover (eval mode, 16-bit features, the 6-channel stem, host tensors) takes the modular path.
"""
import ctypes

import torch

from . import _lib, kernels as K, precision, rownorm, sidestream
from .kernels import _ptr, _stream


def supported(feats, weight_okc, rb, bn) -> bool:
    """True when the fused unit covers this call."""
    if not (feats.is_cuda and K.USE_CONVBN):   # (host tensors: the test doubles' rulebooks)
        return False
    c_out, k, c_in = weight_okc.shape
    return (K._use_pr(rb, c_in, c_out) and c_out % 32 == 0
            and feats.dtype == torch.float32 and weight_okc.dtype == torch.float32
            and rb.n_out > 1 and rb.n_in > 0 and precision.sparse_dtype() is None
            and rownorm.can_fuse(bn, feats) and feats.dim() == 2)


class ConvBNFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, weight_okc, bn_weight, bn_bias, residual, rb, running_mean, running_var,
                relu, eps, momentum):
        feats = feats.contiguous()
        weight_okc = weight_okc.contiguous()
        c_out, _, c_in = weight_okc.shape
        dev = feats.device
        n_out = rb.n_out
        if residual is not None:
            residual = residual.contiguous()
            assert residual.shape == (n_out, c_out) and residual.dtype == torch.float32
        g = rb.geom(c_in, c_out)
        y = torch.empty((n_out, c_out), dtype=torch.float32, device=dev)
        out = torch.empty((n_out, c_out), dtype=torch.float32, device=dev)
        mean_invstd = torch.empty(2 * c_out, dtype=torch.float32, device=dev)
        prod = K.workspace("prod", dev, rb.n_pairs * max(c_in, c_out))
        stats = rownorm._workspace(dev, c_out)
        _lib.check(_lib.lib().pv2_convbn_forward(
            ctypes.byref(g), _ptr(feats), c_in, _ptr(weight_okc), c_out, _ptr(bn_weight),
            _ptr(bn_bias), _ptr(residual), int(relu), float(eps), float(momentum),
            _ptr(running_mean), _ptr(running_var), _ptr(prod), _ptr(stats), _ptr(y),
            _ptr(mean_invstd), _ptr(out), _stream(feats)), "pv2_convbn_forward")
        ctx.save_for_backward(feats, weight_okc, y, out if relu else None, mean_invstd, bn_weight)
        ctx.rb = rb
        ctx.has_residual = residual is not None
        ctx.has_bias = bn_bias is not None
        return out

    @staticmethod
    def backward(ctx, grad_out):
        feats, weight_okc, y, out, mean_invstd, bn_weight = ctx.saved_tensors
        rb = ctx.rb
        c_out, k, c_in = weight_okc.shape
        dev = grad_out.device
        grad_out = grad_out.contiguous()
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g = rb.geom(c_in, c_out)
        gsum = torch.empty(2 * c_out, dtype=torch.float32, device=dev)
        dy = torch.empty_like(y)
        dres = torch.empty_like(grad_out) if (ctx.has_residual and ctx.needs_input_grad[4]) else None
        dx = torch.empty((rb.n_in, c_in), dtype=torch.float32, device=dev) if need_dx else None
        dw = part = side = None
        if need_dw:
            dw = torch.empty((c_out, k, c_in), dtype=torch.float32, device=dev)
            floats = int(_lib.lib().pv2_spconv_wgrad_partial_floats(c_in, c_out, g.n_tiles_w))
            if sidestream.active(grad_out) and sidestream.safe_leaf(weight_okc):
                # (rb: the pair lists the kernels read.  NOT dw: a second reference to the gradient keeps
                # autograd's AccumulateGrad from adopting it - it then CLONES it on the training stream, at
                # once, while the side stream may not have written it yet.  Found in round 6 as an intermittent
                # garbage weight gradient on the output-stationary route, whose short grad-input lets the
                # training stream reach the accumulation first; ``.grad`` itself keeps dw alive until the join.)
                side = sidestream.native_fork(dev, (feats, dy, rb))
                part = K.workspace("wgrad", dev, floats, stream=side)
            else:
                part = K.workspace("wgrad", dev, floats)
        prod = K.workspace("prod", dev, rb.n_pairs * max(c_in, c_out))
        stats = rownorm._workspace(dev, c_out)
        _lib.check(_lib.lib().pv2_convbn_backward(
            ctypes.byref(g), _ptr(grad_out), _ptr(feats), c_in, _ptr(weight_okc), c_out, _ptr(y),
            _ptr(out), _ptr(mean_invstd), _ptr(bn_weight), _ptr(prod), _ptr(stats), _ptr(gsum),
            _ptr(dy), _ptr(dres), _ptr(dx), _ptr(dw), _ptr(part), _stream(grad_out),
            ctypes.c_void_p(side.cuda_stream) if side is not None else None), "pv2_convbn_backward")
        d_bn_weight = gsum[c_out:] if bn_weight is not None else None
        d_bn_bias = gsum[:c_out] if ctx.has_bias else None
        return dx, dw, d_bn_weight, d_bn_bias, dres, None, None, None, None, None, None


def conv_bn(conv, bn, feats, rb, residual=None, relu=True, weight=None, bias=None):
    """[relu](bn(conv(feats)) [+ residual]) for a sparse conv module ``conv`` (no bias) and an
    ``nn.BatchNorm1d`` ``bn`` on the rulebook ``rb``; ``weight`` / ``bias`` override the BatchNorm's
    affine pair (prompt-driven normalisation, see rownorm.fused_bn).  Callers check ``supported``."""
    w = conv.weight.reshape(conv.out_channels, -1, conv.in_channels)
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        rownorm._bump_batches_tracked(bn)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    bw = bn.weight if weight is None else weight.float().contiguous()
    bb = bn.bias if bias is None else bias.float().contiguous()
    return ConvBNFunction.apply(feats, w, bw, bb, residual, rb, rm, rv, relu, bn.eps, bn.momentum)
