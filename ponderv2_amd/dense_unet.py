"""The dense projection U-Net (UNet3D-v1m2 behind its first level) as ONE autograd node.

Reference: ponder/models/ponder/unet3d.py - Abstract3DUNet.forward (:646-671): encoders (MaxPool3d(2) ->
SingleConv "bcr", :292-356, :125-156), decoders (ConvTranspose3d(k3, s2, p1) to the skip's size, summed
with the skip, SingleConv "bcr", :359-493).  The modular route (models/ponder/unet3d.py here) makes ~90
autograd nodes of it and ~250 launches per step, each through a Python call, a tensor allocation
and an autograd node of its own: 4 ms of host time around 5 ms of GPU work.  Here forward and backward
are two flat sequences of C-ABI calls (csrc/dense_conv.hip, dense_pool.hip, rownorm.hip) on
pre-sized buffers:

  forward    per encoder level: pool -> BatchNorm statistics -> conv (affine map in, ReLU out);
             per decoder level: transposed conv (+ bias + skip) -> statistics -> conv
  backward   the chain of grad-inputs on the caller's stream (ReLU mask and BatchNorm backward where
             the forward had them; the skip gradient rides in the un-pooling kernel), then every
             weight / bias gradient in one batch on the backward side stream (sidestream.py): nothing
             waits for them before the optimiser.

Same arithmetic as the modular units of dense_conv.py (tests/test_gpu_dense_unet.py compares the two
bit for bit) - only the bookkeeping differs.
"""
import os

import torch

from . import _lib, dense_conv as dc, rownorm, sidestream
from .kernels import _ptr, _stream, workspace

ENABLED = os.environ.get("PV2_DENSE_UNET", "1") != "0"
DENSE_AMP = os.environ.get("PV2_DENSE_AMP", "1") != "0"


class Spec:
    """What the node needs besides tensors: the modules (for eps / momentum / running statistics)."""

    def __init__(self, encoders, decoders, premask_input=False):
        self.encoders = encoders      # [(bn, conv)], levels 1..
        self.decoders = decoders      # [(convT, bn, conv)]
        # x0 is a ReLU's output whose producer expects its gradient already multiplied by (x0 > 0)
        self.premask_input = premask_input


def supported(net, x0):
    """``net``: UNet3Dv1m2; ``x0``: the first level's output.  Every later level must be a pooled "bcr"
    SingleConv / an Upsampling + "bcr" SingleConv the dense kernels cover, in training mode.  An ambient
    autocast region does not change the answer: the node's kernels are fp32 launches autocast never
    touches - under the reference's ``enable_amp=True`` the dense U-Net then runs at HIGHER precision
    than the reference's 16-bit library convolutions (and faster than those: no tuned 16-bit
    3-D convolutions exist for these shapes - 823 ms per step at 8 scenes per GPU through the library's
    fallback, profiles/r04_bench_shipped_*.json)."""
    if not (ENABLED and dc.ENABLED and x0.is_cuda and x0.dtype == torch.float32 and x0.dim() == 5
            and torch.is_grad_enabled()
            and len(net.decoders) == len(net.encoders) - 1 and len(net.encoders) >= 2):
        return False
    shape, c = list(x0.shape[2:]), x0.shape[1]
    widths = [c]
    for enc in net.encoders[1:]:
        m = enc.basic_module
        if (enc.pooling is None or getattr(m, "_order", None) != "bcr" or any(s % 2 or s < 2 for s in shape)
                or c % 4 or not _level_ok(m.batchnorm, m.conv, c)):
            return False
        shape = [s // 2 for s in shape]
        c = m.conv.out_channels
        widths.insert(0, c)
    for dec, skip_c in zip(net.decoders, widths[1:]):
        m, up = dec.basic_module, dec.upsampling.upsample
        if (getattr(m, "_order", None) != "bcr" or not _up_ok(up) or up.in_channels != c
                or up.out_channels != skip_c or not _level_ok(m.batchnorm, m.conv, skip_c)):
            return False
        c = m.conv.out_channels
    return True


def _level_ok(bn, conv, c_in):
    ok_conv = (isinstance(conv, torch.nn.Conv3d) and tuple(conv.kernel_size) == (3, 3, 3)
               and tuple(conv.stride) == (1, 1, 1) and tuple(conv.padding) == (1, 1, 1)
               and tuple(conv.dilation) == (1, 1, 1) and conv.groups == 1 and conv.padding_mode == "zeros"
               and conv.bias is None and conv.in_channels == c_in and c_in % 32 == 0
               and conv.out_channels % 32 == 0 and conv.weight.dtype == torch.float32)
    ok_bn = (isinstance(bn, torch.nn.BatchNorm3d) and not isinstance(bn, torch.nn.SyncBatchNorm)
             and bn.training and bn.affine and bn.momentum is not None and bn.num_features == c_in)
    return ok_conv and ok_bn


def _up_ok(up):
    return (isinstance(up, torch.nn.ConvTranspose3d) and tuple(up.kernel_size) == (3, 3, 3)
            and tuple(up.stride) == (2, 2, 2) and tuple(up.padding) == (1, 1, 1)
            and tuple(up.dilation) == (1, 1, 1) and up.groups == 1 and up.in_channels % 32 == 0
            and up.out_channels % 32 == 0 and up.weight.dtype == torch.float32 and up.bias is not None)


def _vol(b, c, z, y, x, dev):
    return torch.empty((b, z, y, x, c), dtype=torch.float32, device=dev).permute(0, 4, 1, 2, 3)


def _pack(L, st, w, out_dim, flip, mode=0):
    n_out, n_red = w.shape[out_dim], w.shape[1 - out_dim]
    packed = torch.empty(int(L.pv2_dconv3_packed_floats(n_out, n_red, mode)), dtype=torch.float32,
                         device=w.device)
    s = w.stride()
    _lib.check(L.pv2_dconv3_pack_weights(w.data_ptr(), n_out, n_red, s[out_dim], s[1 - out_dim], s[2],
                                         s[3], s[4], int(flip), mode, packed.data_ptr(), st),
               "pv2_dconv3_pack_weights")
    return packed


def _conv(L, st, x, packed, c_out, mode, scale=None, shift=None, mask=None, bias=None, addend=None,
          relu=False, out_mask=None):
    b, c_in, z, y, xx = x.shape
    if mode == 0:
        out = _vol(b, c_out, z, y, xx, x.device)
    elif mode == 1:
        out = _vol(b, c_out, 2 * z, 2 * y, 2 * xx, x.device)
    else:
        out = _vol(b, c_out, z // 2, y // 2, xx // 2, x.device)
    _lib.check(L.pv2_dconv3_forward(x.data_ptr(), b, z, y, xx, c_in, packed.data_ptr(), c_out, mode, scale,
                                    shift, None if mask is None else mask.data_ptr(),
                                    None if bias is None else bias.data_ptr(),
                                    None if addend is None else addend.data_ptr(), int(relu),
                                    None if out_mask is None else out_mask.data_ptr(), out.data_ptr(), st),
               "pv2_dconv3_forward")
    return out


def _bn_conv(L, st, x, bn, bn_w, bn_b, w):
    """statistics of x -> conv(x * scale + shift) -> ReLU; returns (y, mean_invstd, affine)."""
    b, c, z, y, xx = x.shape
    n = b * z * y * xx
    stats = torch.empty(4 * c, dtype=torch.float32, device=x.device)   # mean, invstd | scale, shift
    base = stats.data_ptr()
    rm = bn.running_mean.data_ptr() if bn.track_running_stats else None
    rv = bn.running_var.data_ptr() if bn.track_running_stats else None
    _lib.check(L.pv2_bn_statistics(x.data_ptr(), n, c, bn_w.data_ptr(), bn_b.data_ptr(), float(bn.eps),
                                   float(bn.momentum), rm, rv, _ptr(rownorm._workspace(x.device, c)), base,
                                   base + 8 * c, st), "pv2_bn_statistics")
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        rownorm._bump_batches_tracked(bn)
    out = _conv(L, st, x, _pack(L, st, w, 0, False), w.shape[0], 0, scale=base + 8 * c, shift=base + 12 * c,
                relu=True)
    return out, stats


def _one_term_mode():
    """True inside a 16-bit autocast region (the reference's ``enable_amp = True``,
    ponder/engines/train.py:183-196, runs this network through the library's 16-bit convolutions): the
    node's products then use the leading bf16 piece of each operand only - one MFMA where the fp32 mode
    issues six (pv2_dconv3_set_one_term; sums and results stay fp32).  PV2_DENSE_AMP=0: fp32 products
    whatever the region says (rounds 4 - 5)."""
    return (DENSE_AMP and torch.is_autocast_enabled("cuda")
            and torch.get_autocast_dtype("cuda") in (torch.bfloat16, torch.float16))


class _one_term:
    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        if self.on:
            _lib.check(_lib.lib().pv2_dconv3_set_one_term(1), "pv2_dconv3_set_one_term")

    def __exit__(self, *exc):
        if self.on:
            _lib.lib().pv2_dconv3_set_one_term(0)
        return False


class _DenseUNet(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, spec, *params):
        ctx.one_term = _one_term_mode()
        with _one_term(ctx.one_term):
            return _DenseUNet._forward(ctx, x0, spec, *params)

    @staticmethod
    def backward(ctx, g):
        with _one_term(ctx.one_term):
            return _DenseUNet._backward(ctx, g)

    @staticmethod
    def _forward(ctx, x0, spec, *params):
        L = _lib.lib()
        x0 = dc._cl(x0)
        st = _stream(x0)
        dev = x0.device
        saved, x, pi = [x0], x0, 0
        skips = [x0]
        for bn, conv in spec.encoders:
            bn_w, bn_b, w = params[pi:pi + 3]
            pi += 3
            b, c, z, y, xx = x.shape
            p = _vol(b, c, z // 2, y // 2, xx // 2, dev)
            idx = torch.empty((b, z // 2, y // 2, xx // 2, c // 4), dtype=torch.int32, device=dev)
            _lib.check(L.pv2_maxpool3d_cl_forward(x.data_ptr(), b, z, y, xx, c, p.data_ptr(), idx.data_ptr(), st),
                       "pv2_maxpool3d_cl_forward")
            x, stats = _bn_conv(L, st, p, bn, bn_w, bn_b, w)
            saved += [p, idx, stats, x]
            skips.insert(0, x)
        for (up, bn, conv), skip in zip(spec.decoders, skips[1:]):
            up_w, up_b, bn_w, bn_b, w = params[pi:pi + 5]
            pi += 5
            s = _conv(L, st, x, _pack(L, st, up_w, 1, False, 1), up_w.shape[1], 1, bias=up_b, addend=skip)
            y, stats = _bn_conv(L, st, s, bn, bn_w, bn_b, w)
            saved += [s, stats, y]
            x = y
        ctx.spec = spec
        ctx.n_params = len(params)
        ctx.save_for_backward(*saved, *params)
        return x

    @staticmethod
    def _backward(ctx, g):
        L = _lib.lib()
        spec = ctx.spec
        tensors = ctx.saved_tensors
        params = tensors[len(tensors) - ctx.n_params:]
        saved = tensors[:len(tensors) - ctx.n_params]
        g = dc._cl(g)
        g0 = g
        st = _stream(g)
        dev = g.device
        n_enc, n_dec = len(spec.encoders), len(spec.decoders)
        x0 = saved[0]
        enc = [saved[1 + 4 * i:5 + 4 * i] for i in range(n_enc)]                 # p, idx, stats, y
        dec = [saved[1 + 4 * n_enc + 3 * j:4 + 4 * n_enc + 3 * j] for j in range(n_dec)]   # s, stats, y
        enc_p = [params[3 * i:3 * i + 3] for i in range(n_enc)]
        dec_p = [params[3 * n_enc + 5 * j:3 * n_enc + 5 * j + 5] for j in range(n_dec)]
        grads = [None] * len(params)
        wjobs = []       # weight-gradient launches, issued in one batch at the end
        # every parameter gradient of the node in ONE flat buffer (64-float aligned pieces): the
        # gradient synchronisation then moves them with one copy (utils/grad_sync.py arena blocks)
        # A BatchNorm's (weight, bias) pair shares one block laid out [bias | weight]: that is the {sum g, sum g xhat}
        # row pv2_bn_backward writes, so it writes the pair's gradients IN PLACE (round 6: twelve small
        # device-to-device copies per step on the training stream's chain before).
        bn_pairs = [(3 * i, 3 * i + 1) for i in range(n_enc)] + \
                   [(3 * n_enc + 5 * j + 2, 3 * n_enc + 5 * j + 3) for j in range(n_dec)]
        pair_of_bias = {ib: iw for iw, ib in bn_pairs}
        offs, total = [0] * len(params), 0
        for i, p in enumerate(params):
            if i in pair_of_bias:
                continue                      # placed with its weight
            if any(i == iw for iw, _ in bn_pairs):
                c = p.numel()
                offs[i + 1], offs[i] = total, total + c
                total += (2 * c + 63) // 64 * 64
            else:
                offs[i] = total
                total += (p.numel() + 63) // 64 * 64
        arena = torch.empty(total, dtype=torch.float32, device=dev)
        # (same strides as the parameter - the conv weights are channels_last_3d -, like empty_like)
        pviews = [arena[o:o + p.numel()].as_strided(p.shape, p.stride()) for o, p in zip(offs, params)]

        def weight_gradients(jobs):
            stw = _stream(g0)    # (the side stream when forked)
            out = []
            for kind, x, scale, shift, gy, mask, w, slot in jobs:
                if kind == 2:    # bias of the transposed conv: column sums of the gradient rows
                    rows = gy.permute(0, 2, 3, 4, 1).reshape(-1, gy.shape[1])
                    gb = pviews[slot]
                    _lib.check(L.pv2_col_sum(rows.data_ptr(), rows.shape[0], rows.shape[1], gb.data_ptr(), stw),
                               "pv2_col_sum")
                    out.append((slot, gb))
                    continue
                b, c_x, z, yy, xx = x.shape
                c_g = gy.shape[1]
                floats = int(L.pv2_dconv3_wgrad_partial_floats(b, z, yy, xx, c_x, c_g, kind))
                part = workspace("dconv_wgrad", dev, floats)
                dw = pviews[slot]
                sw = dw.stride()
                n_dim = 0 if kind == 0 else 1
                _lib.check(L.pv2_dconv3_backward_weight(
                    x.data_ptr(), b, z, yy, xx, c_x, scale, shift, gy.data_ptr(), c_g,
                    None if mask is None else mask.data_ptr(), kind, _ptr(part), dw.data_ptr(), sw[n_dim],
                    sw[1 - n_dim], sw[2], sw[3], sw[4], stw), "pv2_dconv3_backward_weight")
                out.append((slot, dw))
            return out

        # Weight gradients: on the backward side stream (sidestream.py), each level's forked right BEHIND the launch
        # of that level's grad-input kernel - it then runs beside the BatchNorm backward / un-pooling passes that
        # follow (memory-bound, the matrix pipe idle) instead of, all levels in one batch at the end of the node
        # (rounds 4 - 5), beside the sparse backbone's first grad-input products.  PV2_DENSE_WGRAD_BATCH=1: one batch.
        leaf_params = [p_[2] for p_ in enc_p] + [q for p_ in dec_p for q in (p_[0], p_[1], p_[4])]
        forked = sidestream.active(g) and all(sidestream.safe_leaf(w_) for w_ in leaf_params)
        per_level = forked and not WGRAD_ONE_BATCH
        done = []

        def run_jobs(jobs):
            if per_level:
                keep = tuple(t for job in jobs for t in job if torch.is_tensor(t)) + tuple(saved) + (arena,)
                done.extend(sidestream.fork(lambda: weight_gradients(jobs), keep))
            else:
                wjobs.extend(jobs)

        def level_backward(g, x, stats, y, bn_w, w, slot, masked, gsum):
            """through relu(conv(bn(x))): returns d/dx; queues d/dw; fills the BatchNorm gradients.
            ``masked``: ``g`` already passed the ReLU backwards (its producer zeroed it where y <= 0);
            otherwise both consumers apply the mask while they stage it."""
            c = x.shape[1]
            base = stats.data_ptr()
            m = None if masked else y
            gxn = _conv(L, st, g, _pack(L, st, w, 1, True), c, 0, mask=m)
            run_jobs([(0, x, base + 8 * c, base + 12 * c, g, m, w, slot)])
            b, _, z, yy, xx = x.shape
            n = b * z * yy * xx
            gx = _vol(b, c, z, yy, xx, dev)
            _lib.check(L.pv2_bn_backward(gxn.data_ptr(), x.data_ptr(), None, base, bn_w.data_ptr(), n, c,
                                         _ptr(rownorm._workspace(dev, c)), gsum.data_ptr(), gx.data_ptr(), None,
                                         st), "pv2_bn_backward")
            return gx

        gskip = [None] * n_dec
        for j in reversed(range(n_dec)):
            s, stats, y = dec[j]
            up_w, up_b, bn_w, bn_b, w = dec_p[j]
            k = 3 * n_enc + 5 * j
            # (the node's incoming gradient is the only one that arrives unmasked)
            c = s.shape[1]
            assert bn_w.numel() == c and offs[k + 2] == offs[k + 3] + c
            gs = level_backward(g, s, stats, y, bn_w, w, k + 4, masked=j != n_dec - 1,
                                gsum=arena[offs[k + 3]:offs[k + 3] + 2 * c])
            grads[k + 2], grads[k + 3] = pviews[k + 2], pviews[k + 3]
            gskip[j] = gs
            x_in = enc[n_enc - 1][3] if j == 0 else dec[j - 1][2]
            # grad-input of the transposed conv = the gradient of x_in, a ReLU's output (the level
            # below): masked in this kernel's epilogue
            g = _conv(L, st, gs, _pack(L, st, up_w, 0, False, 2), up_w.shape[0], 2, out_mask=x_in)
            run_jobs([(1, x_in, None, None, gs, None, up_w, k), (2, None, None, None, gs, None, up_b, k + 1)])
        for i in reversed(range(n_enc)):
            p, idx, stats, y = enc[i]
            bn_w, bn_b, w = enc_p[i]
            c = p.shape[1]
            assert bn_w.numel() == c and offs[3 * i] == offs[3 * i + 1] + c
            gp = level_backward(g, p, stats, y, bn_w, w, 3 * i + 2, masked=True,
                                gsum=arena[offs[3 * i + 1]:offs[3 * i + 1] + 2 * c])
            grads[3 * i], grads[3 * i + 1] = pviews[3 * i], pviews[3 * i + 1]
            b, _, z, yy, xx = p.shape
            add = gskip[n_enc - 1 - i]      # the pooled tensor is also that decoder level's skip
            g = _vol(b, c, 2 * z, 2 * yy, 2 * xx, dev)
            # the pooled tensor: the previous encoder level's ReLU output (mask its gradient here), or
            # the node's input x0 (whatever produced it owns its own backward)
            below = (enc[i - 1][3].data_ptr() if i > 0
                     else x0.data_ptr() if spec.premask_input else None)
            _lib.check(L.pv2_maxpool3d_cl_backward_add(gp.data_ptr(), idx.data_ptr(), add.data_ptr(), below, b,
                                                       2 * z, 2 * yy, 2 * xx, c, g.data_ptr(), st),
                       "pv2_maxpool3d_cl_backward_add")

        if wjobs and forked:      # (one batch)
            keep = tuple(t for job in wjobs for t in job if torch.is_tensor(t)) + tuple(saved) + (arena,)
            done.extend(sidestream.fork(lambda: weight_gradients(wjobs), keep))
        elif wjobs:
            done.extend(weight_gradients(wjobs))
        for slot, t in done:
            grads[slot] = t
        gx0 = g if ctx.needs_input_grad[0] else None
        return (gx0, None) + tuple(grads)


WGRAD_ONE_BATCH = os.environ.get("PV2_DENSE_WGRAD_BATCH", "0") == "1"


def forward(net, x0, premask_input=False):
    """encoders[1:] and decoders of ``net`` (UNet3Dv1m2) applied to the first level's output.
    ``premask_input``: x0 is a ReLU output and its producer takes the gradient already masked with
    (x0 > 0) - applied for free in the epilogue of the node's last backward kernel
    (cells_level.claim_premasked)."""
    encoders = [(e.basic_module.batchnorm, e.basic_module.conv) for e in net.encoders[1:]]
    decoders = [(d.upsampling.upsample, d.basic_module.batchnorm, d.basic_module.conv) for d in net.decoders]
    params = []
    for bn, conv in encoders:
        params += [bn.weight, bn.bias, conv.weight]
    for up, bn, conv in decoders:
        params += [up.weight, up.bias, bn.weight, bn.bias, conv.weight]
    return _DenseUNet.apply(x0, Spec(encoders, decoders, premask_input), *params)
