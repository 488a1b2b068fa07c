"""Fused training-mode BatchNorm1d (+ residual add + ReLU) on the active-voxel feature matrix and
a column-sum, on the gfx950 kernels of csrc/rownorm.hip.

``fused_bn(bn_module, x, residual=None, relu=False)`` uses the parameters and running buffers of a
stock ``nn.BatchNorm1d`` (so state_dicts stay reference-compatible) and computes
``[relu](bn(x) [+ residual])`` in two short launches, and two for the backward (per-block partial
statistics, then an apply kernel that adds them in a fixed order: no atomics, csrc/rownorm.hip).  Anything the kernels
do not cover (eval mode, other dtypes, host tensors under the test doubles) takes the module path.
"""
import torch
import torch.nn.functional as F

from . import _lib, precision
from .kernels import DTYPE_CODE, _ptr, _require_device, _stream


_WORKSPACES = {}
_WS_FLOATS = {}
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_of(device):
    if _raw_stream is not None:
        return _raw_stream(device.index if device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(device).cuda_stream


def _workspace(device, channels):
    """The statistics kernels' scratch (per-block partial sums, csrc/rownorm.hip): launches on one
    stream are ordered, so ONE buffer per stream serves every layer.  Grown on demand."""
    key = (device.index, _stream_of(device))
    need = _WS_FLOATS.get(channels)
    if need is None:
        need = _WS_FLOATS[channels] = int(_lib.lib().pv2_bn_workspace_floats(channels))
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 64 * 2 * 256), dtype=torch.float32, device=device)
        _WORKSPACES[key] = ws
    return ws


def _bump_batches_tracked(bn):
    """``num_batches_tracked += 1`` without a kernel launch per layer per step: counted on the host
    and written back whenever the module's state is read (state_dict) - the value only matters for
    checkpoints here (momentum is not None)."""
    if not hasattr(bn, "_pv2_pending_batches"):
        bn._pv2_pending_batches = 0

        def flush(module, *args):
            if module._pv2_pending_batches and module.num_batches_tracked is not None:
                module.num_batches_tracked.add_(module._pv2_pending_batches)
            module._pv2_pending_batches = 0

        def forget(*args):   # the loaded buffer is the truth: steps counted before it are history
            bn._pv2_pending_batches = 0

        bn.register_state_dict_pre_hook(flush)
        bn._register_load_state_dict_pre_hook(forget)
    bn._pv2_pending_batches += 1


def flush_bn_counters(model):
    """Write every pending ``num_batches_tracked`` count to its buffer.  Call before anything reads
    the buffers directly - ``model.buffers()`` broadcasts, ``copy.deepcopy`` for an EMA model;
    ``state_dict()`` does it by itself."""
    for m in model.modules():
        pending = getattr(m, "_pv2_pending_batches", 0)
        if pending and getattr(m, "num_batches_tracked", None) is not None:
            m.num_batches_tracked.add_(pending)
            m._pv2_pending_batches = 0


class _FusedBNFunction(torch.autograd.Function):
    """x (fp32 or 16-bit) -> y in ``out_dtype``; the residual and the incoming gradient are in
    ``out_dtype``, dx in x's dtype; parameters, statistics and their gradients fp32."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, running_mean, running_var, relu, eps, momentum,
                out_dtype):
        _require_device(x)
        x = x.contiguous()
        n, c = x.shape
        if residual is not None:
            residual = residual.to(out_dtype).contiguous()
        y = torch.empty((n, c), dtype=out_dtype, device=x.device)
        sums = _workspace(x.device, c)
        mean_invstd = torch.empty(2 * c, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().pv2_bn_forward_mixed(
            _ptr(x), DTYPE_CODE[x.dtype], n, c, _ptr(weight), _ptr(bias), _ptr(residual), int(relu),
            float(eps), float(momentum), _ptr(running_mean), _ptr(running_var), _ptr(sums),
            _ptr(mean_invstd), _ptr(y), DTYPE_CODE[out_dtype], _stream(x)), "pv2_bn_forward")
        ctx.save_for_backward(x, y if relu else None, mean_invstd, weight)
        ctx.has_residual = residual is not None
        ctx.has_bias = bias is not None
        ctx.out_dtype = out_dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean_invstd, weight = ctx.saved_tensors
        dy = dy.to(ctx.out_dtype).contiguous()
        n, c = x.shape
        dx = torch.empty_like(x)
        dres = torch.empty_like(dy) if ctx.has_residual else None
        gsum = torch.empty(2 * c, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().pv2_bn_backward_mixed(
            _ptr(dy), _ptr(x), DTYPE_CODE[x.dtype], _ptr(y), DTYPE_CODE[ctx.out_dtype],
            _ptr(mean_invstd), _ptr(weight), n, c, _ptr(_workspace(x.device, c)), _ptr(gsum),
            _ptr(dx), _ptr(dres), _stream(x)), "pv2_bn_backward")
        dweight = gsum[c:] if weight is not None else None
        dbias = gsum[:c] if ctx.has_bias else None
        return dx, dweight, dbias, dres, None, None, None, None, None, None


def _momentum(bn):
    """The running-average factor of this call (``momentum=None``: the cumulative average)."""
    if bn.momentum is not None:
        return bn.momentum
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        return 1.0 / float(bn.num_batches_tracked)
    return 0.0


def can_fuse(bn, x):
    """True when the rownorm.hip kernels cover this call: a training-mode device matrix.  Autocast
    does not change the answer - the kernels are fp32 launches it never touches; reduced-precision
    inputs are widened on the way in and the result is fp32."""
    return (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16, torch.float16) and bn.training
            and x.dim() == 2 and x.shape[0] > 1 and bn.momentum is not None
            and bn.weight.dtype == torch.float32
            and not isinstance(bn, torch.nn.SyncBatchNorm))  # SyncBN reduces across ranks itself


def fused_bn(bn, x, residual=None, relu=False, weight=None, bias=None):
    """[relu](bn(x) [+ residual]) with ``bn`` an nn.BatchNorm1d.

    ``weight`` / ``bias`` (C,) replace the module's affine pair in the kernel epilogue: prompt-driven
    normalisation folds its per-condition modulation ``y * (1 + scale) + shift`` into them
    (spconv_unet_v1m3_pdnorm.py), gradients flow back to whatever produced them.  Callers that
    may pass them whatever ``can_fuse`` says: the fallback is ``F.batch_norm`` with the same pair."""
    if not can_fuse(bn, x):
        if weight is None and bias is None:
            y = bn(x)
        else:
            # the same normalisation with the caller's affine pair (a one-row output of a strided
            # conv, eval mode, a host tensor: whatever kept the kernels away)
            if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
                bn.num_batches_tracked.add_(1)
            use_batch = bn.training or bn.running_mean is None
            y = F.batch_norm(x, None if use_batch and not bn.track_running_stats else bn.running_mean,
                             None if use_batch and not bn.track_running_stats else bn.running_var,
                             bn.weight if weight is None else weight.to(x.dtype),
                             bn.bias if bias is None else bias.to(x.dtype),
                             use_batch, _momentum(bn), bn.eps)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        _bump_batches_tracked(bn)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    w = bn.weight if weight is None else weight.float().contiguous()
    b = bn.bias if bias is None else bias.float().contiguous()
    # 16-bit feature matrices stay 16-bit; fp32 ones become 16-bit where the reduced-precision
    # training mode says so (precision.sparse_dtype: the stem's output is the fp32 -> 16-bit edge)
    out_dtype = x.dtype if x.dtype != torch.float32 else (precision.sparse_dtype() or torch.float32)
    return _FusedBNFunction.apply(x, w, b, residual, rm, rv, relu, bn.eps, bn.momentum, out_dtype)


class _ColSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _require_device(x)
        x = x.contiguous()
        ctx.rows = x.shape[0]
        out = torch.empty(x.shape[1], dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().pv2_col_sum(_ptr(x), x.shape[0], x.shape[1], _ptr(out), _stream(x)),
                   "pv2_col_sum")
        return out

    @staticmethod
    def backward(ctx, g):
        return g.unsqueeze(0).expand(ctx.rows, -1)


def col_sum(x):
    """x[M, N].sum(0) for large fp32 device matrices."""
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= 4096:
        return _ColSum.apply(x)
    return x.sum(0)
