"""Reductions that stay correct inside a captured hipGraph.

ATen splits a reduction over several workgroups once a thread would fold >= 256 values (>= 131 072
inputs per output with its 512-thread blocks; torch/include/ATen/native/cuda/Reduce.cuh:1173-1183);
the workgroups meet at a semaphore array that the kernel only increments (:690-702) and that the
host clears with ``cudaMemsetAsync`` before every launch (:1294-1301).  On this ROCm build
memset nodes of a captured graph are not reliably re-executed on replay (csrc/common.h records the
same finding for this library's own memsets), so from the second replay on the semaphores are stale
and the reduction's output is whatever its buffer last held.  That is what turned the render head's
sdf / free-space / eikonal loss VALUES (sums over ~135 k samples) into garbage on some replays
while every loss over the 1 k rays stayed right.

While the current stream is capturing, the helpers below route such reductions through
``pv2_col_sum`` (its accumulator is cleared by a kernel) or restructure them so that no single
output reduces more than a few hundred values; outside capture they are the plain torch ops.
"""
import torch

from .rownorm import _ColSum


def capturing(t):
    return t.is_cuda and torch.cuda.is_current_stream_capturing()


def sum_all(x):
    """x.sum() -> 0-dim tensor (differentiable, any number of times)."""
    if capturing(x) and x.numel() >= 4096:
        return _ColSum.apply(x.to(torch.float32).reshape(-1, 1)).reshape(())
    return x.sum()


def mean_all(x):
    return sum_all(x) / x.numel()


class _ScaleByScalar(torch.autograd.Function):
    """x * s with s a one-element parameter-derived tensor: the gradient of s is a reduction over
    all of x, done with ``sum_all``."""

    @staticmethod
    def forward(ctx, x, s):
        ctx.save_for_backward(x, s)
        return x * s

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        gx = g * s if ctx.needs_input_grad[0] else None
        gs = sum_all(g * x).reshape(s.shape) if ctx.needs_input_grad[1] else None
        return gx, gs


def scale_by_scalar(x, s):
    if capturing(x) and s.numel() == 1 and x.numel() >= 4096:
        return _ScaleByScalar.apply(x, s)
    return x * s


def rowwise_min_max(x, group):
    """(x.amin(1), x.amax(1)) for x of shape (B, n) with n a multiple of ``group``: reduced in two
    steps (``group`` values, then n / group values per output) - identical results, and neither
    step is large enough for the split (semaphore) reduction path."""
    b, n = x.shape
    if group > 1 and n % group == 0:
        y = x.reshape(b, n // group, group)
        return y.amin(2).amin(1), y.amax(2).amax(1)
    return x.amin(1), x.amax(1)
