"""fp32 MFMA linear layers for the render field's MLP heads.

``linear(x, weight, bias)`` has ``torch.nn.functional.linear`` semantics.  On the device it runs on
two gfx950 kernels (csrc/sparse_conv.hip: ``tall_gemm_nt_kernel`` and the identity mode of
``spconv_wgrad_lds_kernel``) instead of rocBLAS/hipBLASLt, whose fp32 heuristics pick tiny macro
tiles for these tall-skinny shapes (M = rays x samples ~ 1e5, K, N <= 512).

The two autograd Functions are closed under differentiation - the backward of each is written
with the other - so the graph can be differentiated any number of times, which the NeuS head needs
(eikonal / normal / colour terms depend on d(sdf)/d(points)):

    TallGemmNT(X[M,K], W[N,K]) = X W^T         dX = TallGemmNT(dY, W^T)     dW = ReduceGemmTN(dY, X)
    ReduceGemmTN(A[M,I], B[M,J]) = A^T B       dA = TallGemmNT(B, G)        dB = TallGemmNT(A, G^T)

Feature dimensions are zero-padded to multiples of 8 at this level (134 -> 136, 65 -> 72, 3 -> 8):
the kernels only see 16-byte aligned rows.
"""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib
from .kernels import _ptr, _require_device, _stream
from .rownorm import col_sum

MIN_ROWS = 2048  # below this the GEMM is launch-bound either way: leave it to the BLAS


class TallGemmNT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias=None):
        _require_device(x, w)
        x, w = x.contiguous(), w.contiguous()
        m, k = x.shape
        n = w.shape[0]
        assert w.shape[1] == k and k % 8 == 0, (x.shape, w.shape)
        y = torch.empty((m, n), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().pv2_gemm_nt(_ptr(x), m, k, _ptr(w), n, _ptr(bias), _ptr(y),
                                          _stream(x)), "pv2_gemm_nt")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = tall_gemm_nt(gy, w.t())
        if ctx.needs_input_grad[1]:
            gw = reduce_gemm_tn(gy, x)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = col_sum(gy)
        return gx, gw, gb


class ReduceGemmTN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        _require_device(a, b)
        a, b = a.contiguous(), b.contiguous()
        m, i = a.shape
        j = b.shape[1]
        assert b.shape[0] == m and i % 4 == 0 and j % 4 == 0, (a.shape, b.shape)
        c = torch.zeros((i, j), dtype=torch.float32, device=a.device)
        _lib.check(_lib.lib().pv2_gemm_tn(_ptr(a), _ptr(b), m, i, j, _ptr(c), _stream(a)),
                   "pv2_gemm_tn")
        ctx.save_for_backward(a, b)
        return c

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = tall_gemm_nt(b, g)
        if ctx.needs_input_grad[1]:
            gb = tall_gemm_nt(a, g.t())
        return ga, gb


def _pad_last(t, mult):
    r = (-t.shape[-1]) % mult
    return t if r == 0 else F.pad(t, (0, r))


def tall_gemm_nt(x, w, bias=None):
    """x [M,K] @ w[N,K]^T (+ bias) with K zero-padded to a multiple of 8."""
    return TallGemmNT.apply(_pad_last(x, 8), _pad_last(w, 8), bias)


def reduce_gemm_tn(a, b):
    """a[M,I]^T @ b[M,J] with I, J zero-padded to multiples of 4."""
    i, j = a.shape[1], b.shape[1]
    c = ReduceGemmTN.apply(_pad_last(a, 4), _pad_last(b, 4))
    return c if c.shape == (i, j) else c[:i, :j]


def linear(x, weight, bias=None):
    """``F.linear`` on the MFMA kernels for large device batches (fp32 arithmetic also inside
    autocast regions: the kernels are launches autocast never touches, reduced-precision inputs are
    widened on the way in); small batches and host tensors under the test doubles go to torch."""
    rows = x.numel() // max(x.shape[-1], 1)
    if (not x.is_cuda or x.dtype not in (torch.float32, torch.bfloat16, torch.float16)
            or weight.dtype != torch.float32 or rows < MIN_ROWS):
        return F.linear(x, weight, bias)
    x = x.float()
    y = tall_gemm_nt(x.reshape(rows, x.shape[-1]), weight, bias)
    return y.reshape(*x.shape[:-1], weight.shape[0])
