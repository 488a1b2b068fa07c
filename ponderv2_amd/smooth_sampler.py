"""Drop-in for the reference's ``smooth_sampler`` package
(libs/smooth-sampler/smooth_sampler/modules.py:14-101): ``SmoothSampler.apply(input, grid,
padding_mode, align_corners, apply_smoothstep)`` with first and second order gradients, backed by
the gfx950 kernels in csrc/trilinear.hip instead of the CUDA extension ``smooth_sampler._C``.

Differences that are deliberate (same results, fewer host round trips):
  * the all-zero tests on incoming gradients (modules.py:45-47,90: ``.item()`` device syncs) are
    replaced by static ``needs_input_grad`` / ``None`` checks;
  * a channels-last (NDHWC) ``input`` is consumed as is, and the sampled features come back with
    the channel axis innermost, so ``out.squeeze(0).squeeze(1).permute(1, 2, 0)`` is a free view.
"""
import torch

from . import kernels as K


class SmoothSamplerBackward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, grid, grad_out, padding_mode="zeros", align_corners=True,
                apply_smoothstep=False):
        ctx.cfg = (padding_mode, align_corners, apply_smoothstep)
        ctx.set_materialize_grads(False)  # an unused grad_input output must not cost 268 MB of zeros
        need_in = input.requires_grad
        grad_input, grad_grid = K.trilinear_backward(
            grad_out, input, grid, padding_mode, align_corners, apply_smoothstep, need_in)
        ctx.save_for_backward(input, grid, grad_out)
        ctx.had_grad_input = need_in
        if grad_input is None:
            grad_input = torch.zeros((), dtype=input.dtype, device=input.device)  # placeholder
            ctx.mark_non_differentiable(grad_input)
        return grad_input, grad_grid

    @staticmethod
    def backward(ctx, g_ginput, g_ggrid):
        input, grid, grad_out = ctx.saved_tensors
        padding_mode, align_corners, smooth = ctx.cfg
        if g_ginput is None and g_ggrid is None:
            return None, None, None, None, None, None
        if not ctx.had_grad_input:
            g_ginput = None
        if g_ggrid is None:
            g_ggrid = torch.zeros_like(grid)
        grad_input2, grad_grid2, gg_out = K.trilinear_backward_backward(
            g_ginput, g_ggrid, input, grid, grad_out, padding_mode, align_corners, smooth,
            ctx.needs_input_grad[0])
        return grad_input2, grad_grid2, gg_out, None, None, None


class SmoothSampler(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, grid, padding_mode="zeros", align_corners=True,
                apply_smoothstep=False):
        output = K.trilinear_forward(input, grid, padding_mode, align_corners, apply_smoothstep)
        ctx.save_for_backward(input, grid)
        ctx.cfg = (padding_mode, align_corners, apply_smoothstep)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        input, grid = ctx.saved_tensors
        padding_mode, align_corners, smooth = ctx.cfg
        d_input, d_grid = SmoothSamplerBackward.apply(
            input, grid, grad_out, padding_mode, align_corners, smooth)
        if not ctx.needs_input_grad[0]:
            d_input = None
        return d_input, d_grid, None, None, None
