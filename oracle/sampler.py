"""torch-CPU restatement of the twice-differentiable trilinear grid sampler, composed from
differentiable index / lerp ops so autograd can differentiate it to any order.

Follows libs/smooth-sampler/smooth_sampler/csrc/smooth_sampler_kernel.cu:73-152 (forward; the
backward kernels :207-355 and :419-618 are what autograd derives from this composition) and ATen's
GridSampler.cuh coordinate helpers (unnormalise / clip / reflect, with their gradient multipliers
falling out of autograd).  API = SmoothSampler.apply of smooth_sampler/modules.py:63-101.
"""
import torch


def _unnormalize(g, size, align_corners):
    if align_corners:
        return ((g + 1.0) / 2.0) * (size - 1)
    return ((g + 1.0) * size - 1.0) / 2.0


def _clip(x, size):
    return x.clamp(min=0.0, max=float(size - 1))


def _reflect(x, twice_low, twice_high):
    if twice_low == twice_high:
        return torch.zeros_like(x)
    lo = twice_low / 2.0
    span = (twice_high - twice_low) / 2.0
    v = (x - lo).abs()
    extra = torch.fmod(v, span)
    flips = torch.floor(v / span)
    even = (flips % 2) == 0
    return torch.where(even, extra + lo, span - extra + lo)


def _source_index(g, size, padding_mode, align_corners):
    x = _unnormalize(g, size, align_corners)
    if padding_mode == "border":
        x = _clip(x, size)
    elif padding_mode == "reflection":
        if align_corners:
            x = _reflect(x, 0, 2 * (size - 1))
        else:
            x = _reflect(x, -1, 2 * size - 1)
        x = _clip(x, size)
    return x


def _smoothstep(t):
    return t * t * (3.0 - 2.0 * t)


def smooth_sample(input, grid, padding_mode="zeros", align_corners=True, apply_smoothstep=False):
    """input (N,C,D,H,W), grid (N,Do,Ho,Wo,3) -> (N,C,Do,Ho,Wo)."""
    N, C, D, H, W = input.shape
    _, Do, Ho, Wo, _ = grid.shape
    ix = _source_index(grid[..., 0], W, padding_mode, align_corners)
    iy = _source_index(grid[..., 1], H, padding_mode, align_corners)
    iz = _source_index(grid[..., 2], D, padding_mode, align_corners)
    x0, y0, z0 = torch.floor(ix).detach(), torch.floor(iy).detach(), torch.floor(iz).detach()
    tx, ty, tz = ix - x0, iy - y0, iz - z0
    if apply_smoothstep:
        tx, ty, tz = _smoothstep(tx), _smoothstep(ty), _smoothstep(tz)
    x0, y0, z0 = x0.long(), y0.long(), z0.long()
    flat = input.reshape(N, C, D * H * W)
    out = input.new_zeros((N, C, Do, Ho, Wo))
    for cz in (0, 1):
        for cy in (0, 1):
            for cx in (0, 1):
                xi, yi, zi = x0 + cx, y0 + cy, z0 + cz
                w = (tx if cx else 1 - tx) * (ty if cy else 1 - ty) * (tz if cz else 1 - tz)
                inb = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H) & (zi >= 0) & (zi < D)
                lin = (zi.clamp(0, D - 1) * H + yi.clamp(0, H - 1)) * W + xi.clamp(0, W - 1)
                vals = torch.gather(flat, 2, lin.reshape(N, 1, -1).expand(N, C, -1))
                vals = vals.reshape(N, C, Do, Ho, Wo)
                out = out + vals * (w * inb.to(w.dtype)).unsqueeze(1)
    return out


class SmoothSampler:
    """Same call shape as the reference's autograd.Function: ``SmoothSampler.apply(...)``."""

    @staticmethod
    def apply(input, grid, padding_mode="zeros", align_corners=True, apply_smoothstep=False):
        return smooth_sample(input, grid, padding_mode, align_corners, apply_smoothstep)
