"""Alpha-compositing reductions (ponder/models/ponder/render_utils/renderers.py: RGB :5-28,
depth :31-51 incl. the clip to the global [min, max] of the sample starts :50, normal :54-63,
semantic :66-75)."""
import torch

from ponderv2_amd import raymarch
from torch import nn

from .rays import device_constant


def _composite(weights, values):
    """sum over the sample axis of weights (R,S,1) * values (R,S,F) -> (R,F)."""
    if raymarch.ENABLED and raymarch.supported(weights, values):
        return raymarch.weighted_sum(weights, values)  # one launch each way (csrc/raymarch.hip)
    return torch.sum(weights * values, dim=-2)


class RGBRenderer(nn.Module):
    def __init__(self, background_color=(0.0, 0.0, 0.0)):
        super().__init__()
        self.background_color = background_color

    def forward(self, rgb, weights):
        comp = _composite(weights, rgb)
        acc = torch.sum(weights, dim=-2)
        comp = comp + device_constant(self.background_color, comp.device, comp.dtype) * (1.0 - acc)
        if not self.training:
            comp = comp.clamp(0.0, 1.0)
        return comp


class DepthRenderer(nn.Module):
    def forward(self, ray_samples, weights):
        steps = ray_samples.frustums.starts
        depth = _composite(weights, steps) / (torch.sum(weights, -2) + 1e-10)
        B = getattr(ray_samples, "num_scenes", 1)
        if B == 1:
            return torch.clip(depth, steps.min(), steps.max())
        per_scene = steps.reshape(B, -1)  # the clip range is per rendered scene (reference :50)
        lo, hi = per_scene.amin(1), per_scene.amax(1)
        lo = lo.repeat_interleave(depth.shape[0] // B).reshape(-1, 1)
        hi = hi.repeat_interleave(depth.shape[0] // B).reshape(-1, 1)
        return torch.maximum(torch.minimum(depth, hi), lo)


class NormalRenderer(nn.Module):
    def forward(self, normals, weights):
        return _composite(weights, normals)


class SemanticRenderer(nn.Module):
    def forward(self, semantic, weights):
        return _composite(weights, semantic)
