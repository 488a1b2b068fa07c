"""Near/far assignment for ray bundles.  Restates
ponder/models/ponder/render_utils/scene_colliders.py (AABBBoxCollider :23-99 incl. the
``1/(d+1e-6)`` slab test :38-85; NearFarCollider :102-120)."""
import torch
import torch.nn as nn

from .builder import COLLIDERS
from .rays import device_constant


class SceneCollider(nn.Module):
    def __init__(self, **kwargs):
        self.kwargs = kwargs
        super().__init__()

    def set_nears_and_fars(self, ray_bundle):
        raise NotImplementedError

    def forward(self, ray_bundle):
        if ray_bundle.nears is not None and ray_bundle.fars is not None:
            return ray_bundle
        return self.set_nears_and_fars(ray_bundle)


@COLLIDERS.register_module()
class AABBBoxCollider(SceneCollider):
    def __init__(self, bbox, near_plane, **kwargs):
        super().__init__(**kwargs)
        self.bbox = bbox  # [xmin, ymin, zmin, xmax, ymax, zmax]
        self.near_plane = near_plane

    def _intersect_with_aabb(self, rays_o, rays_d, aabb):
        inv = 1.0 / (rays_d + 1e-6)
        lo = device_constant(aabb[:3], rays_o.device, rays_o.dtype)
        hi = device_constant(aabb[3:], rays_o.device, rays_o.dtype)
        ta, tb = (lo - rays_o) * inv, (hi - rays_o) * inv
        nears = torch.minimum(ta, tb).max(dim=1).values
        fars = torch.maximum(ta, tb).min(dim=1).values
        nears = torch.clamp(nears, min=self.near_plane)
        hit = nears < fars
        zero = torch.zeros_like(nears)
        return torch.where(hit, nears, zero), torch.where(hit, fars, zero)

    def set_nears_and_fars(self, ray_bundle):
        nears, fars = self._intersect_with_aabb(ray_bundle.origins, ray_bundle.directions, self.bbox)
        ray_bundle.nears, ray_bundle.fars = nears[..., None], fars[..., None]
        return ray_bundle


@COLLIDERS.register_module()
class NearFarCollider(SceneCollider):
    def __init__(self, near_plane, far_plane, **kwargs):
        super().__init__(**kwargs)
        self.near_plane, self.far_plane = near_plane, far_plane

    def set_nears_and_fars(self, ray_bundle):
        ones = torch.ones_like(ray_bundle.origins[..., 0:1])
        ray_bundle.nears, ray_bundle.fars = ones * self.near_plane, ones * self.far_plane
        return ray_bundle
