"""NeuS renderer (ponder/models/ponder/render_utils/models/neus.py:8-36): NeuS sampler with the
field's SDF as the proposal, then field evaluation with alphas and compositing weights."""
from functools import partial

from ponderv2_amd import fused_head, narrow_head
from ..builder import RENDERERS
from .base_surface_model import SurfaceModel


@RENDERERS.register_module()
class NeuSModel(SurfaceModel):
    def __init__(self, field, collider, sampler, loss, **kwargs):
        super().__init__(field=field, collider=collider, sampler=sampler, loss=loss)
        self.anneal_end = 50000

    def get_outputs(self, ray_bundle, volume_feature, **kwargs):
        # the shipped head shape runs on the fused ray-march kernels (csrc/raymarch_fused.hip);
        # every other configuration keeps the modular path below
        if fused_head.usable(self, ray_bundle, volume_feature):
            return fused_head.render_outputs(self, ray_bundle, volume_feature)
        # narrow SDF-only heads (the nuScenes configuration): csrc/raymarch_narrow.hip
        if narrow_head.usable(self, ray_bundle, volume_feature):
            return narrow_head.render_outputs(self, ray_bundle, volume_feature)
        # (a projection network that left its final convolution to the fused head applies it now)
        return super().get_outputs(ray_bundle, fused_head.unfold(volume_feature), **kwargs)

    def sample_and_forward_field(self, ray_bundle, volume_feature):
        sampled = self.sampler(ray_bundle, occupancy_fn=self.field.get_occupancy,
                               sdf_fn=partial(self.field.get_sdf, volume_feature=volume_feature))
        ray_samples = sampled.pop("ray_samples")
        field_outputs = self.field(ray_samples, volume_feature, return_alphas=True)
        weights, _ = ray_samples.get_weights_and_transmittance_from_alphas(field_outputs["alphas"])
        return dict(ray_samples=ray_samples, field_outputs=field_outputs, weights=weights,
                    sampled_points=ray_samples.frustums.get_start_positions(), **sampled)
