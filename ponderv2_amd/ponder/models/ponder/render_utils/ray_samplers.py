"""Ray samplers of the NeuS head.

Restates ponder/models/ponder/render_utils/ray_samplers.py: SpacedSampler/UniformSampler
:36-127 (stratified bin edges :70-88), PDFSampler :206-322 (inverse-CDF importance sampling
:243-313), NeuSSampler :325-463 (coarse pass -> fixed-inv_s alphas :426-463 -> PDF resample ->
sorted merge).  Random numbers are drawn through ``self.rand`` so parity tests can inject the
reference's jitter.
"""
import torch
from torch import nn

from .builder import SAMPLERS
from .rays import alphas_to_weights, device_linspace


class Sampler(nn.Module):
    def __init__(self, num_samples=None):
        super().__init__()
        self.num_samples = num_samples
        self.rand = torch.rand  # injectable RNG: rand(shape, dtype=, device=)

    def generate_ray_samples(self, *args, **kwargs):
        raise NotImplementedError

    def forward(self, *args, **kwargs):
        return self.generate_ray_samples(*args, **kwargs)


class SpacedSampler(Sampler):
    def __init__(self, spacing_fn, spacing_fn_inv, num_samples=None, train_stratified=True,
                 single_jitter=False):
        super().__init__(num_samples=num_samples)
        self.train_stratified, self.single_jitter = train_stratified, single_jitter
        self.spacing_fn, self.spacing_fn_inv = spacing_fn, spacing_fn_inv

    def generate_ray_samples(self, ray_bundle, num_samples=None):
        assert ray_bundle is not None and ray_bundle.nears is not None and ray_bundle.fars is not None
        n = num_samples or self.num_samples
        rays = ray_bundle.origins.shape[0]
        dev = ray_bundle.origins.device
        bins = device_linspace(0.0, 1.0, n + 1, dev).expand(rays, -1)
        if self.train_stratified and self.training:
            t_rand = self.rand((rays, 1 if self.single_jitter else n + 1), dtype=bins.dtype,
                               device=dev)
            centers = (bins[..., 1:] + bins[..., :-1]) / 2.0
            upper = torch.cat([centers, bins[..., -1:]], -1)
            lower = torch.cat([bins[..., :1], centers], -1)
            bins = lower + (upper - lower) * t_rand
        s_near = self.spacing_fn(ray_bundle.nears.clone())
        s_far = self.spacing_fn(ray_bundle.fars.clone())

        def to_euclid(x):
            return self.spacing_fn_inv(x * s_far + (1 - x) * s_near)

        e = to_euclid(bins)
        return ray_bundle.get_ray_samples(e[..., :-1, None], e[..., 1:, None], bins[..., :-1, None],
                                          bins[..., 1:, None], to_euclid)


@SAMPLERS.register_module()
class UniformSampler(SpacedSampler):
    def __init__(self, num_samples=None, train_stratified=True, single_jitter=False):
        super().__init__(lambda x: x, lambda x: x, num_samples, train_stratified, single_jitter)


@SAMPLERS.register_module()
class PDFSampler(Sampler):
    def __init__(self, num_samples=None, train_stratified=True, single_jitter=False,
                 include_original=True, histogram_padding=0.01):
        super().__init__(num_samples=num_samples)
        self.train_stratified, self.single_jitter = train_stratified, single_jitter
        self.include_original, self.histogram_padding = include_original, histogram_padding

    def generate_ray_samples(self, ray_bundle, ray_samples, weights, num_samples=None, eps=1e-5):
        n = num_samples or self.num_samples
        nb = n + 1
        w = weights[..., 0]
        w_sum = torch.sum(w, dim=-1, keepdim=True)
        pad = torch.relu(eps - w_sum)  # rays with ~zero weight get a flat pdf
        w = w + pad / w.shape[-1]
        w_sum = w_sum + pad
        pdf = w / w_sum
        cdf = torch.min(torch.ones_like(pdf), torch.cumsum(pdf, dim=-1))
        cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
        u = torch.linspace(0.0, 1.0 - 1.0 / nb, steps=nb, device=cdf.device)
        if self.train_stratified and self.training:
            shape = (*cdf.shape[:-1], 1 if self.single_jitter else nb)
            u = u.expand(*cdf.shape[:-1], nb) + self.rand(shape, device=cdf.device) / nb
        else:
            u = (u + 1.0 / (2 * nb)).expand(*cdf.shape[:-1], nb)
        u = u.contiguous()
        assert ray_samples.spacing_starts is not None and ray_samples.spacing_ends is not None
        assert ray_samples.spacing_to_euclidean_fn is not None
        edges = torch.cat([ray_samples.spacing_starts[..., 0],
                           ray_samples.spacing_ends[..., -1:, 0]], dim=-1)
        idx = torch.searchsorted(cdf, u, right=True)
        below = torch.clamp(idx - 1, 0, edges.shape[-1] - 1)
        above = torch.clamp(idx, 0, edges.shape[-1] - 1)
        cdf0, cdf1 = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
        e0, e1 = torch.gather(edges, -1, below), torch.gather(edges, -1, above)
        denom = cdf1 - cdf0
        denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
        t = torch.clip((u - cdf0) / denom, 0, 1)
        bins = (e0 + t * (e1 - e0)).detach()
        fn = ray_samples.spacing_to_euclidean_fn
        e = fn(bins)
        return ray_bundle.get_ray_samples(e[..., :-1, None], e[..., 1:, None], bins[..., :-1, None],
                                          bins[..., 1:, None], fn)


_INITIAL_SAMPLERS = {"UniformSampler": UniformSampler}


@SAMPLERS.register_module()
class NeuSSampler(Sampler):
    def __init__(self, initial_sampler, num_samples, num_samples_importance, num_upsample_steps,
                 base_variance=64.0, train_stratified=True, single_jitter=True):
        super().__init__()
        self.num_samples = num_samples
        self.num_samples_importance = num_samples_importance
        self.num_upsample_steps = num_upsample_steps
        self.base_variance = base_variance
        if initial_sampler not in _INITIAL_SAMPLERS:
            raise KeyError(f"initial_sampler {initial_sampler!r} is not on the pre-training path")
        self.initial_sampler = _INITIAL_SAMPLERS[initial_sampler](
            num_samples=num_samples, train_stratified=train_stratified, single_jitter=single_jitter)
        self.pdf_sampler = PDFSampler(train_stratified=train_stratified, single_jitter=single_jitter)

    def generate_ray_samples(self, ray_bundle, sdf_fn, **kwargs):
        ray_samples = self.initial_sampler(ray_bundle)
        new_samples, sorted_index, sdf = ray_samples, None, None
        out = {}
        for it in range(self.num_upsample_steps):
            with torch.no_grad():
                new_points = new_samples.frustums.get_start_positions()
                new_sdf = sdf_fn(new_points)[0]
            if sorted_index is None:
                sdf = new_sdf
            else:
                merged = torch.cat([sdf.squeeze(-1), new_sdf.squeeze(-1)], -1)
                sdf = torch.gather(merged, 1, sorted_index).unsqueeze(-1)
            alphas = self.rendering_sdf_with_fixed_inv_s(ray_samples, sdf.squeeze(-1),
                                                         inv_s=self.base_variance * 2 ** it)
            weights, _ = alphas_to_weights(alphas.unsqueeze(-1))
            weights = torch.cat((weights, torch.zeros_like(weights[:, :1])), dim=1)
            if it == 0:
                out.update(init_sampled_points=new_points, init_weights=weights)
            new_samples = self.pdf_sampler(
                ray_bundle, ray_samples, weights,
                num_samples=self.num_samples_importance // self.num_upsample_steps)
            pts = new_samples.frustums.get_start_positions()
            out["new_sampled_points"] = pts if "new_sampled_points" not in out else torch.cat(
                [out["new_sampled_points"], pts], dim=1)
            ray_samples, sorted_index = ray_bundle.merge_ray_samples(ray_samples, new_samples)
        out["ray_samples"] = ray_samples
        return out

    @staticmethod
    def rendering_sdf_with_fixed_inv_s(ray_samples, sdf, inv_s):
        """NeuS section alphas with a fixed sharpness: slope = min(this, previous) section slope,
        clipped to [-1e3, 0]."""
        prev_sdf, next_sdf = sdf[:, :-1], sdf[:, 1:]
        dist = ray_samples.deltas[:, :-1, 0]
        mid = (prev_sdf + next_sdf) * 0.5
        cos = (next_sdf - prev_sdf) / (dist + 1e-5)
        prev_cos = torch.cat([torch.zeros_like(cos[:, :1]), cos[:, :-1]], dim=-1)
        cos = torch.minimum(prev_cos, cos).clip(-1e3, 0.0)
        prev_cdf = torch.sigmoid((mid - cos * dist * 0.5) * inv_s)
        next_cdf = torch.sigmoid((mid + cos * dist * 0.5) * inv_s)
        return (prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)
