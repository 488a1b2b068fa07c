"""Ray containers of the NeuS head: bundle (origins, directions, near/far), frustums (per-sample
start/end distances) and samples (with the alpha -> weight compositing rule).

Restates ponder/models/ponder/render_utils/rays.py: Frustums :5-34, RaySamples :37-105
(weights from alphas :83-105), RayBundle :108-227 (merge :118-153, get_ray_samples :190-227).
Plain Python objects here (the reference makes them nn.Modules without parameters).
"""
import torch

from ponderv2_amd import raymarch


_CONST_CACHE = {}


def device_constant(values, device, dtype=torch.float32):
    """A small constant tensor on `device`, uploaded once and then reused: host->device copies
    inside the render head would both cost a launch every step and make the region impossible to
    capture into a hipGraph."""
    key = (tuple(float(v) for v in values), str(device), dtype)
    t = _CONST_CACHE.get(key)
    if t is None:
        t = torch.tensor(list(values), dtype=dtype).to(device)
        _CONST_CACHE[key] = t
    return t


def device_linspace(start, end, steps, device):
    """torch.linspace evaluated on the host (bit-identical to the reference's CPU linspace,
    ray_samplers.py:70-74) and cached on the device."""
    key = ("linspace", float(start), float(end), int(steps), str(device))
    t = _CONST_CACHE.get(key)
    if t is None:
        t = torch.linspace(start, end, steps).to(device)
        _CONST_CACHE[key] = t
    return t


class Frustums:
    def __init__(self, origins, directions, starts, ends):
        self.origins, self.directions, self.starts, self.ends = origins, directions, starts, ends

    def get_positions(self):
        return self.origins + self.directions * (self.starts + self.ends) / 2

    def get_start_positions(self):
        return self.origins + self.directions * self.starts


class _CumprodNoZero(torch.autograd.Function):
    """torch.cumprod along dim 1 for inputs that are never zero (here 1 - alpha + 1e-7 >= 1e-7).
    ATen's cumprod backward first asks the device whether the input contains zeros (a host sync,
    which also makes the region impossible to capture into a hipGraph); for non-zero inputs it then
    uses exactly this formula: dx_i = (sum_{j>=i} g_j y_j) / x_i."""

    @staticmethod
    def forward(ctx, x):
        y = torch.cumprod(x, dim=1)
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        w = g * y
        rev = torch.flip(torch.cumsum(torch.flip(w, dims=[1]), dim=1), dims=[1])
        return rev / x


def alphas_to_weights(alphas):
    """w_k = alpha_k * prod_{j<k} (1 - alpha_j + 1e-7); also returns the (S+1) transmittance."""
    if raymarch.ENABLED and raymarch.supported(alphas):
        return raymarch.composite_weights(alphas)  # one launch each way (csrc/raymarch.hip)
    ones = torch.ones((alphas.shape[0], 1, 1), device=alphas.device, dtype=alphas.dtype)
    transmittance = _CumprodNoZero.apply(torch.cat([ones, 1.0 - alphas + 1e-7], dim=1))
    return alphas * transmittance[:, :-1, :], transmittance


class RaySamples:
    def __init__(self, frustums, deltas, spacing_starts=None, spacing_ends=None,
                 spacing_to_euclidean_fn=None, num_scenes=1):
        self.num_scenes = num_scenes  # rays are scene-major with equal counts per scene
        self.frustums = frustums
        self.deltas = deltas
        self.spacing_starts, self.spacing_ends = spacing_starts, spacing_ends
        self.spacing_to_euclidean_fn = spacing_to_euclidean_fn

    def get_weights_and_transmittance_from_alphas(self, alphas):
        return alphas_to_weights(alphas)

    def get_weights_and_transmittance(self, densities):
        dd = self.deltas * densities
        acc = torch.cumsum(dd[..., :-1, :], dim=-2)
        acc = torch.cat([torch.zeros_like(acc[..., :1, :]), acc], dim=-2)
        transmittance = torch.exp(-acc)
        return (1 - torch.exp(-dd)) * transmittance, transmittance


class RayBundle:
    def __init__(self, origins, directions, nears=None, fars=None, num_scenes=1):
        self.origins, self.directions = origins, directions  # (R,3)
        self.nears, self.fars = nears, fars                  # (R,1)
        # >1: the bundle holds the rays of several scenes (scene-major, equal counts); the field
        # then samples a batched (B,C,Z,Y,X) volume in ONE launch instead of looping over scenes
        self.num_scenes = num_scenes

    def get_ray_samples(self, bin_starts, bin_ends, spacing_starts, spacing_ends,
                        spacing_to_euclidean_fn):
        deltas = bin_ends - bin_starts
        shape = [*deltas.shape[:-1], -1]
        frustums = Frustums(self.origins[..., None, :].expand(shape),
                            self.directions[..., None, :].expand(shape), bin_starts, bin_ends)
        return RaySamples(frustums, deltas, spacing_starts, spacing_ends, spacing_to_euclidean_fn,
                          num_scenes=self.num_scenes)

    def merge_ray_samples(self, samples_a, samples_b):
        """Union of two sample sets sorted by spacing start; returns (samples, sorted_index) where
        sorted_index gathers per-sample values from cat([a, b])."""
        starts = torch.cat([samples_a.spacing_starts[..., 0], samples_b.spacing_starts[..., 0]], -1)
        last = torch.maximum(samples_a.spacing_ends[..., -1:, 0], samples_b.spacing_ends[..., -1:, 0])
        bins, sorted_index = torch.sort(starts, -1)
        bins = torch.cat([bins, last], dim=-1).detach()
        euclid = samples_a.spacing_to_euclidean_fn(bins)
        merged = self.get_ray_samples(euclid[..., :-1, None], euclid[..., 1:, None],
                                      bins[..., :-1, None], bins[..., 1:, None],
                                      samples_a.spacing_to_euclidean_fn)
        return merged, sorted_index
