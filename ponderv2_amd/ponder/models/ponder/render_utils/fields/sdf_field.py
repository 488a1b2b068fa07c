"""SDF field of the NeuS head: trilinear feature lookup in the projected volume, SDF / colour /
semantic MLPs, analytic-by-autograd SDF gradient, NeuS alpha.

Restates ponder/models/ponder/render_utils/fields/sdf_field.py: LaplaceDensity :10-35,
SingleVarianceNetwork :38-55, normalize_3d_coordinate :58-74, SDFField :77-284 (get_alpha
:122-146, feature_sampling :148-183, get_sdf :185-197, forward :211-284).  The feature lookup is
``ponderv2_amd.smooth_sampler.SmoothSampler`` (csrc/trilinear.hip), twice differentiable.

Reference quirk kept on purpose: ``get_sdf`` as used by the sampler's coarse pass
receives UN-normalised points, whereas ``forward`` normalises them first when ``norm_pts``.
"""
import torch
import torch.nn.functional as F
from torch import nn

from ponderv2_amd.smooth_sampler import SmoothSampler
from ..builder import FIELDS
from ..decoders import RGBDecoder, SDFDecoder, SemanticDecoder


class LaplaceDensity(nn.Module):
    """alpha * Laplace(0, beta).cdf(-sdf) (VolSDF); present for checkpoint compatibility."""

    def __init__(self, init_val, beta_min=0.0001):
        super().__init__()
        self.register_parameter("beta_min", nn.Parameter(beta_min * torch.ones(1), requires_grad=False))
        self.register_parameter("beta", nn.Parameter(init_val * torch.ones(1), requires_grad=True))

    def get_beta(self):
        return self.beta.abs() + self.beta_min

    def forward(self, sdf, beta=None):
        beta = self.get_beta() if beta is None else beta
        return (1.0 / beta) * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))


class SingleVarianceNetwork(nn.Module):
    def __init__(self, init_val):
        super().__init__()
        self.register_parameter("variance", nn.Parameter(init_val * torch.ones(1), requires_grad=True))

    def forward(self, x):
        return torch.ones([len(x), 1], device=x.device) * torch.exp(self.variance * 10.0)

    def get_variance(self):
        return torch.exp(self.variance * 10.0).clip(1e-6, 1e6)


def normalize_3d_coordinate(p, padding=0.1):
    """[-0.5-pad/2, 0.5+pad/2] -> [0, 1); outliers are pushed to the [0, 1-1e-3] ends (same values
    as the reference's data-dependent branches, without the two host syncs)."""
    p_nor = p / (1 + padding + 10e-4) + 0.5
    p_nor = torch.where(p_nor >= 1, torch.full_like(p_nor, 1 - 10e-4), p_nor)
    return torch.where(p_nor < 0, torch.zeros_like(p_nor), p_nor)


@FIELDS.register_module()
class SDFField(nn.Module):
    def __init__(self, sdf_decoder, beta_init, use_gradient=True, volume_type="default",
                 padding_mode="zeros", share_volume=True, rgb_decoder=None, semantic_decoder=None,
                 norm_pts=False, norm_padding=0.1, hoist_semantic=True):
        super().__init__()
        if volume_type != "default":
            raise NotImplementedError(f"volume_type={volume_type!r}")
        self.beta_init, self.volume_type = beta_init, volume_type
        self.padding_mode, self.share_volume = padding_mode, share_volume
        self.sdf_decoder = SDFDecoder(**sdf_decoder)
        self.rgb_decoder = RGBDecoder(**rgb_decoder) if rgb_decoder is not None else None
        self.semantic_decoder = (SemanticDecoder(**semantic_decoder)
                                 if semantic_decoder is not None else None)
        self.use_gradient = use_gradient
        self.laplace_density = LaplaceDensity(init_val=beta_init)
        self.deviation_network = SingleVarianceNetwork(init_val=beta_init)
        self._cos_anneal_ratio = 1.0
        self.norm_pts, self.norm_padding = norm_pts, norm_padding
        # composite the semantic head before its last (linear) layer - exact up to fp
        # re-association; False reproduces the reference's per-sample evaluation order
        self.hoist_semantic = hoist_semantic

    def set_cos_anneal_ratio(self, anneal):
        self._cos_anneal_ratio = anneal

    def get_alpha(self, ray_samples, sdf, gradients):
        inv_s = self.deviation_network.get_variance()
        true_cos = (ray_samples.frustums.directions * gradients).sum(-1, keepdim=True)
        r = self._cos_anneal_ratio
        iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - r) + F.relu(-true_cos) * r)
        half = iter_cos * ray_samples.deltas * 0.5
        prev_cdf = torch.sigmoid((sdf - half) * inv_s)
        next_cdf = torch.sigmoid((sdf + half) * inv_s)
        return ((prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)).clip(0.0, 1.0)

    def feature_sampling(self, pts_norm, volume_feature):
        """pts_norm (R,S,3) in [0,1]; volume_feature list of (C,Z,Y,X) -> (R,S,L*C) with the first
        halves of every level's channels first."""
        feats = []
        for vol in volume_feature:
            if vol.dim() == 5:  # (B,C,Z,Y,X): rays are scene-major, R/B rays per scene
                B = vol.shape[0]
                grid = (pts_norm * 2 - 1).reshape(B, 1, pts_norm.shape[0] // B, *pts_norm.shape[1:])
                out = SmoothSampler.apply(vol.to(pts_norm.dtype), grid, self.padding_mode, True,
                                          False)  # (B,C,1,R/B,S)
                feats.append(out.squeeze(2).permute(0, 2, 3, 1).reshape(*pts_norm.shape[:2], -1))
            else:
                grid = (pts_norm * 2 - 1)[None, None]  # (1,1,R,S,3)
                out = SmoothSampler.apply(vol.unsqueeze(0).to(pts_norm.dtype), grid,
                                          self.padding_mode, True, False)  # (1,C,1,R,S)
                feats.append(out.squeeze(0).squeeze(1).permute(1, 2, 0))
        if len(feats) == 1:
            return feats[0]  # cat([f[:h], f[h:]]) of a single level is the identity
        ret = torch.stack(feats, dim=-2)
        h = ret.shape[-1] // 2
        return torch.cat([ret[..., :h].flatten(-2, -1), ret[..., h:].flatten(-2, -1)], dim=-1)

    def _split(self, point_features, which):
        return point_features if self.share_volume else torch.chunk(point_features, 2, dim=-1)[which]

    def get_sdf(self, points, volume_feature):
        point_features = self.feature_sampling(points, volume_feature)
        h = self.sdf_decoder(points, self._split(point_features, 0))
        return h[..., :1], h[..., 1:], point_features

    def get_density(self, ray_samples, volume_feature):
        return self.laplace_density(self.get_sdf(ray_samples.frustums.get_start_positions(),
                                                 volume_feature)[0])

    def get_occupancy(self, sdf):
        return torch.sigmoid(-10.0 * sdf)

    def forward(self, ray_samples, volume_feature, return_alphas=False):
        points = ray_samples.frustums.get_start_positions()
        if self.norm_pts:
            points = normalize_3d_coordinate(points, self.norm_padding)
        points.requires_grad_(True)
        with torch.enable_grad():
            sdf, geo_features, point_features = self.get_sdf(points, volume_feature)
        gradients = torch.autograd.grad(sdf, points, torch.ones_like(sdf), create_graph=True,
                                        retain_graph=True, only_inputs=True)[0]
        directions = ray_samples.frustums.directions
        cond = ([gradients] if self.use_gradient else []) + [self._split(point_features, 1),
                                                             geo_features]
        outputs = {}
        if self.rgb_decoder is not None:
            outputs["rgb"] = self.rgb_decoder(points, torch.cat(cond + [directions], dim=-1))
        if self.semantic_decoder is not None:
            if self.hoist_semantic and self.semantic_decoder.out_activation is None:
                outputs["semantic_hidden"] = self.semantic_decoder.hidden(points, torch.cat(cond, dim=-1))
            else:
                outputs["semantic"] = self.semantic_decoder(points, torch.cat(cond, dim=-1))
        outputs.update(density=self.laplace_density(sdf), sdf=sdf, gradients=gradients,
                       normal=F.normalize(gradients, dim=-1))
        if return_alphas:
            outputs["alphas"] = self.get_alpha(ray_samples, sdf, gradients)
        return outputs
