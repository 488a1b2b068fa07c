"""PonderOutdoor-v2: lidar point-cloud pre-training by differentiable depth rendering (nuScenes).

Mirror of ponder/models/ponder/ponder_outdoor_base.py (PonderOutdoor :18-265).  Same constructor
arguments, registry name, parameter names and ``forward(data_dict) -> dict(loss=..., ...)``
contract.  Differences from the indoor model that matter for the kernels: rays come from the
dataset (``ray_start``/``ray_end``/``ray_offset``: lidar returns that project into the cameras,
datasets/lidar.py), the scene box is fixed per dataset (``scene_bbox``) so no ``to_unit_cube``,
the dense grid is a flat 180 x 180 x 5 slab, the projection is one Conv3d block
(``SimpleConv3D-v1m1``) and the only loss is depth.

GPU organisation (vs the reference's per-scene Python loops):
  * ``to_dense`` (:176-209) is one scatter-mean launch straight into a channels-last
    (B,Z,Y,X,C) grid - the layout both MIOpen's NDHWC conv and the trilinear sampler want;
  * ``render_func`` (:217-252): when every scene of the batch carries the same number of rays
    (the normal case: ``RaySample(point_nsample=512)`` x 6 cameras) all scenes are rendered in one
    batched pass; ragged batches fall back to one pass per scene;
  * block masking (:93-137) ranks all scenes' blocks in one device pass (masking.py).
"""
from collections.abc import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from ponderv2_amd import precision
from ponderv2_amd.torch_scatter import scatter
from ..builder import MODELS, build_model
from ..utils import offset2batch
from .masking import mask_blocks
from .ponder_indoor_base import PonderIndoor, stub_text_embeddings
from .render_utils import RayBundle, build_renderer


def _per_condition(value):
    """The reference's defaults are written ``((a, b, c))`` - a flat tuple, not a tuple of tuples
    (ponder_outdoor_base.py:26-28); accept both spellings."""
    if isinstance(value, Sequence) and len(value) and not isinstance(value[0], Sequence):
        return (tuple(value),)
    return tuple(tuple(v) for v in value)


@MODELS.register_module("PonderOutdoor-v2")
class PonderOutdoor(nn.Module):
    def __init__(self, backbone, projection, renderer, mask=None,
                 scene_bbox=((-54.0, -54.0, -5.0, 54.0, 54.0, 3.0),),
                 grid_shape=((180, 180, 5),), grid_size=((0.6, 0.6, 1.6),), val_ray_split=8192,
                 pool_type="mean", share_volume=True, render_semantic=False, conditions=None,
                 template=None, clip_model=None, class_name=None, valid_index=None,
                 dense_channels_last=True, proj_autocast=None, batched_render=True,
                 sparse_dense_input=True):
        super().__init__()
        self.grid_shape = _per_condition(grid_shape)
        self.grid_size = _per_condition(grid_size)
        self.scene_bbox = _per_condition(scene_bbox)
        self.pool_type, self.val_ray_split = pool_type, val_ray_split
        self.share_volume, self.mask = share_volume, mask
        self.dense_channels_last = dense_channels_last
        self.proj_autocast = proj_autocast
        self.batched_render = batched_render
        self.sparse_dense_input = sparse_dense_input  # first conv from occupied cells only
        if mask is not None:
            p = nn.Parameter(torch.zeros(1, mask.channel))
            nn.init.trunc_normal_(p, mean=0, std=0.02, a=-0.02, b=0.02)
            self.register_parameter("mtoken", p)
        self.backbone = build_model(backbone)
        self.proj_net = build_model(projection)
        if dense_channels_last:
            self.proj_net = self.proj_net.to(memory_format=torch.channels_last_3d)
        self.renderer = build_renderer(renderer)
        self.render_semantic = render_semantic
        self.conditions, self.valid_index = conditions, valid_index
        if render_semantic:
            self.load_semantic(template, clip_model, class_name)

    # ------------------------------------------------------------------ language targets
    def load_semantic(self, template, clip_model, class_name):
        try:  # real CLIP text encoder when the package and its weights are present (:62-91)
            import clip  # noqa: F401

            embedding, _ = PonderIndoor._clip_text_embeddings(template, clip_model, class_name)
        except Exception:
            embedding = stub_text_embeddings(len(class_name))
        self.register_buffer("class_embedding", embedding.float().cpu())

    _valid_embedding = PonderIndoor._valid_embedding

    def _condition_index(self, data_dict):
        condition = data_dict["condition"][0]
        assert condition in self.conditions
        return self.conditions.index(condition)

    def _const(self, name, idx, device, dtype):
        """Per-condition constants (scene box, grid size) as device tensors, uploaded once."""
        cache = self.__dict__.setdefault("_const_cache", {})
        key = (name, idx, str(device), dtype)
        if key not in cache:
            cache[key] = torch.tensor(getattr(self, name)[idx], dtype=dtype, device=device)
        return cache[key]

    # ------------------------------------------------------------------ backbone
    def extract_feature(self, data_dict):
        if self.mask is not None:
            data_dict["feat"] = mask_blocks(
                data_dict["grid_coord"], data_dict["feat"], data_dict["offset"], self.mask.size,
                self.mask.ratio, self.mtoken, rand=data_dict.get("mask_rand"))
        # the ambient reduced precision (enable_amp) reaches the sparse U-Net as 16-bit feature
        # matrices between its layers (ponderv2_amd/precision.py); what comes out is fp32 again
        with precision.sparse_activations(getattr(self, "_ambient_amp", None)):
            data_dict["sparse_backbone_feat"] = self.backbone(data_dict).float()
        return data_dict

    # ------------------------------------------------------------------ rays (no grad)
    @torch.no_grad()
    def prepare_ray(self, data_dict):
        """Normalise the dataset's rays into the [0,1]^3 scene box (:139-174)."""
        idx = self._condition_index(data_dict)
        start, end = data_dict["ray_start"], data_dict["ray_end"]
        box = self._const("scene_bbox", idx, start.device, start.dtype)
        lo, extent = box[:3], box[3:] - box[:3]
        ray_start, ray_end = (start - lo) / extent, (end - lo) / extent
        delta = ray_end - ray_start
        ray_dict = dict(ray_offset=data_dict["ray_offset"], ray_o=ray_start,
                        ray_d=F.normalize(delta, dim=-1),
                        depth=torch.linalg.norm(delta, dim=-1, keepdim=True))
        if "ray_offset_host" in data_dict:
            ray_dict["ray_offset_host"] = data_dict["ray_offset_host"]
        if "ray_color" in data_dict:
            ray_dict["rgb"] = data_dict["ray_color"]
        if "ray_segment" in data_dict and self.render_semantic:
            assert len(set(data_dict["condition"])) == 1, "assume same condition in one batch"
            seg = data_dict["ray_segment"]
            table = self._valid_embedding(idx, seg.device)
            ray_dict["semantic"] = table[seg.long()]
            ray_dict["segment"] = seg
        return ray_dict

    # ------------------------------------------------------------------ dense volume
    def _dense_rows(self, data_dict):
        coord, offset = data_dict["coord"], data_dict["offset"]
        idx = self._condition_index(data_dict)
        box = self._const("scene_bbox", idx, coord.device, coord.dtype)
        size = self._const("grid_size", idx, coord.device, coord.dtype)
        G0, G1, G2 = self.grid_shape[idx]
        batch = offset2batch(offset, coord.shape[0])
        g = ((coord - box[:3]) / size).long()
        if self.dense_channels_last:
            lin = (g[:, 2] * G1 + g[:, 1]) * G0 + g[:, 0]  # memory order (Z,Y,X), channels last
        else:
            lin = (g[:, 0] * G1 + g[:, 1]) * G2 + g[:, 2]  # the reference's (X,Y,Z) order
        return lin + batch * (G0 * G1 * G2)

    def to_dense(self, data_dict):
        """Scatter-mean the backbone features into the fixed scene grid -> (B, C, Z, Y, X)."""
        feat, offset = data_dict["sparse_backbone_feat"], data_dict["offset"]
        assert len(data_dict["coord"]) == len(feat)
        G0, G1, G2 = self.grid_shape[self._condition_index(data_dict)]
        B, C = offset.numel(), feat.shape[1]
        lin = self._dense_rows(data_dict)
        grid = feat.new_zeros((B * G0 * G1 * G2, C))
        grid = scatter(feat, lin[:, None], dim=0, reduce=self.pool_type, out=grid)
        if self.dense_channels_last:
            return grid.view(B, G2, G1, G0, C).permute(0, 4, 1, 2, 3)  # channels_last_3d view
        return grid.view(B, G0, G1, G2, C).permute(0, 4, 3, 2, 1).contiguous()

    def prepare_volume(self, data_dict):
        if (self.sparse_dense_input and self.dense_channels_last and self.pool_type == "mean"
                and hasattr(self.proj_net, "forward_cells")
                and getattr(self.proj_net, "cells_supported", lambda: True)()):
            from .sparse_input import cells_from_voxels

            G0, G1, G2 = self.grid_shape[self._condition_index(data_dict)]
            dense = cells_from_voxels(data_dict["sparse_backbone_feat"], self._dense_rows(data_dict),
                                      data_dict["offset"].numel(), (G2, G1, G0))
            project = self.proj_net.forward_cells
        else:
            dense = self.to_dense(data_dict)
            project = self.proj_net
        amp_dtype = self._projection_dtype(data_dict["coord"].device)
        if amp_dtype is not None:
            with torch.autocast(data_dict["coord"].device.type, dtype=amp_dtype):
                volume = project(dense)
            volume = volume.float()
        else:
            volume = project(dense)
        if self.dense_channels_last:
            volume = volume.contiguous(memory_format=torch.channels_last_3d)
        return [volume]

    # ------------------------------------------------------------------ rendering + losses
    @staticmethod
    def _ray_edges(ray_dict):
        """Cumulative ray counts as a Python list (from the collate function when it kept them on
        the host, else one device->host read)."""
        oh = ray_dict.get("ray_offset_host")
        if oh is None:
            oh = [int(v) for v in ray_dict["ray_offset"].tolist()]
            ray_dict["ray_offset_host"] = oh
        return [0] + list(oh)

    def _uniform_rays(self, ray_dict):
        edges = self._ray_edges(ray_dict)
        counts = {b - a for a, b in zip(edges[:-1], edges[1:])}
        return len(counts) == 1 and next(iter(counts)) > 0

    def render_func(self, ray_dict, volume_feature):
        edges = self._ray_edges(ray_dict)
        B = len(edges) - 1
        if self.training and self.batched_render and self._uniform_rays(ray_dict):
            bundle = RayBundle(origins=ray_dict["ray_o"], directions=ray_dict["ray_d"], num_scenes=B)
            return self.renderer(bundle, volume_feature)
        outs = []
        for i in range(B):
            vols = [v[i] for v in volume_feature]
            o, d = ray_dict["ray_o"][edges[i]:edges[i + 1]], ray_dict["ray_d"][edges[i]:edges[i + 1]]
            if self.training:
                outs.append(self.renderer(RayBundle(origins=o, directions=d), vols))
            else:
                parts = [self.renderer(RayBundle(origins=po, directions=pd), vols)
                         for po, pd in zip(o.split(self.val_ray_split), d.split(self.val_ray_split))]
                outs.append({k: torch.cat([p[k].detach() for p in parts], 0) for k in parts[0]})
        return {k: torch.cat([o[k] for o in outs], dim=0) for k in outs[0]}

    def render_loss(self, render_out, ray_dict):
        loss_dict = self.renderer.get_loss(render_out, ray_dict)
        return sum(v for k, v in loss_dict.items() if "loss" in k), loss_dict

    def _projection_dtype(self, device):
        """Reduced precision of the dense projection network: ``proj_autocast`` (a dtype name) or
        the ambient autocast dtype the trainer entered (``enable_amp=True``); None = fp32."""
        if self.proj_autocast is not None and device.type == "cuda":
            return getattr(torch, self.proj_autocast)
        return getattr(self, "_ambient_amp", None)

    def prefetch(self, data_dict):
        """Input-pipeline hook: launch the sparse backbone's geometry for this (device-resident)
        batch on the side stream (SpUNet.prefetch_geometry); trainers call it one batch ahead."""
        fn = getattr(self.backbone, "prefetch_geometry", None)
        return fn(data_dict) if fn is not None else data_dict

    def forward(self, data_dict):
        """Under an ambient autocast region (the reference's ``enable_amp=True``) the reduced
        precision is SCOPED to the dense projection network - the one part of the path that runs
        on library convolutions: the sparse backbone, the ray march and the losses are fp32
        hand-written kernels and small torch ops that autocast would only wrap in casts (measured:
        whole-model autocast is host-bound at 47 ms per step, the scoped form runs 35 ms)."""
        dev_type = data_dict["coord"].device.type
        self._ambient_amp = (torch.get_autocast_dtype(dev_type)
                             if torch.is_autocast_enabled(dev_type) else None)
        with torch.autocast(dev_type, enabled=False):
            return self._forward(data_dict)

    def _forward(self, data_dict):
        data_dict = self.extract_feature(data_dict)
        ray_dict = self.prepare_ray(data_dict)
        volume_feature = self.prepare_volume(data_dict)
        render_out = self.render_func(ray_dict, volume_feature)
        loss, loss_dict = self.render_loss(render_out, ray_dict)
        return dict(loss=loss, **loss_dict)
