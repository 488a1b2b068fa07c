"""PonderIndoor-v2: point-cloud pre-training by differentiable neural rendering (indoor RGB-D).

Mirror of ponder/models/ponder/ponder_indoor_base.py (PonderIndoor :19-706).  Same constructor
arguments, registry name, parameter names and ``forward(data_dict) -> dict(loss=..., ...)``
contract; the data flow is re-organised for the GPU:

  * ``to_unit_cube`` (:344-444), ``ray_sample`` (:499-620) and ``to_dense`` (:177-342) run as
    batched device ops over all scenes / views at once - no Python loop per view, no
    ``.cpu().numpy()`` round trips; rays are generated only for the sampled pixels instead of for
    the whole 480x640 image;
  * the sparse->dense scatter-mean writes straight into a channels-last (B,Z,Y,X,C) grid with one
    launch for the whole batch (csrc/dense_scatter.hip), which is also the layout the trilinear
    sampler wants for the projected volume;
  * pixel choice is injectable (``data_dict["ray_pixels"]``) because the reference draws it with
    the CPU generator (:548-551).
Reference quirks preserved: class id 0 gets a zero semantic target (:594-596, Q3); ``ppt_loss`` is
reported but not added to ``loss`` (:699-704, Q10).
"""
import math
from collections.abc import Sequence

import inspect

import torch
import torch.nn as nn
import torch.nn.functional as F

from ponderv2_amd import precision, ray_setup
from ponderv2_amd.linear import linear

from ponderv2_amd.torch_scatter import scatter
from ..builder import MODELS, build_model
from ..losses import build_criteria
from ..utils import offset2batch, offsets_host
from .masking import mask_blocks
from .render_utils import RayBundle, build_renderer
from .render_utils.rays import device_constant

import os

# the ray set-up rides with the batch's staging (PonderIndoor.prefetch); PV2_PREFETCH_RAYS=0: inside the step
PREFETCH_RAYS = os.environ.get("PV2_PREFETCH_RAYS", "1") != "0"


def stub_text_embeddings(num_classes, dim=512, seed=0):
    """Deterministic stand-in for CLIP text embeddings (no network / weights in this environment):
    unit vectors from a seeded generator.  Same role as load_semantic (:85-118)."""
    g = torch.Generator().manual_seed(seed)
    e = torch.randn(num_classes, dim, generator=g)
    return e / e.norm(dim=-1, keepdim=True)


def _inv(a):
    """Inverse of a batch of camera / scene transforms (..., n, n), n <= 4.  On the device: ONE launch
    (``pv2_small_inverse``: Gauss-Jordan in double precision per matrix) instead of the library's
    11 (rocSOLVER getrf + getri behind torch.linalg.inv_ex, four calls per step).  Elsewhere
    torch.linalg.inv without its error check - that check reads a status word back, i.e. stalls the
    host once per call (camera matrices are invertible by construction)."""
    n = a.shape[-1]
    if a.is_cuda and a.dtype == torch.float32 and n <= 4 and a.shape[-2] == n and not a.requires_grad:
        from ponderv2_amd import _lib
        from ponderv2_amd.kernels import _ptr, _stream

        src = a.contiguous()
        out = torch.empty_like(src)
        _lib.check(_lib.lib().pv2_small_inverse(_ptr(src), src.numel() // (n * n), n, _ptr(out),
                                                _stream(src)), "pv2_small_inverse")
        return out
    return torch.linalg.inv_ex(a, check_errors=False).inverse


@MODELS.register_module("PonderIndoor-v2")
class PonderIndoor(nn.Module):
    def __init__(self, backbone, projection, renderer, mask=None, grid_shape=64, grid_size=0.02,
                 val_ray_split=10240, ray_nsample=128, padding=0.1, backbone_out_channels=96,
                 context_channels=256, pool_type="mean", render_semantic=False, conditions=None,
                 template=None, clip_model=None, class_name=None, valid_index=None,
                 ppt_loss_weight=1.0, ppt_criteria=None, dense_channels_last=True,
                 proj_autocast=None, batched_render=True,
                 sparse_dense_input=True):
        super().__init__()
        self.grid_shape = tuple(grid_shape) if isinstance(grid_shape, Sequence) else (grid_shape,) * 3
        self.grid_size, self.pool_type = grid_size, pool_type
        self.val_ray_split, self.ray_nsample = val_ray_split, ray_nsample
        self.mask = mask
        self.dense_channels_last = dense_channels_last
        # dtype name ("bfloat16"/"float16") to run ONLY the dense projection U-Net under autocast,
        # as the reference does for the whole model with enable_amp=True; None = fp32 (parity mode)
        self.proj_autocast = proj_autocast
        self.batched_render = batched_render
        # evaluate the projection network's first conv from the occupied cells only (the dense
        # 96-channel grid is never built, sparse_input.py); False = the reference's dense path
        self.sparse_dense_input = sparse_dense_input
        h = 0.5 + padding / 2
        self.bounds = [[-h, -h, -h], [h, h, h]]
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
        self.embedding_table = nn.Embedding(len(conditions), context_channels)
        self.backbone_out_channels = backbone_out_channels
        self.ppt_loss_weight = ppt_loss_weight if render_semantic else 0.0
        if render_semantic:
            self.load_semantic(template, clip_model, class_name)
        if self.ppt_loss_weight > 0:
            assert ppt_criteria is not None, "Please provide PPT's loss function."
            self.ppt_criteria = build_criteria(ppt_criteria)

    # ------------------------------------------------------------------ language targets
    def load_semantic(self, template, clip_model, class_name):
        embedding = None
        try:  # real CLIP text encoder when the package and its weights are present
            import clip  # noqa: F401

            embedding, scale = self._clip_text_embeddings(template, clip_model, class_name)
        except Exception:
            embedding, scale = stub_text_embeddings(len(class_name)), math.log(100.0)
        self.register_buffer("class_embedding", embedding.float().cpu())
        self.logit_scale = nn.Parameter(torch.tensor(float(scale)), requires_grad=False)
        if self.ppt_loss_weight > 0:
            self.proj_head = nn.Linear(self.backbone_out_channels, embedding.shape[1])

    @staticmethod
    def _clip_text_embeddings(template, clip_model, class_name):
        import clip

        model, _ = clip.load(clip_model, device="cpu", download_root="./.cache/clip")
        model.requires_grad_(False)
        templates = [template] if isinstance(template, str) else list(template)
        prompts = [t.replace("[x]", n) for n in class_name for t in templates]
        emb = model.encode_text(clip.tokenize(prompts))
        emb = emb / emb.norm(dim=-1, keepdim=True)
        if len(templates) > 1:
            # prompts are class-major; the reference reshapes as (T, K, D) (:101-104), kept as is
            emb = emb.reshape(len(templates), len(class_name), -1).mean(0)
            emb = emb / emb.norm(dim=-1, keepdim=True)
        return emb.float(), float(model.logit_scale)

    def _valid_embedding(self, cond_idx, device):
        """class_embedding rows of one condition's valid classes, sliced once per device (indexing
        with a Python list every step is a host->device copy and a sync)."""
        cache = self.__dict__.setdefault("_valid_emb_cache", {})
        key = (cond_idx, str(device))
        if key not in cache:
            vi = list(self.valid_index[cond_idx])
            cache[key] = self.class_embedding.to(device)[vi, :].contiguous()
        return cache[key]

    def _condition_index(self, data_dict):
        condition = data_dict["condition"][0]
        assert condition in self.conditions
        return self.conditions.index(condition)

    # ------------------------------------------------------------------ backbone
    def extract_feature(self, data_dict):
        if self.mask is not None:
            data_dict["feat"] = self._mask_blocks(data_dict)
        if "condition" in data_dict:
            idx = device_constant((self._condition_index(data_dict),), data_dict["coord"].device,
                                  torch.int64)
            data_dict["context"] = self.embedding_table(idx)
        # the ambient reduced precision (enable_amp) reaches the sparse U-Net as 16-bit feature
        # matrices between its layers (ponderv2_amd/precision.py); what comes out is fp32 again
        with precision.sparse_activations(getattr(self, "_ambient_amp", None)):
            data_dict["sparse_backbone_feat"] = self.backbone(data_dict).float()
        return data_dict

    def _mask_blocks(self, data_dict):
        """Replace the features of a random ``ratio`` of the size^3-voxel blocks by ``mtoken``
        (reference :133-162), per scene, without host loops."""
        return mask_blocks(data_dict["grid_coord"], data_dict["feat"], data_dict["offset"],
                           self.mask.size, self.mask.ratio, self.mtoken,
                           rand=data_dict.get("mask_rand"))

    # ------------------------------------------------------------------ geometry (no grad)
    @torch.no_grad()
    def to_unit_cube(self, data_dict, z_level=-0.5):
        coords = data_dict["coord"]
        offset = data_dict["offset"]
        B = offset.numel()
        batch = offset2batch(offset, coords.shape[0])
        edges = [0] + offsets_host(data_dict)

        def seg_minmax(x):  # per-scene min/max over contiguous row ranges (B is small)
            lo = torch.stack([x[a:b].amin(0) for a, b in zip(edges[:-1], edges[1:])])
            hi = torch.stack([x[a:b].amax(0) for a, b in zip(edges[:-1], edges[1:])])
            return lo - 1e-5, hi + 1e-5

        lo, hi = seg_minmax(coords)
        loc = (lo + hi) / 2
        extent = (hi - lo).max(dim=1).values
        scale = 1.0 / extent
        tmp_z = (coords[:, 2] - loc[batch, 2]) * scale[batch]
        z_min = torch.stack([tmp_z[a:b].amin() for a, b in zip(edges[:-1], edges[1:])])

        eye = torch.eye(4, device=coords.device).expand(B, 4, 4)
        S_loc = eye.clone()
        S_loc[:, :3, 3] = -loc
        S_scale = eye * scale[:, None, None]
        S_scale[:, 3, 3] = 1
        S_loc2 = eye.clone()
        S_loc2[:, 2, 3] = -z_min + z_level
        S = S_loc2 @ S_scale @ S_loc  # (B,4,4)

        # S is a per-scene scale + translation: its 3x3 block is diagonal, so the per-point
        # matrix-vector product (a 46 k-batch bmm of 3x3 blocks, 0.4 ms) is an elementwise
        # multiply-add with the SAME entries - the dropped terms are exact zeros, the result is
        # bit-identical
        diag = torch.diagonal(S[:, :3, :3], dim1=1, dim2=2)
        new = coords * diag[batch] + S[:, :3, 3][batch]
        new = torch.clip(new, min=-0.5 + 1e-5, max=0.5 - 1e-5).float()

        pose = data_dict["extrinsic"].clone().float()  # (B,V,4,4)
        pose[:, :, 3, 3] = 1
        data_dict["extrinsic"] = pose @ _inv(S.float())[:, None]
        data_dict["depth_scale"] = scale * data_dict["depth_scale"]
        data_dict["pc_scale"] = extent.to(data_dict["depth_scale"].dtype)
        lo2, hi2 = seg_minmax(new)
        pc = data_dict["pc_scale"]
        data_dict["bbox"] = (torch.stack([lo2, hi2], dim=1) + 0.5) * pc[:, None, None]
        data_dict["coord"] = (new + 0.5) * pc[batch][:, None]
        return data_dict

    @torch.no_grad()
    def get_mask_at_box(self, ray_o, ray_d):
        """Slab test of rays against the padded unit cube (reference :480-497, numpy there).
        ray_o (...,3) one origin per view, ray_d (...,n,3)."""
        d = ray_d / torch.linalg.norm(ray_d, dim=-1, keepdim=True)
        d = torch.where((d < 1e-5) & (d > -1e-10), torch.full_like(d, 1e-5), d)
        d = torch.where((d > -1e-5) & (d < 1e-10), torch.full_like(d, -1e-5), d)
        inv = (1.0 / d).double()
        lo = device_constant(self.bounds[0], d.device, torch.float64)  # (uploaded once: a host->device
        hi = device_constant(self.bounds[1], d.device, torch.float64)  #  copy per step stalls the host)
        o = ray_o.double()[..., None, :]
        ta, tb = (lo - o) * inv, (hi - o) * inv
        near = torch.minimum(ta, tb).max(dim=-1).values.clamp(min=0.1)
        far = torch.maximum(ta, tb).min(dim=-1).values
        return near < far

    @torch.no_grad()
    def _choose_pixels(self, valid, n):
        """valid (B,V,H,W) bool -> flat pixel ids (B,V,n): a uniform random subset of the valid
        pixels of every view (device-side replacement of randperm on the host)."""
        B, V, H, W = valid.shape
        key = torch.rand((B, V, H * W), device=valid.device)
        key = torch.where(valid.reshape(B, V, -1), key, torch.full_like(key, 2.0))
        return torch.topk(key, n, dim=-1, largest=False).indices

    def _semantic_table(self, data_dict, dev):
        """[zero row; one text embedding per class]: row ``semantic + 1`` is a pixel's target."""
        index2semantic = data_dict.get("index2semantic")
        if "condition" in data_dict:
            index2semantic = self._valid_embedding(self._condition_index(data_dict), dev)
            data_dict["index2semantic"] = index2semantic
        if index2semantic is None:
            index2semantic = self.class_embedding
        index2semantic = index2semantic.to(dev)
        assert index2semantic.shape[0] > 0
        return torch.cat([index2semantic.new_zeros((1, index2semantic.shape[1])), index2semantic], 0)

    @torch.no_grad()
    def ray_sample(self, data_dict):
        colors = data_dict["rgb"].float()        # (B,V,H,W,3)
        depths = data_dict["depth"].float()      # (B,V,H,W)
        intr = data_dict["intrinsic"].float()
        extr = data_dict["extrinsic"].float()    # (B,V,4,4) world(unit cube) -> camera
        dscale = data_dict["depth_scale"].float()
        B, V, H, W = depths.shape
        n = self.ray_nsample
        dev = depths.device
        if intr.dim() == 3:
            intr = intr[:, None].expand(B, V, *intr.shape[-2:])
        Kmat = intr[..., :3, :3]

        if self.render_semantic:
            table = self._semantic_table(data_dict, dev)

        pix = data_dict.get("ray_pixels")  # optional (B,V,n,2) [y,x] from the caller
        if pix is None:
            flat = self._choose_pixels(depths > 0, n)
        else:
            flat = (pix[..., 0].long() * W + pix[..., 1].long()).to(dev)
        py, px = flat // W, flat % W

        # camera -> world
        RT = torch.zeros((B, V, 4, 4), device=dev)
        RT[..., :3, :4] = extr[..., :3, :4]
        RT[..., 3, 3] = 1
        pose = _inv(RT)
        p = torch.stack([px.float(), py.float(), torch.ones_like(px, dtype=torch.float32)], -1)
        p = (_inv(Kmat)[:, :, None] @ p[..., None]).squeeze(-1)       # (B,V,n,3)
        v = p / torch.linalg.norm(p, ord=2, dim=-1, keepdim=True)
        v = (pose[:, :, None, :3, :3] @ v[..., None]).squeeze(-1)
        ray_d = F.normalize(v, dim=-1)
        ray_o = pose[:, :, None, :3, 3].expand_as(ray_d)

        def pick(img):  # (B,V,H,W,...) -> (B,V,n,...)
            f = img.reshape(B, V, H * W, *img.shape[4:])
            ix = flat.reshape(B, V, n, *([1] * (f.dim() - 3))).expand(-1, -1, -1, *f.shape[3:])
            return torch.gather(f, 2, ix)

        color = pick(colors)
        depth = pick(depths * (depths > 0).float()) * dscale[:, None, None]
        # plane-to-plane depth -> distance along the ray
        cam2world = _inv(extr)
        ez = device_constant((0.0, 0.0, 1.0, 1.0), dev)
        plane = (cam2world @ ez)[..., :3] - ray_o[:, :, 0]
        plane = plane / torch.linalg.norm(plane, dim=-1, keepdim=True)
        depth = depth / (ray_d * plane[:, :, None]).sum(-1)

        inside = self.get_mask_at_box(ray_o[:, :, 0], ray_d)
        color = torch.where(inside[..., None], color, torch.zeros_like(color))
        depth = torch.where(inside, depth, torch.full_like(depth, -0.001))
        ray_dict = dict(ray_o=ray_o.reshape(B, V * n, 3).float(),
                        ray_d=ray_d.reshape(B, V * n, 3).float(),
                        rgb=color.reshape(-1, 3).float(), depth=depth.reshape(-1, 1).float())
        if self.render_semantic:
            sem = pick(data_dict["semantic"])
            sem = torch.where(inside, sem, torch.full_like(sem, -1))
            # class 0 and ignore (-1) -> zero row (reference uses `semantic > 0`)
            row = torch.where(sem > 0, sem + 1, torch.zeros_like(sem)).long()
            ray_dict["semantic"] = table[row.reshape(-1)].float()
        return ray_dict

    @torch.no_grad()
    def prepare_ray(self, data_dict):
        if ray_setup.usable(self, data_dict):   # four launches (csrc/ray_setup.hip), same arithmetic
            return ray_setup.prepare_ray(self, data_dict)
        data_dict = self.to_unit_cube(data_dict)
        return self.ray_sample(data_dict), data_dict

    # ------------------------------------------------------------------ dense volume
    def grid_sample(self, data_dict):
        data_dict["bbox"] = (data_dict["bbox"] // self.grid_size).int()
        data_dict["resolution"] = (data_dict["bbox"][:, 1] - data_dict["bbox"][:, 0]).max(dim=1)[0].int() + 1
        return data_dict

    def _small_scenes(self, data_dict):
        """Scenes whose voxel resolution is below the smallest grid side (< 0.64 m at the ScanNet
        settings): the reference up-samples those instead of pooling them (:218-247).  One host
        read per batch, cached in the dict."""
        if "small_scenes" not in data_dict:
            # the collates pass every scene's coordinate extent on the host: far from the
            # threshold the answer needs no read (the bound allows for the voxel rounding of
            # ``resolution`` and a device-side GridSample that trims the cloud by a voxel or two)
            ext = data_dict.get("extent_host")
            if ext is not None and all(e / self.grid_size - 4 >= min(self.grid_shape) for e in ext):
                data_dict["small_scenes"] = []
                return data_dict["small_scenes"]
            res = data_dict["resolution"] + 1
            data_dict["small_scenes"] = torch.nonzero(res < min(self.grid_shape)).flatten().tolist()
        return data_dict["small_scenes"]

    def _dense_rows(self, data_dict):
        """Row of every voxel in the (B, Z, Y, X) channels-last (or (B, X, Y, Z)) dense grid for
        the pooling branch of the reference (:199-216, scene resolution >= grid)."""
        batch = offset2batch(data_dict["offset"], data_dict["coord"].shape[0])
        G0, G1, G2 = self.grid_shape
        voxel = (data_dict["coord"] // self.grid_size).int()
        res = (data_dict["resolution"] + 1).to(torch.float32)  # current_resolution, (B,)
        shape = device_constant(self.grid_shape, voxel.device)
        cell = res[:, None] / shape[None, :]            # (B,3) anisotropic bin size in voxels
        g = (voxel // cell[batch]).long()
        if self.dense_channels_last:
            lin = (g[:, 2] * G1 + g[:, 1]) * G0 + g[:, 0]  # memory order (Z,Y,X), channels last
        else:
            lin = (g[:, 0] * G1 + g[:, 1]) * G2 + g[:, 2]  # the reference's (X,Y,Z) order
        return lin + batch * (G0 * G1 * G2)

    def _upsampled_scene(self, data_dict, i):
        """(G, C) rows of scene i for a scene SMALLER than the grid: scatter-mean at its own
        resolution, then trilinear resize to the grid (reference :218-247)."""
        edges = [0] + offsets_host(data_dict)
        feat = data_dict["sparse_backbone_feat"][edges[i]:edges[i + 1]]
        voxel = (data_dict["coord"][edges[i]:edges[i + 1]] // self.grid_size).long()
        cur = int(data_dict["resolution"][i] + 1)
        G0, G1, G2 = self.grid_shape
        index = (voxel[:, 0] * cur + voxel[:, 1]) * cur + voxel[:, 2]
        own = scatter(feat, index[:, None], dim=0, reduce=self.pool_type,
                      out=feat.new_zeros((cur ** 3, feat.shape[1])))
        own = own.view(1, cur, cur, cur, -1).permute(0, 4, 3, 2, 1)          # (1, C, z, y, x)
        up = F.interpolate(own, size=(G2, G1, G0), mode="trilinear")          # (1, C, G2, G1, G0)
        if self.dense_channels_last:
            return up.permute(0, 2, 3, 4, 1).reshape(G0 * G1 * G2, -1)
        return up.permute(0, 4, 3, 2, 1).reshape(G0 * G1 * G2, -1)

    def to_dense(self, data_dict):
        """Per-voxel backbone features -> (B, C, Z, Y, X) grid: scatter-mean pooling for scenes at
        least as large as the grid (one launch for all of them), resize for smaller ones."""
        feat = data_dict["sparse_backbone_feat"]
        B, C = data_dict["offset"].numel(), feat.shape[1]
        G0, G1, G2 = self.grid_shape
        G = G0 * G1 * G2
        lin = self._dense_rows(data_dict)
        small = self._small_scenes(data_dict)
        if small:
            batch = offset2batch(data_dict["offset"], data_dict["coord"].shape[0])
            pooled = torch.ones_like(batch, dtype=torch.bool)
            for i in small:
                pooled &= batch != i
            feat, lin = feat[pooled], lin[pooled]
        grid = feat.new_zeros((B * G, C))
        grid = scatter(feat, lin[:, None], dim=0, reduce=self.pool_type, out=grid)
        if small:
            parts = [self._upsampled_scene(data_dict, i) if i in small else grid[i * G:(i + 1) * G]
                     for i in range(B)]
            grid = torch.cat(parts, dim=0)
        if self.dense_channels_last:
            return grid.view(B, G2, G1, G0, C).permute(0, 4, 1, 2, 3)  # channels_last_3d view
        return grid.view(B, G0, G1, G2, C).permute(0, 4, 3, 2, 1).contiguous()

    def _use_cells(self):
        return (self.sparse_dense_input and self.dense_channels_last and self.pool_type == "mean"
                and hasattr(self.proj_net, "forward_cells")
                and getattr(self.proj_net, "cells_supported", lambda: True)())

    def prepare_volume(self, data_dict):
        if not data_dict.pop("_grid_sampled", False):   # (done with the batch's staging: prefetch)
            data_dict = self.grid_sample(data_dict)
        if self._use_cells() and not self._small_scenes(data_dict):
            from .sparse_input import cells_from_voxels

            G0, G1, G2 = self.grid_shape
            geo = data_dict.pop("_cells_geometry", None)
            lin = None if geo is not None else self._dense_rows(data_dict)
            dense = cells_from_voxels(data_dict["sparse_backbone_feat"], lin,
                                      data_dict["offset"].numel(), (G2, G1, G0), geometry=geo)
            project = self.proj_net.forward_cells
        else:
            dense = self.to_dense(data_dict)
            project = self.proj_net
        # UNet3D's last layer is a 1x1x1 convolution, which commutes with trilinear sampling: when
        # all scenes are rendered in one pass by the fused head (training), the head applies it per
        # sample and the 128-channel volume is never materialised (fused_head.FoldedVolume)
        kw = {}
        entry = project.forward if isinstance(project, nn.Module) else project
        if (self.training and self.batched_render and self.dense_channels_last
                and "fold_final" in inspect.signature(entry).parameters):
            kw["fold_final"] = True
        amp_dtype = self._projection_dtype(data_dict["coord"].device)
        if amp_dtype is not None:
            with torch.autocast(data_dict["coord"].device.type, dtype=amp_dtype):
                volume = project(dense, **kw)
            if torch.is_tensor(volume):
                volume = volume.float()
        else:
            volume = project(dense, **kw)
        if self.dense_channels_last and torch.is_tensor(volume):
            volume = volume.contiguous(memory_format=torch.channels_last_3d)
        return [volume]

    # ------------------------------------------------------------------ rendering + losses
    def render_func(self, ray_dict, volume_feature):
        B, R = ray_dict["ray_o"].shape[:2]
        if self.training and self.batched_render:
            # all scenes in one pass: rays are scene-major with equal counts, the field samples the
            # batched volume with a single launch per call (the reference loops over scenes)
            bundle = RayBundle(origins=ray_dict["ray_o"].reshape(B * R, 3),
                               directions=ray_dict["ray_d"].reshape(B * R, 3), num_scenes=B)
            return self.renderer(bundle, volume_feature)
        outs = []
        for i in range(B):
            vols = [v[i] for v in volume_feature]
            o, d = ray_dict["ray_o"][i], ray_dict["ray_d"][i]
            if self.training:
                outs.append(self.renderer(RayBundle(origins=o, directions=d), vols))
            else:
                parts = [self.renderer(RayBundle(origins=po, directions=pd), vols)
                         for po, pd in zip(o.split(self.val_ray_split), d.split(self.val_ray_split))]
                outs.append({k: torch.cat([p[k].detach() for p in parts], 0) for k in parts[0]})
        return {k: torch.cat([o[k] for o in outs], dim=0) for k in outs[0]}

    def render_loss(self, render_out, ray_dict):
        loss_dict = self.renderer.get_loss(render_out, ray_dict)
        total = getattr(loss_dict, "total", None)   # (formed with the terms by the fused loss node)
        if total is None:
            total = sum(v for k, v in loss_dict.items() if "loss" in k)
        return total, loss_dict

    def ppt_loss(self, data_dict):
        # two tall-skinny GEMMs over all voxels (96 -> 512 -> ~20 classes): the MFMA kernels of
        # ponderv2_amd.linear instead of the BLAS's 256x16 macro tiles (measured 165 us per call)
        feat = linear(data_dict["sparse_backbone_feat"], self.proj_head.weight, self.proj_head.bias)
        feat = feat / feat.norm(dim=-1, keepdim=True)
        sim = linear(feat, self._valid_embedding(self._condition_index(data_dict), feat.device))
        return self.ppt_criteria(self.logit_scale.exp() * sim, data_dict["segment"])

    def _projection_dtype(self, device):
        """Reduced precision of the dense projection network: ``proj_autocast`` (a dtype name) or
        the ambient autocast dtype the trainer entered (``enable_amp=True``); None = fp32."""
        if self.proj_autocast is not None and device.type == "cuda":
            return getattr(torch, self.proj_autocast)
        return getattr(self, "_ambient_amp", None)

    def prefetch(self, data_dict):
        """Input-pipeline hook, called one batch ahead on the input stream: launch the sparse backbone's
        geometry for this (device-resident) batch on the geometry stream (SpUNet.prefetch_geometry) and,
        in training, do the ray set-up (``prepare_ray``: scene normalisation, pixel choice, ray
        generation - a function of the batch, constant tables and the random generator only; the
        outdoor reference has it in its dataset transforms) so that it is off the training stream."""
        fn = getattr(self.backbone, "prefetch_geometry", None)
        if fn is not None:
            data_dict = fn(data_dict)
        if (PREFETCH_RAYS and self.training and "_ray_dict" not in data_dict
                and ray_setup.usable(self, data_dict)):
            # prepare_ray rewrites coord / extrinsic / depth_scale to unit-cube values; in the reference
            # (and without the prefetch) extract_feature runs FIRST, on the metric ones.  The metric tensors
            # are kept and put back for the backbone / masking in _forward, so a backbone that reads
            # ``coord`` sees the same thing on every path.  What does differ between the paths
            # is the ORDER of the random draws (pixel choice before the masking draw here).
            metric = {k: data_dict[k] for k in ("coord", "extrinsic", "depth_scale") if k in data_dict}
            ray_dict, data_dict = ray_setup.prepare_ray(self, data_dict)
            data_dict["_ray_dict"] = ray_dict
            data_dict["_metric_inputs"] = metric
            if self._use_cells():
                # ... and the geometry of the projection network's first level (which cells are
                # occupied, the cell -> grid-row pairs of its convolution): coordinates only
                from .sparse_input import cells_geometry

                with torch.no_grad():
                    data_dict = self.grid_sample(data_dict)
                    data_dict["_grid_sampled"] = True
                    if not self._small_scenes(data_dict):
                        G0, G1, G2 = self.grid_shape
                        data_dict["_cells_geometry"] = cells_geometry(
                            self._dense_rows(data_dict), data_dict["offset"].numel(), (G2, G1, G0),
                            build_rulebook=True)
        return data_dict

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
        metric = data_dict.pop("_metric_inputs", None)
        if metric:   # the rays were set up ahead of time: the backbone still sees the metric inputs
            unit = {k: data_dict[k] for k in metric}
            data_dict.update(metric)
            data_dict = self.extract_feature(data_dict)
            data_dict.update(unit)
        else:
            data_dict = self.extract_feature(data_dict)
        ray_dict = data_dict.pop("_ray_dict", None)     # set up with the batch, one step ahead (prefetch)
        if ray_dict is None:
            ray_dict, data_dict = self.prepare_ray(data_dict)
        volume_feature = self.prepare_volume(data_dict)
        render_out = self.render_func(ray_dict, volume_feature)
        loss, loss_dict = self.render_loss(render_out, ray_dict)
        out = dict(loss=loss, **loss_dict)
        if self.ppt_loss_weight > 0:
            out["ppt_loss"] = self.ppt_loss(data_dict)  # reported only (reference :699-704)
        return out
