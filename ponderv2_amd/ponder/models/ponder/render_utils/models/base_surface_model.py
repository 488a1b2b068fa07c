"""Surface-rendering model skeleton: collide -> sample + field -> composite -> losses.

Restates ponder/models/ponder/render_utils/models/base_surface_model.py (get_outputs :34-89,
forward :91-100, get_loss :102-211).  Only the training branch of the semantic loss is
implemented - the reference's evaluation branch references an undefined variable (:162-163,
SURVEY Q2) and cannot run.
"""
import torch
import torch.nn.functional as F
from torch import nn

from ponderv2_amd import ray_epilogue, surface_loss
from ..builder import build_collider, build_field, build_sampler
from ..renderers import DepthRenderer, NormalRenderer, RGBRenderer, SemanticRenderer


class SurfaceModel(nn.Module):
    def __init__(self, field, collider, sampler, loss, **kwargs):
        super().__init__()
        self.field = build_field(field)
        self.collider = build_collider(collider)
        self.sampler = build_sampler(sampler)
        self.rgb_renderer = RGBRenderer()
        self.depth_renderer = DepthRenderer()
        self.normal_renderer = NormalRenderer()
        self.semantic_renderer = SemanticRenderer()
        self.loss = loss

    def sample_and_forward_field(self, ray_bundle, volume_feature):
        raise NotImplementedError

    def get_outputs(self, ray_bundle, volume_feature, **kwargs):
        s = self.sample_and_forward_field(ray_bundle, volume_feature)
        field_outputs, ray_samples, weights = s["field_outputs"], s["ray_samples"], s["weights"]
        outputs = {}
        if "rgb" in field_outputs:
            outputs["rgb"] = self.rgb_renderer(rgb=field_outputs["rgb"], weights=weights)
        if "semantic" in field_outputs:
            outputs["semantic"] = self.semantic_renderer(semantic=field_outputs["semantic"],
                                                         weights=weights)
        elif "semantic_hidden" in field_outputs:  # composite first, last linear layer per ray
            lin = self.field.semantic_decoder.last_linear
            comp = self.semantic_renderer(semantic=field_outputs["semantic_hidden"], weights=weights)
            outputs["semantic"] = F.linear(comp, lin.weight) + lin.bias * torch.sum(weights, dim=-2)
        outputs.update(
            depth=self.depth_renderer(ray_samples=ray_samples, weights=weights),
            normal=self.normal_renderer(normals=field_outputs["normal"], weights=weights),
            weights=weights, sdf=field_outputs["sdf"], gradients=field_outputs["gradients"],
            z_vals=ray_samples.frustums.starts, sampled_points=s["sampled_points"])
        if s.get("init_sampled_points") is not None:
            outputs.update(init_sampled_points=s["init_sampled_points"],
                           init_weights=s["init_weights"],
                           new_sampled_points=s["new_sampled_points"])
        if self.loss.weights.get("sparse_points_sdf_loss", 0.0) > 0:
            outputs["sparse_points_sdf"] = self.field.get_sdf(
                kwargs["points"].unsqueeze(0), volume_feature)[0].squeeze(0)
        return outputs

    def forward(self, ray_bundle, volume_feature, **kwargs):
        return self.get_outputs(self.collider(ray_bundle), volume_feature, **kwargs)

    def _semantic_loss(self, preds_dict, targets, valid):
        if not self.training:
            raise NotImplementedError("semantic loss is only defined for training (SURVEY Q2)")
        sem_pred = F.normalize(preds_dict["semantic"], dim=-1)
        sem_gt = targets["semantic"]
        ok = (valid * sem_gt.any(dim=-1, keepdim=True)).squeeze(-1).bool()
        logits = torch.mm(sem_pred, sem_gt.transpose(1, 0)) / self.loss.temperature
        labels = torch.arange(sem_pred.shape[0], dtype=torch.long, device=sem_pred.device)
        labels = torch.where(ok, labels, torch.full_like(labels, -100))
        # all-ignored -> exactly 0 (cross_entropy would give nan); no host sync
        ce = F.cross_entropy(logits, labels, reduction="sum") / ok.sum().clamp(min=1)
        return ce * self.loss.weights.semantic_loss

    def get_loss(self, preds_dict, targets):
        lw = self.loss.weights
        out = {}
        if ray_epilogue.usable(preds_dict, targets, self.loss) and self.training:
            # the fused head kept its composite rows: per-ray epilogue, semantic head and EVERY term in
            # one autograd node (ponderv2_amd/ray_epilogue.py); ``.total`` is their sum
            return ray_epilogue.ray_losses(preds_dict, targets, self.loss)
        depth_gt = targets["depth"]
        valid = depth_gt > 0.0
        if surface_loss.usable(preds_dict, targets) and lw.get("sparse_points_sdf_loss", 0.0) <= 0:
            # depth / colour / free-space / SDF / eikonal terms in three launches (csrc/surface_loss.hip);
            # the semantic term below is the same code as on the modular route
            fused = surface_loss.surface_losses(preds_dict, targets, self.loss)
            for k in ("depth_loss", "rgb_loss", "psnr"):
                if k in fused:
                    out[k] = fused[k]
            if lw.get("semantic_loss", 0.0) > 0:
                out["semantic_loss"] = self._semantic_loss(preds_dict, targets, valid)
            for k in ("free_space_loss", "sdf_loss", "eikonal_loss"):
                if k in fused:
                    out[k] = fused[k]
            return out
        if lw.get("depth_loss", 0.0) > 0:
            l1 = torch.sum(valid * torch.abs(depth_gt - preds_dict["depth"]))
            out["depth_loss"] = l1 / torch.clamp(torch.sum(valid), min=1.0) * lw.depth_loss
        if lw.get("rgb_loss", 0.0) > 0:
            rgb_pred, rgb_gt = preds_dict["rgb"], targets["rgb"]
            out["rgb_loss"] = torch.mean(torch.abs(rgb_pred - rgb_gt)) * lw.rgb_loss
            out["psnr"] = 20.0 * torch.log10(1.0 / torch.mean((rgb_pred - rgb_gt).pow(2)).sqrt())
        if lw.get("semantic_loss", 0.0) > 0:
            out["semantic_loss"] = self._semantic_loss(preds_dict, targets, valid)
        pred_sdf = preds_dict["sdf"][..., 0]
        z_vals = preds_dict["z_vals"][..., 0]
        trunc = self.loss.sensor_depth_truncation
        front = valid & (z_vals < (depth_gt - trunc))
        back = valid & (z_vals > (depth_gt + trunc))
        near = valid & (~front) & (~back)
        if lw.get("free_space_loss", 0.0) > 0:
            fs = torch.sum(F.relu(trunc - pred_sdf) * front) / torch.clamp(torch.sum(front), min=1.0)
            out["free_space_loss"] = fs * lw.free_space_loss
        if lw.get("sdf_loss", 0.0) > 0:
            sl = torch.sum(torch.abs(z_vals + pred_sdf - depth_gt) * near) / torch.clamp(torch.sum(near), min=1.0)
            out["sdf_loss"] = sl * lw.sdf_loss
        if lw.get("eikonal_loss", 0.0) > 0:
            g = preds_dict["gradients"]
            out["eikonal_loss"] = torch.mean((g.norm(2, dim=-1) - 1) ** 2) * lw.eikonal_loss
        if lw.get("sparse_points_sdf_loss", 0.0) > 0:
            out["sparse_points_sdf_loss"] = torch.mean(torch.abs(preds_dict["sparse_points_sdf"])) \
                * lw.sparse_points_sdf_loss
        return out
