"""SpUNet-v1m3: SpUNet with prompt-driven normalisation (PDNorm) for multi-dataset pre-training.

Mirror of ponder/models/sparse_unet/spconv_unet_v1m3_pdnorm.py (PDBatchNorm :23-72, BasicBlock
:75-146, SPConvDown/Up/PatchEmbedding :149-233, SpUNetBase :236-427).  Same module / parameter
names (``bns.{k}`` per condition, ``modulation.1``) so reference checkpoints load, same
``forward(input_dict)`` contract (``condition`` list + optional ``context`` row).

Every normalisation layer owns one BatchNorm1d per dataset ("decouple") and, when "adaptive", a
SiLU -> Linear map of the dataset's context vector to a per-channel (shift, scale).  On the GPU
the modulation costs nothing extra: ``bn(x) * (1 + scale) + shift`` is BatchNorm with the affine
pair (gamma * (1 + scale), beta * (1 + scale) + shift), so the fused BatchNorm(+add+ReLU) kernel
(csrc/rownorm.hip) runs with those two C-vectors in its epilogue and autograd carries their
gradients back to gamma / beta / the modulation Linear.
"""
from collections import OrderedDict
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from ponderv2_amd import spunet_native
from ponderv2_amd.rownorm import can_fuse, fused_bn
from ponderv2_amd.spconv import pytorch as spconv
from ..builder import MODELS
from ..utils import offset2batch
from .spconv_unet_v1m1_base import trunc_normal_


class PreparedContext:
    """A forward's context row together with every PDNorm layer's effective affine pair,
    computed for all layers at once by ``SpUNetBase._prepare_modulation``."""

    __slots__ = ("context", "pairs")

    def __init__(self, context, pairs):
        self.context, self.pairs = context, pairs


class PDBatchNorm(nn.Module):
    def __init__(self, num_features, context_channels=256, eps=1e-3, momentum=0.01,
                 conditions=("ScanNet", "S3DIS", "Structured3D"), decouple=True, adaptive=False,
                 affine=True):
        super().__init__()
        self.conditions, self.decouple = conditions, decouple
        self.adaptive, self.affine = adaptive, affine
        make = partial(nn.BatchNorm1d, num_features=num_features, eps=eps, momentum=momentum,
                       affine=affine)
        if decouple:
            self.bns = nn.ModuleList([make() for _ in conditions])
        else:
            self.bn = make()
        if adaptive:
            self.modulation = nn.Sequential(nn.SiLU(),
                                            nn.Linear(context_channels, 2 * num_features, bias=True))

    def forward(self, feat, condition=None, context=None, residual=None, relu=False):
        """[relu](modulate(bn_condition(feat)) [+ residual])"""
        if self.decouple:
            assert condition in self.conditions
            bn = self.bns[self.conditions.index(condition)]
        else:
            bn = self.bn
        if not self.adaptive:
            return fused_bn(bn, feat, residual=residual, relu=relu)
        assert context is not None
        if isinstance(context, PreparedContext):
            if can_fuse(bn, feat):
                weight, bias = context.pairs[self]
                return fused_bn(bn, feat, residual=residual, relu=relu, weight=weight, bias=bias)
            context = context.context
        shift, scale = self.modulation(context).chunk(2, dim=1)
        if can_fuse(bn, feat) and shift.shape[0] == 1:
            gain = 1.0 + scale[0]
            if bn.affine:
                weight, bias = bn.weight * gain, bn.bias * gain + shift[0]
            else:
                weight, bias = gain, shift[0]
            return fused_bn(bn, feat, residual=residual, relu=relu, weight=weight, bias=bias)
        y = bn(feat) * (1.0 + scale) + shift  # the reference's evaluation order (:69-71)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y


def _conv_norm(conv, norm, x, condition, context, residual=None, relu=False):
    """``[relu](norm(conv(x)) [+ residual])`` for a sparse conv and a PDBatchNorm: the fused conv +
    BatchNorm unit (ponderv2_amd/convbn.py, with the layer's effective affine pair) where it applies."""
    bn = norm.bns[norm.conditions.index(condition)] if norm.decouple else norm.bn
    if not norm.adaptive:
        return conv.forward_bn(x, bn, residual=residual, relu=relu)
    if isinstance(context, PreparedContext) and can_fuse(bn, x.features):
        weight, bias = context.pairs[norm]
        return conv.forward_bn(x, bn, residual=residual, relu=relu, weight=weight, bias=bias)
    y = conv(x)
    return y.replace_feature(norm(y.features, condition, context, residual=residual, relu=relu))


class BasicBlock(spconv.SparseModule):
    expansion = 1

    def __init__(self, in_channels, embed_channels, stride=1, norm_fn=None, indice_key=None,
                 bias=False):
        super().__init__()
        assert norm_fn is not None
        self.in_channels, self.embed_channels = in_channels, embed_channels
        if in_channels == embed_channels:
            self.proj = spconv.SparseSequential(nn.Identity())
        else:
            self.proj_conv = spconv.SubMConv3d(in_channels, embed_channels, kernel_size=1, bias=False)
            self.proj_norm = norm_fn(embed_channels)
        conv = partial(spconv.SubMConv3d, kernel_size=3, stride=stride, padding=1, bias=bias,
                       indice_key=indice_key)
        self.conv1 = conv(in_channels, embed_channels)
        self.bn1 = norm_fn(embed_channels)
        self.relu = nn.ReLU()
        self.conv2 = conv(embed_channels, embed_channels)
        self.bn2 = norm_fn(embed_channels)
        self.stride = stride

    def forward(self, x):
        x, condition, context = x
        y = _conv_norm(self.conv1, self.bn1, x, condition, context, relu=True)
        if self.in_channels == self.embed_channels:
            shortcut = self.proj(x).features
        else:
            shortcut = _conv_norm(self.proj_conv, self.proj_norm, x, condition, context).features
        y = _conv_norm(self.conv2, self.bn2, y, condition, context, residual=shortcut, relu=True)
        return y, condition, context


class _ConvNormReLU(nn.Module):
    """conv -> PDNorm -> ReLU on a [tensor, condition, context] triple (returns the tensor)."""

    def forward(self, x):
        x, condition, context = x
        return _conv_norm(self.conv, self.bn, x, condition, context, relu=True)


class SPConvDown(_ConvNormReLU):
    def __init__(self, in_channels, out_channels, indice_key, kernel_size=2, bias=False,
                 norm_fn=None):
        super().__init__()
        self.conv = spconv.SparseConv3d(in_channels, out_channels, kernel_size=kernel_size,
                                        stride=kernel_size, bias=bias, indice_key=indice_key)
        self.bn = norm_fn(out_channels)
        self.relu = nn.ReLU()


class SPConvUp(_ConvNormReLU):
    def __init__(self, in_channels, out_channels, indice_key, kernel_size=2, bias=False,
                 norm_fn=None):
        super().__init__()
        self.conv = spconv.SparseInverseConv3d(in_channels, out_channels, kernel_size=kernel_size,
                                               bias=bias, indice_key=indice_key)
        self.bn = norm_fn(out_channels)
        self.relu = nn.ReLU()


class SPConvPatchEmbedding(_ConvNormReLU):
    def __init__(self, in_channels, out_channels, kernel_size=5, norm_fn=None):
        super().__init__()
        self.conv = spconv.SubMConv3d(in_channels, out_channels, kernel_size=kernel_size,
                                      padding=1, bias=False, indice_key="stem")
        self.bn = norm_fn(out_channels)
        self.relu = nn.ReLU()


@MODELS.register_module("SpUNet-v1m3")
class SpUNetBase(nn.Module):
    def __init__(self, in_channels, num_classes=0, base_channels=32, context_channels=256,
                 channels=(32, 64, 128, 256, 256, 128, 96, 96), layers=(2, 3, 4, 6, 2, 2, 2, 2),
                 cls_mode=False, conditions=("ScanNet", "S3DIS", "Structured3D"), zero_init=True,
                 norm_decouple=True, norm_adaptive=True, norm_affine=False):
        super().__init__()
        assert len(layers) % 2 == 0 and len(layers) == len(channels)
        self.in_channels, self.num_classes = in_channels, num_classes
        self.base_channels, self.channels, self.layers = base_channels, channels, layers
        self.num_stages = len(layers) // 2
        self.cls_mode, self.conditions, self.zero_init = cls_mode, conditions, zero_init
        norm_fn = partial(PDBatchNorm, eps=1e-3, momentum=0.01, conditions=conditions,
                          context_channels=context_channels, decouple=norm_decouple,
                          adaptive=norm_adaptive, affine=norm_affine)

        self.conv_input = SPConvPatchEmbedding(in_channels, base_channels, kernel_size=5,
                                               norm_fn=norm_fn)
        self.down, self.up = nn.ModuleList(), nn.ModuleList()
        self.enc = nn.ModuleList()
        self.dec = nn.ModuleList() if not cls_mode else None
        enc_c, dec_c = base_channels, channels[-1]
        n = len(channels)
        for s in range(self.num_stages):
            self.down.append(SPConvDown(enc_c, channels[s], kernel_size=2, bias=False,
                                        indice_key=f"spconv{s + 1}", norm_fn=norm_fn))
            self.enc.append(spconv.SparseSequential(OrderedDict(
                (f"block{i}", BasicBlock(channels[s], channels[s], norm_fn=norm_fn,
                                         indice_key=f"subm{s + 1}"))
                for i in range(layers[s]))))
            if not cls_mode:
                self.up.append(SPConvUp(channels[n - s - 2], dec_c, kernel_size=2, bias=False,
                                        indice_key=f"spconv{s + 1}", norm_fn=norm_fn))
                self.dec.append(spconv.SparseSequential(OrderedDict(
                    (f"block{i}", BasicBlock(dec_c + enc_c if i == 0 else dec_c, dec_c,
                                             norm_fn=norm_fn, indice_key=f"subm{s}"))
                    for i in range(layers[n - s - 1]))))
            enc_c, dec_c = channels[s], channels[n - s - 2]

        final_in = channels[-1] if not cls_mode else channels[self.num_stages - 1]
        self.final = (spconv.SubMConv3d(final_in, num_classes, kernel_size=1, padding=1, bias=True)
                      if num_classes > 0 else spconv.Identity())
        self.apply(self._init_weights)
        self._index_modulation()

    def _index_modulation(self):
        """Where each layer's (shift, scale) halves sit in the concatenation of all modulation
        outputs - static, so the per-forward work is one GEMV and a handful of vector ops."""
        self._pd_norms = [m for m in self.modules() if isinstance(m, PDBatchNorm) and m.adaptive]
        shift_idx, scale_idx, o = [], [], 0
        for m in self._pd_norms:
            c = m.modulation[1].out_features // 2
            shift_idx.append(torch.arange(o, o + c))
            scale_idx.append(torch.arange(o + c, o + 2 * c))
            o += 2 * c
        if self._pd_norms:
            self.register_buffer("_shift_index", torch.cat(shift_idx), persistent=False)
            self.register_buffer("_scale_index", torch.cat(scale_idx), persistent=False)

    def _prepare_modulation(self, condition, context):
        """All layers' ``SiLU -> Linear`` modulations as ONE matrix-vector product, folded with the
        selected dataset's BatchNorm affine parameters into per-layer (weight, bias) views for the
        fused kernel's epilogue.  Per layer this replaces ~6 tiny launches (and ~10 in backward)
        by none; autograd splits the concatenated gradients back onto the per-layer parameters."""
        norms = self._pd_norms
        if (not norms or context is None or not context.is_cuda or context.shape[0] != 1
                or context.dtype != torch.float32 or not self.training
                or torch.is_autocast_enabled()):
            return context
        lin = [m.modulation[1] for m in norms]
        mod = F.linear(F.silu(context), torch.cat([l.weight for l in lin]),
                       torch.cat([l.bias for l in lin]))[0]
        gain = 1.0 + mod[self._scale_index]
        shift = mod[self._shift_index]
        bns = [m.bns[m.conditions.index(condition)] if m.decouple else m.bn for m in norms]
        if bns[0].affine:
            weight = torch.cat([b.weight for b in bns]) * gain
            bias = torch.cat([b.bias for b in bns]) * gain + shift
        else:
            weight, bias = gain, shift
        pairs, o = {}, 0
        for m, b in zip(norms, bns):
            pairs[m] = (weight[o:o + b.num_features], bias[o:o + b.num_features])
            o += b.num_features
        return PreparedContext(context, pairs)

    def _init_weights(self, m):
        # Module.apply visits children first, so zero_init wins over the Linear's own init (:389-404)
        if isinstance(m, (nn.Linear, spconv.SubMConv3d)):
            trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.BatchNorm1d):
            if m.affine:
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        elif isinstance(m, PDBatchNorm):
            if self.zero_init and m.adaptive:
                nn.init.constant_(m.modulation[-1].weight, 0)
                nn.init.constant_(m.modulation[-1].bias, 0)

    def _geometry(self, feat, batch, grid_coord, sparse_shape, pending=None):
        """All ten rulebooks of the U-Net with ONE device->host read (kernels.prepare_unet_geometry)
        instead of one or two per rulebook - or none at all when the batch carries the handle of a
        build launched a step ahead (``prefetch_geometry``); host tensors (the CPU test doubles)
        build lazily."""
        if not feat.is_cuda or getattr(self, "cls_mode", False):
            return None
        from ponderv2_amd import kernels as K

        if isinstance(pending, K.PendingGeometry) and pending.n_rows == grid_coord.shape[0]:
            return pending.result()
        indices = torch.cat([batch.unsqueeze(-1).int(), grid_coord.int()], dim=1).contiguous()
        return K.prepare_unet_geometry(indices, sparse_shape, n_levels=self.num_stages)

    def prefetch_geometry(self, input_dict):
        """Launch this batch's rulebook builds on the geometry side stream and leave the handle
        in ``input_dict["geometry"]`` (kernels.prefetch_unet_geometry).  Input-pipeline work: call
        it for batch i+1 before step i is enqueued, and step i+1 starts without a host stall.
        Needs ``sparse_shape`` in the batch (the collates provide it)."""
        grid_coord = input_dict["grid_coord"]
        if (not grid_coord.is_cuda or getattr(self, "cls_mode", False)
                or input_dict.get("sparse_shape") is None):
            return input_dict
        from ponderv2_amd import kernels as K

        batch = offset2batch(input_dict["offset"], grid_coord.shape[0])
        indices = torch.cat([batch.unsqueeze(-1).int(), grid_coord.int()], dim=1).contiguous()
        input_dict["geometry"] = K.prefetch_unet_geometry(indices, input_dict["sparse_shape"],
                                                          n_levels=self.num_stages)
        return input_dict

    def forward(self, input_dict):
        grid_coord, feat, offset = input_dict["grid_coord"], input_dict["feat"], input_dict["offset"]
        condition = input_dict["condition"][0]
        context = self._prepare_modulation(condition, input_dict.get("context"))
        batch = offset2batch(offset, grid_coord.shape[0])
        sparse_shape = input_dict.get("sparse_shape")
        if sparse_shape is None:
            sparse_shape = torch.add(torch.max(grid_coord, dim=0).values, 96).tolist()
        x = spconv.SparseConvTensor(
            features=feat,
            indices=torch.cat([batch.unsqueeze(-1).int(), grid_coord.int()], dim=1).contiguous(),
            spatial_shape=sparse_shape, batch_size=offset.numel(),
            indice_dict=self._geometry(feat, batch, grid_coord, sparse_shape,
                                       input_dict.get("geometry")))
        native = spunet_native.run(self, x, condition, context) if feat.is_cuda else None
        if native is not None:   # one native call per direction (ponderv2_amd/spunet_native.py)
            x = x.replace_feature(native)
        else:
            x = self.conv_input([x, condition, context])
            skips = [x]
            for s in range(self.num_stages):
                x = self.down[s]([x, condition, context])
                x, _, _ = self.enc[s]([x, condition, context])
                skips.append(x)
            x = skips.pop(-1)
            if not self.cls_mode:
                for s in reversed(range(self.num_stages)):
                    x = self.up[s]([x, condition, context])
                    skip = skips.pop(-1)
                    x = x.replace_feature(torch.cat((x.features, skip.features), dim=1))
                    x, _, _ = self.dec[s]([x, condition, context])
        x = self.final(x)
        if self.cls_mode:
            b = x.indices[:, 0].long()
            summed = x.features.new_zeros((offset.numel(), x.features.shape[1])).index_add(
                0, b, x.features)
            x = x.replace_feature(summed / torch.bincount(b, minlength=offset.numel())
                                  .clamp(min=1).unsqueeze(1))
        return x.features
