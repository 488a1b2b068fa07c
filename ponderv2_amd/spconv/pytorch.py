"""MI355X-native stand-in for ``spconv.pytorch`` as used by the reference backbone
(ponder/models/sparse_unet/spconv_unet_v1m1_base.py:11 and its call sites :41,47,58,112,135,
171,220,249): ``SparseConvTensor``, ``SubMConv3d``, ``SparseConv3d``, ``SparseInverseConv3d``,
``SparseSequential``, ``SparseModule``, ``Identity``.

Semantics (spconv itself is not vendored by the reference; these are the definitions we restate
and that oracle/ checks against dense torch convolutions):
  * indices are int32 rows (b, x, y, z); weights keep spconv 2.x's ``[Cout, kx, ky, kz, Cin]``
    layout so reference checkpoints load;
  * SubMConv3d: output sites == input sites; out[i] = sum_k W[k] in[j], coord[j] = coord[i] + k - K//2
    (``padding`` is ignored, as in spconv's submanifold mode);
  * SparseConv3d (kernel == stride, padding 0 - the only form on the path): output sites are the
    unique ``coord // stride`` sorted by (b,x,y,z); offset k = coord % stride;
  * SparseInverseConv3d: the same pairs read backwards; restores the sites saved under
    ``indice_key`` in their original order.
Rulebooks are built once per ``indice_key`` by csrc/rulebook.hip and cached on the tensor.
"""
import math
from collections import OrderedDict
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import convbn, kernels as K
from ..rownorm import fused_bn


MAX_SPATIAL_DIM = 65519   # x + 16 must fit 16 bits
MAX_BATCH_SIZE = 65535


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, indice_dict=None):
        assert features.dim() == 2 and indices.dim() == 2 and indices.shape[1] == 4
        assert indices.dtype == torch.int32, "indices must be int32 (b, x, y, z)"
        self.features = features
        self.indices = indices
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        # the rulebook kernels pack (b, x+16, y+16, z+16) into four 16-bit fields of one 64-bit key
        # (csrc/rulebook.hip pack_key): larger grids would alias distinct voxels silently
        if max(self.spatial_shape) > MAX_SPATIAL_DIM or self.batch_size > MAX_BATCH_SIZE:
            raise ValueError(
                f"SparseConvTensor: spatial_shape {self.spatial_shape} / batch_size {self.batch_size} "
                f"exceed the rulebook key range (dims <= {MAX_SPATIAL_DIM}, batch <= {MAX_BATCH_SIZE})")
        self.indice_dict = {} if indice_dict is None else indice_dict

    def replace_feature(self, feature):
        return SparseConvTensor(feature, self.indices, self.spatial_shape, self.batch_size,
                                self.indice_dict)

    def dense(self, channels_first=True):
        b, (x, y, z), c = self.batch_size, self.spatial_shape, self.features.shape[1]
        out = torch.zeros((b, x, y, z, c), dtype=self.features.dtype, device=self.features.device)
        i = self.indices.long()
        out[i[:, 0], i[:, 1], i[:, 2], i[:, 3]] = self.features
        return out.permute(0, 4, 1, 2, 3).contiguous() if channels_first else out


class SparseModule(nn.Module):
    """Marker base class: modules that consume / produce SparseConvTensor."""


class Identity(nn.Identity):
    pass


def _triple(v) -> List[int]:
    return [int(v)] * 3 if isinstance(v, int) else [int(a) for a in v]


class _SparseConvBase(SparseModule):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None):
        super().__init__()
        ks, st, dl = _triple(kernel_size), _triple(stride), _triple(dilation)
        if len(set(ks)) != 1 or len(set(st)) != 1:
            raise NotImplementedError("only cubic kernels / isotropic strides are on the hot path")
        if dl != [1, 1, 1] or groups != 1:
            raise NotImplementedError("dilation / groups are not used by SpUNet")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = ks, st, _triple(padding)
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.empty(out_channels, *ks, in_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        fan_in = self.in_channels * self.kernel_size[0] ** 3
        bound = 1.0 / math.sqrt(fan_in)
        nn.init.uniform_(self.weight, -bound, bound)  # kaiming_uniform(a=sqrt(5))
        if self.bias is not None:
            nn.init.uniform_(self.bias, -bound, bound)

    def _apply_conv(self, features, rb):
        w = self.weight.reshape(self.out_channels, -1, self.in_channels)
        if K.spconv16_supported(features, w, rb):
            # 16-bit feature matrices (the reduced-precision training mode, precision.py): the
            # bf16 / fp16 MFMA kernels, fp32 accumulation, 16-bit result
            cache = self.__dict__.setdefault("_pv2_packed", {})
            return K.SparseConv16Function.apply(features, w, rb, self.bias, cache)
        # fp32 kernels otherwise, also inside autocast regions (widen whatever the region produced)
        if features.dtype in (torch.bfloat16, torch.float16):
            features = features.float()
        if self.in_channels % 4 and features.is_cuda:
            # zero-pad the reduction axis to a multiple of 8 (the 6-channel stem): exact, and it puts
            # the layer on the vector-load forward kernel and the deterministic weight gradient
            pad = -self.in_channels % 8
            features = F.pad(features, (0, pad))
            w = F.pad(w, (0, pad))
        out = K.SparseConvFunction.apply(features, w, rb)
        if self.bias is not None:
            out = out + self.bias
        return out

    def forward(self, x: SparseConvTensor) -> SparseConvTensor:
        rb, make = self._prepare(x)
        return make(self._apply_conv(x.features, rb))

    def forward_bn(self, x: SparseConvTensor, bn, residual=None, relu=False, weight=None,
                   bias=None) -> SparseConvTensor:
        """``[relu](bn(self(x)) [+ residual])``: one native call per direction where the fused unit
        applies (ponderv2_amd/convbn.py), the conv followed by ``fused_bn`` otherwise.  ``weight`` /
        ``bias``: affine overrides of the BatchNorm (see rownorm.fused_bn)."""
        rb, make = self._prepare(x)
        if x.indices.shape[0] == 0:
            return make(self._apply_conv(x.features, rb))
        feats = x.features
        w = self.weight.reshape(self.out_channels, -1, self.in_channels)
        if self.bias is None and convbn.supported(feats, w, rb, bn):
            return make(convbn.conv_bn(self, bn, feats, rb, residual=residual, relu=relu,
                                       weight=weight, bias=bias))
        y = self._apply_conv(feats, rb)
        if weight is not None or bias is not None:
            return make(fused_bn(bn, y, residual=residual, relu=relu, weight=weight, bias=bias))
        return make(fused_bn(bn, y, residual=residual, relu=relu))

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, "
                f"stride={self.stride}, indice_key={self.indice_key}")


class SubMConv3d(_SparseConvBase):
    def _prepare(self, x: SparseConvTensor):
        ks = self.kernel_size[0]
        key = self.indice_key
        entry = x.indice_dict.get(key) if key is not None else None
        if entry is not None and entry["kind"] == "subm" and entry["ksize"] == ks \
                and entry["n"] == x.indices.shape[0]:
            rb = entry["rulebook"]
        else:
            rb = K.build_subm_rulebook(x.indices, ks)
            if key is not None:
                x.indice_dict[key] = dict(kind="subm", ksize=ks, n=x.indices.shape[0], rulebook=rb)
        return rb, x.replace_feature


class SparseConv3d(_SparseConvBase):
    def _prepare(self, x: SparseConvTensor):
        ks, st = self.kernel_size[0], self.stride[0]
        if ks != st or self.padding != [0, 0, 0]:
            raise NotImplementedError(
                "SparseConv3d: only kernel_size == stride, padding == 0 (the SpUNet down-conv)")
        out_shape = [(s - ks) // st + 1 for s in x.spatial_shape]
        entry = x.indice_dict.get(self.indice_key) if self.indice_key is not None else None
        if (entry is not None and entry.get("prebuilt") and entry["kind"] == "down"
                and entry["ksize"] == ks and entry["rulebook"].n_in == x.indices.shape[0]
                and entry["out_shape"] == out_shape):
            # built ahead by kernels.prepare_unet_geometry (one host read for the whole U-Net)
            rb, out_indices = entry["rulebook"], entry["out_indices"]
            entry["in_indices"] = x.indices
        else:
            rb, out_indices = K.build_downsample_rulebook(x.indices, st, out_shape)
            if self.indice_key is not None:
                x.indice_dict[self.indice_key] = dict(
                    kind="down", ksize=ks, rulebook=rb, in_indices=x.indices,
                    in_spatial_shape=x.spatial_shape)
        return rb, lambda f: SparseConvTensor(f, out_indices, out_shape, x.batch_size, x.indice_dict)


class SparseInverseConv3d(_SparseConvBase):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key=None, bias=True):
        super().__init__(in_channels, out_channels, kernel_size, bias=bias, indice_key=indice_key)

    def _prepare(self, x: SparseConvTensor):
        entry = x.indice_dict.get(self.indice_key)
        if entry is None or entry["kind"] != "down":
            raise RuntimeError(f"SparseInverseConv3d: no strided conv saved under "
                               f"indice_key={self.indice_key!r}")
        rb = entry.get("rulebook_t")
        if rb is None:   # one transposed view per strided conv (its position tables are cached on it)
            rb = entry["rulebook_t"] = entry["rulebook"].transposed()
        return rb, lambda f: SparseConvTensor(f, entry["in_indices"], entry["in_spatial_shape"],
                                              x.batch_size, x.indice_dict)


class SparseSequential(SparseModule):
    """nn.Sequential that lets dense modules (BatchNorm1d, ReLU, ...) act on ``.features``."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for name, module in args[0].items():
                self.add_module(name, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            self.add_module(name, module)

    def __len__(self):
        return len(self._modules)

    def __getitem__(self, idx):
        return list(self._modules.values())[idx]

    def forward(self, x):
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            module = mods[i]
            if (isinstance(module, _SparseConvBase) and i + 1 < len(mods)
                    and type(mods[i + 1]) is nn.BatchNorm1d and isinstance(x, SparseConvTensor)):
                # conv + BatchNorm1d (+ ReLU): one fused unit (convbn.py)
                relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
                x = module.forward_bn(x, mods[i + 1], relu=relu)
                i += 1 + int(relu)
            elif isinstance(module, SparseModule):
                x = module(x)
            elif isinstance(x, SparseConvTensor):
                if x.indices.shape[0] != 0:
                    if isinstance(module, nn.BatchNorm1d):
                        # BatchNorm1d (+ ReLU) on the feature matrix: one fused op (rownorm.hip)
                        relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                        x = x.replace_feature(fused_bn(module, x.features, relu=relu))
                        i += int(relu)
                    else:
                        x = x.replace_feature(module(x.features))
            else:
                x = module(x)
            i += 1
        return x
