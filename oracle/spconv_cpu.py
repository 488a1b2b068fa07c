"""CPU stand-in for ``spconv.pytorch`` built on oracle.rulebook + oracle.sparse_ops.

Lets (a) the reference's own ponder/models/sparse_unet/spconv_unet_v1m1_base.py and (b) the
product's model files run on the host, for golden-vector generation, CPU parity tests and the
bench's cpu_baseline leg.  Same public names as the reference imports at
spconv_unet_v1m1_base.py:11,21,41,47,112,135,171,220,249.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from . import rulebook as rbk
from .sparse_ops import sparse_conv


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, indice_dict=None):
        self.features = features
        self.indices = indices
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = indice_dict if indice_dict is not None else {}

    def replace_feature(self, feature):
        return SparseConvTensor(feature, self.indices, self.spatial_shape, self.batch_size,
                                self.indice_dict)


class SparseModule(nn.Module):
    pass


class Identity(nn.Identity):
    pass


def _as_rule(pin, pout, kstart, n_in, n_out):
    return dict(pair_in=torch.from_numpy(pin.astype(np.int64)),
                pair_out=torch.from_numpy(pout.astype(np.int64)),
                kstart=[int(v) for v in kstart], n_in=n_in, n_out=n_out)


class _Conv(SparseModule):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.ksize = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        self.stride = stride if isinstance(stride, int) else stride[0]
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.empty(out_channels, self.ksize, self.ksize, self.ksize,
                                               in_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        bound = 1.0 / math.sqrt(in_channels * self.ksize ** 3)
        nn.init.uniform_(self.weight, -bound, bound)
        if self.bias is not None:
            nn.init.uniform_(self.bias, -bound, bound)

    def _run(self, feats, rule, transposed=False):
        w = self.weight.reshape(self.out_channels, -1, self.in_channels)
        if transposed:
            out = sparse_conv(feats, w, rule["pair_out"], rule["pair_in"], rule["kstart"],
                              rule["n_in"])
        else:
            out = sparse_conv(feats, w, rule["pair_in"], rule["pair_out"], rule["kstart"],
                              rule["n_out"])
        return out if self.bias is None else out + self.bias


class SubMConv3d(_Conv):
    def forward(self, x):
        key = self.indice_key
        ent = x.indice_dict.get(key) if key is not None else None
        if ent is None or ent.get("ksize") != self.ksize or ent["rule"]["n_in"] != len(x.indices):
            pin, pout, ks = rbk.subm_rulebook(x.indices.numpy(), self.ksize)
            n = len(x.indices)
            ent = dict(ksize=self.ksize, rule=_as_rule(pin, pout, ks, n, n))
            if key is not None:
                x.indice_dict[key] = ent
        return x.replace_feature(self._run(x.features, ent["rule"]))


class SparseConv3d(_Conv):
    def forward(self, x):
        assert self.ksize == self.stride
        out_shape = [(s - self.ksize) // self.stride + 1 for s in x.spatial_shape]
        oc, pin, pout, ks = rbk.downsample_rulebook(x.indices.numpy(), self.stride, out_shape)
        rule = _as_rule(pin, pout, ks, len(x.indices), len(oc))
        if self.indice_key is not None:
            x.indice_dict[self.indice_key] = dict(rule=rule, in_indices=x.indices,
                                                  in_shape=x.spatial_shape)
        return SparseConvTensor(self._run(x.features, rule), torch.from_numpy(oc), out_shape,
                                x.batch_size, x.indice_dict)


class SparseInverseConv3d(_Conv):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key=None, bias=True):
        super().__init__(in_channels, out_channels, kernel_size, bias=bias, indice_key=indice_key)

    def forward(self, x):
        ent = x.indice_dict[self.indice_key]
        return SparseConvTensor(self._run(x.features, ent["rule"], transposed=True),
                                ent["in_indices"], ent["in_shape"], x.batch_size, x.indice_dict)


class SparseSequential(SparseModule):
    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for name, mod in args[0].items():
                self.add_module(name, mod)
        else:
            for i, mod in enumerate(args):
                self.add_module(str(i), mod)
        for name, mod in kwargs.items():
            self.add_module(name, mod)

    def forward(self, x):
        for mod in self._modules.values():
            if isinstance(mod, SparseModule):
                x = mod(x)
            elif isinstance(x, SparseConvTensor):
                x = x.replace_feature(mod(x.features))
            else:
                x = mod(x)
        return x
