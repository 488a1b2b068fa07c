"""Closed-form "random-looking" tensors keyed by a name, so the reference model and the product
model get IDENTICAL parameters without sharing an RNG stream or a checkpoint file."""
import math
import zlib

import torch


def formula_tensor(name, shape, scale=1.0, dtype=torch.float32):
    n = 1
    for s in shape:
        n *= int(s)
    phase = (zlib.crc32(name.encode()) % 100003) / 100003.0 * 2 * math.pi
    idx = torch.arange(n, dtype=torch.float64)
    vals = torch.sin(idx * 0.7391 + phase) * torch.cos(idx * 0.0137 + 2 * phase)
    return (vals * scale).reshape(*shape).to(dtype)


def fill_deterministic(module):
    """Overwrite every parameter with >1 element: weights ~ U-ish(+-sqrt(3/fan_in)) * 0.8,
    norm scales 1 +- 0.1, biases +- 0.05.  Scalars (variance, logit_scale, ...) keep their init."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if p.numel() <= 1:
                continue
            if p.dim() >= 2:
                fan_in = p.numel() // p.shape[0]
                v = formula_tensor(name, p.shape, 0.8 * math.sqrt(3.0 / fan_in) * 2.0)
            elif name.endswith("weight"):  # norm scale
                v = 1.0 + formula_tensor(name, p.shape, 0.2)
            else:
                v = formula_tensor(name, p.shape, 0.1)
            p.copy_(v.to(p.dtype))
    return module


def real_init_modulation(module, scale=0.05):
    """For the real-initialisation fixtures of SpUNet-v1m3: the reference zero-initialises the PDNorm
    modulation layers (spconv_unet_v1m3_pdnorm.py:389-404), which would leave the modulation path without
    signal.  Both sides call this after their constructors: small closed-form values for exactly those
    layers, everything else keeps the constructor's (seeded) initialisation."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if ".modulation." in name:
                p.copy_(formula_tensor(name, p.shape, scale).to(p.dtype))
    return module


GRAD_PROBES = 8


def grad_probe(name, i, shape):
    """The i-th random probe of parameter ``name``: iid standard normals from a CPU generator seeded
    by the name (same torch build here and on the GPU box => same values).  A gradient's projection
    on it estimates the gradient ERROR without storing 40 M reference values: for e = g - g_ref,
    E[(probe . e)^2] = |e|^2."""
    g = torch.Generator().manual_seed(zlib.crc32(f"{name}|{i}".encode()))
    return torch.randn(tuple(shape), generator=g, dtype=torch.float32)
