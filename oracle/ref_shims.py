"""Import-time shims that let the reference's OWN model files run on the host (this container
only - /root/reference does not exist on the GPU box).  Used by oracle/make_golden.py and by the
CPU parity tests that compare our restatement with the reference itself.

Installs into sys.modules: spconv(.pytorch) -> oracle.spconv_cpu, smooth_sampler -> oracle.sampler,
torch_scatter / torch_geometric.utils -> oracle.scatter, timm.models.layers.trunc_normal_ -> torch's,
clip -> a deterministic fake text encoder (same embeddings as the product's stub).
"""
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("PONDERV2_REFERENCE", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "ponder"))


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _FakeClipModel(torch.nn.Module):
    def __init__(self, table):
        super().__init__()
        self.table = table
        self.text_projection = torch.nn.Parameter(torch.zeros(table.shape[1], table.shape[1]))
        self.logit_scale = torch.nn.Parameter(torch.tensor(4.605170185988092))  # ln(100)

    def encode_text(self, tokens):
        return self.table[tokens]


def install(num_classes=20, dim=512, seed=0):
    from . import sampler, scatter, spconv_cpu

    if not reference_available():
        raise RuntimeError(f"reference checkout not found at {REFERENCE_ROOT}")
    sp = _module("spconv")
    sp.pytorch = spconv_cpu
    sys.modules["spconv.pytorch"] = spconv_cpu
    _module("smooth_sampler", SmoothSampler=sampler.SmoothSampler)
    _module("torch_scatter", scatter=scatter.scatter)
    tg = _module("torch_geometric")
    tg.utils = _module("torch_geometric.utils", scatter=scatter.scatter)

    def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return torch.nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)

    timm = _module("timm")
    timm.models = _module("timm.models")
    timm.models.layers = _module("timm.models.layers", trunc_normal_=trunc_normal_)

    g = torch.Generator().manual_seed(seed)
    table = torch.randn(num_classes, dim, generator=g)
    table = table / table.norm(dim=-1, keepdim=True)

    def clip_load(name, device="cpu", download_root=None):
        # the reference asks for "cuda" whenever a GPU is visible (ponder_indoor_base.py:87-90) and
        # moves the tokens there: the stand-in's table has to live on the same device
        return _FakeClipModel(table.to(device)), None

    def clip_tokenize(prompts):
        # prompts are class-major with T templates per class; the reference then views the
        # encodings as (T, K, D) and averages over T (ponder_indoor_base.py:101-104), so token
        # p -> class p % K makes that average reproduce row k exactly.
        return torch.arange(len(prompts)) % num_classes

    _module("clip", load=clip_load, tokenize=clip_tokenize)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def load_reference_file(relpath, name=None):
    """Import ONE reference source file as a stand-alone module (bypasses package __init__ files
    that pull in dependencies which are absent here, e.g. ponder/datasets/__init__.py)."""
    import importlib.util

    path = os.path.join(REFERENCE_ROOT, relpath)
    name = name or "_ref_" + os.path.splitext(os.path.basename(relpath))[0]
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod  # the reference's Registry inspects the defining module
    spec.loader.exec_module(mod)
    return mod
