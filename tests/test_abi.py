"""The C-ABI shared library loads on a CPU-only box and exports every symbol that
include/ponderv2_hip.h declares (no compute calls here)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "ponderv2_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pv2_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from ponderv2_amd import _lib

    handle = _lib.lib()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(handle, n), f"{n} declared in the header but not exported"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert handle.pv2_abi_version() == _lib.ABI_VERSION == 16


def test_ops_reject_cpu_tensors():
    """No CPU fallback: product ops fail loudly on host tensors."""
    import pytest
    import torch

    from ponderv2_amd import kernels as K
    from ponderv2_amd.smooth_sampler import SmoothSampler

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        K.build_subm_rulebook(torch.zeros((4, 4), dtype=torch.int32), 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        SmoothSampler.apply(torch.zeros(1, 2, 2, 2, 2), torch.zeros(1, 1, 1, 1, 3))
