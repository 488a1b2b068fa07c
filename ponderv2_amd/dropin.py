"""Zero-edit drop-in: make the reference's three native import boundaries resolve to this package.

The reference reaches native code through ``import spconv.pytorch as spconv``
(ponder/models/sparse_unet/spconv_unet_v1m1_base.py:11), ``from smooth_sampler import SmoothSampler``
(ponder/models/ponder/render_utils/fields/sdf_field.py:3) and ``from torch_scatter import scatter``
(ponder/models/ponder/ponder_indoor_base.py:10, ponder_outdoor_base.py:10).  ``install()`` aliases
those module names to the mirrors over libponderv2_hip.so, so an UNMODIFIED reference checkout
runs on MI355X (exercised by tests/test_gpu_zero_edit.py):

    python -c "import ponderv2_amd.dropin as d; d.install()" ...   # or from a sitecustomize.py
"""
import sys

_NAMES = ("spconv", "spconv.pytorch", "smooth_sampler", "torch_scatter")


def install(force=True):
    """Alias ``spconv`` / ``spconv.pytorch`` / ``smooth_sampler`` / ``torch_scatter`` in
    ``sys.modules``.  ``force=False`` leaves an already imported module of that name alone (a box
    that does have the CUDA wheels).  Returns the names that now point here."""
    from . import smooth_sampler, spconv, torch_scatter
    from .spconv import pytorch as spconv_pytorch

    mirrors = {"spconv": spconv, "spconv.pytorch": spconv_pytorch,
               "smooth_sampler": smooth_sampler, "torch_scatter": torch_scatter}
    done = []
    for name, module in mirrors.items():
        if force or name not in sys.modules:
            sys.modules[name] = module
            done.append(name)
    return done


def installed():
    """True when all four names resolve to this package's mirrors."""
    return all(getattr(sys.modules.get(n), "__name__", "").startswith("ponderv2_amd") for n in _NAMES)
