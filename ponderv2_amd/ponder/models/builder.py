"""Model registries (same names as ponder/models/builder.py:10-16)."""
from ..utils.registry import Registry

MODELS = Registry("models")
MODULES = Registry("modules")


def build_model(cfg):
    return MODELS.build(cfg)
