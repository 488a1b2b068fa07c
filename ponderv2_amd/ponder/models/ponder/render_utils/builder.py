"""Registries of the render head (names as in ponder/models/ponder/render_utils/builder.py:3-26)."""
from ....utils.registry import Registry

RENDERERS = Registry("renderers")
FIELDS = Registry("fields")
COLLIDERS = Registry("colliders")
SAMPLERS = Registry("samplers")


def build_renderer(cfg, **kwargs):
    return RENDERERS.build(cfg, default_args=kwargs)


def build_field(cfg, **kwargs):
    return FIELDS.build(cfg, default_args=kwargs)


def build_collider(cfg, **kwargs):
    return COLLIDERS.build(cfg, default_args=kwargs)


def build_sampler(cfg, **kwargs):
    return SAMPLERS.build(cfg, default_args=kwargs)
