"""Registries of the render head.  Same public names as the reference's
ponder/models/ponder/render_utils/builder.py (RENDERERS / FIELDS / COLLIDERS / SAMPLERS and their
``build_*`` helpers), produced by one factory."""
from ....utils.registry import Registry


def _registry_with_builder(name):
    registry = Registry(name)

    def build(cfg, **default_args):
        return registry.build(cfg, default_args=default_args)

    build.__doc__ = f"Instantiate ``cfg['type']`` from the {name} registry."
    return registry, build


RENDERERS, build_renderer = _registry_with_builder("renderers")
FIELDS, build_field = _registry_with_builder("fields")
COLLIDERS, build_collider = _registry_with_builder("colliders")
SAMPLERS, build_sampler = _registry_with_builder("samplers")
