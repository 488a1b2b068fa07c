from .builder import (COLLIDERS, FIELDS, RENDERERS, SAMPLERS, build_collider, build_field,  # noqa
                      build_renderer, build_sampler)
from .rays import Frustums, RayBundle, RaySamples  # noqa: F401
from . import scene_colliders, ray_samplers  # noqa: F401  (registration)
from .fields import sdf_field  # noqa: F401
from .models import neus  # noqa: F401
