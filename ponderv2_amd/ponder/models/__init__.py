from .builder import MODELS, MODULES, build_model  # noqa: F401
from .losses import LOSSES, build_criteria  # noqa: F401
from .sparse_unet import SpUNetBase  # noqa: F401
from .ponder import PonderIndoor, UNet3Dv1m2  # noqa: F401
