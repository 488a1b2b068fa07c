from .unet3d import SimpleConv3D, UNet3Dv1m2  # noqa: F401
from .ponder_indoor_base import PonderIndoor  # noqa: F401
from .ponder_outdoor_base import PonderOutdoor  # noqa: F401
