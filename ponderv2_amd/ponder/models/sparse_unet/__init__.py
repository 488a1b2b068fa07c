from .spconv_unet_v1m1_base import BasicBlock, SpUNetBase  # noqa: F401
from .spconv_unet_v1m3_pdnorm import PDBatchNorm  # noqa: F401
