from .spconv_unet_v1m1_base import BasicBlock, SpUNetBase  # noqa: F401
