"""Element type of the sparse backbone's activations.

The reference trains with ``enable_amp = True`` (configs/scannet/pretrain-ponder-spunet-v1m1-0-base.py
:12): ``torch.cuda.amp.autocast`` around the model call (ponder/engines/train.py:183-196), under
which spconv runs its convolutions on 16-bit features.  Here the same mode stores the active-voxel
feature matrices in bf16 / fp16 between the layers of the sparse U-Net - the 16-bit MFMA kernels
of csrc/sparse_conv16.hip read and write them, BatchNorm statistics and every accumulation stay
fp32 (csrc/rownorm.hip), the master weights stay fp32.

``sparse_dtype()`` answers "which 16-bit type, if any" for the code that creates those matrices
(rownorm.fused_bn): the ambient autocast dtype, or whatever an enclosing ``sparse_activations``
block says - the pretraining models run their forward with autocast disabled (their render head is
fp32 by construction) and scope the mode to the backbone with that context manager.
``PV2_SPARSE_AMP=0`` keeps fp32 feature matrices (and the fp32 kernels) under autocast.
"""
import contextlib
import os

import torch

HALF_DTYPES = (torch.bfloat16, torch.float16)
ENABLED = os.environ.get("PV2_SPARSE_AMP", "1") != "0"

_UNSET = object()
_scoped = _UNSET


def sparse_dtype():
    """torch.bfloat16 / torch.float16 when sparse activations are to be stored in 16 bits, else None."""
    if not ENABLED:
        return None
    if _scoped is not _UNSET:
        return _scoped if _scoped in HALF_DTYPES else None
    if torch.is_autocast_enabled():
        dt = torch.get_autocast_gpu_dtype()
        return dt if dt in HALF_DTYPES else None
    return None


@contextlib.contextmanager
def sparse_activations(dtype):
    """Inside the block ``sparse_dtype()`` is ``dtype`` (None: fp32) whatever the autocast state."""
    global _scoped
    saved, _scoped = _scoped, dtype
    try:
        yield
    finally:
        _scoped = saved
