"""Drop-in for the one torch_scatter entry point on the hot path:
``scatter(src, index, dim=0, reduce="mean"|"sum", out=...)`` as called at
ponder/models/ponder/ponder_indoor_base.py:214 and ponder_outdoor_base.py:204 (index is an
(M,1) or (M,) int64 tensor of destination rows, broadcast over the channel axis).
"""
import torch

from .kernels import ScatterRowsFunction


def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
    if dim != 0 or src.dim() != 2:
        raise NotImplementedError("ponderv2_amd.torch_scatter.scatter: only dim=0 on (M,C) rows")
    if reduce not in ("sum", "add", "mean"):
        raise NotImplementedError(f"reduce={reduce!r} is not on the PonderV2 hot path")
    index = index.reshape(index.shape[0], -1)[:, 0]
    if out is None:
        if dim_size is None:
            dim_size = int(index.max().item()) + 1 if index.numel() else 0
        out = torch.zeros((dim_size, src.shape[1]), dtype=src.dtype, device=src.device)
    return ScatterRowsFunction.apply(out, src, index.long(), reduce == "mean")
