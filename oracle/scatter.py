"""torch-CPU restatement of torch_scatter.scatter(src, index, dim=0, reduce=..., out=...) as used at
ponder/models/ponder/ponder_indoor_base.py:214 (torch_scatter is not vendored by the reference)."""
import torch


def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
    assert dim == 0 and src.dim() == 2
    index = index.reshape(index.shape[0], -1)[:, 0].long()
    if out is None:
        if dim_size is None:
            dim_size = int(index.max()) + 1 if index.numel() else 0
        out = src.new_zeros((dim_size, src.shape[1]))
    summed = out.index_add(0, index, src)
    if reduce in ("sum", "add"):
        return summed
    assert reduce == "mean"
    count = torch.zeros(out.shape[0], dtype=src.dtype).index_add(
        0, index, torch.ones(index.shape[0], dtype=src.dtype))
    return summed / count.clamp(min=1).unsqueeze(1)
