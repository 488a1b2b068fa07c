"""offset <-> batch helpers (ponder/models/utils.py:11-30), computed on the device with no Python
loop over scenes and no host round trip."""
import torch


def offset2batch(offset: torch.Tensor, n: int = None) -> torch.Tensor:
    """(B,) cumulative point counts -> (N,) int64 scene index of every point.  ``n`` = the point
    count when the caller knows it (the row count of the point tensors): without it
    ``repeat_interleave`` reads the output size back from the device - a host stall."""
    offset = offset.long()
    counts = torch.diff(offset, prepend=offset.new_zeros(1))
    return torch.repeat_interleave(torch.arange(offset.numel(), device=offset.device), counts,
                                   output_size=n)


def batch2offset(batch: torch.Tensor) -> torch.Tensor:
    return torch.cumsum(batch.bincount(), dim=0).long()


def offsets_host(data_dict):
    """Python list of the cumulative point counts.  Collate functions that still hold the counts
    on the host can pass ``offset_host``; otherwise this is ONE device->host read per batch, shared
    by every consumer (the reference does one per use, e.g. models/utils.py:11-26)."""
    oh = data_dict.get("offset_host")
    if oh is None:
        oh = [int(v) for v in data_dict["offset"].tolist()]
        data_dict["offset_host"] = oh
    return oh
