"""Batch assembly for point-cloud samples (ponder/datasets/utils.py: collate_fn :16-57,
point_collate_fn :60-72).

Per-point tensors of different samples are concatenated along dim 0; every key containing
"offset" holds one length per sample and becomes the cumulative row count; strings are listed;
anything else goes through torch's default collation.  ``mix_prob`` merges neighbouring pairs of
samples into one scene (Mix3D) by dropping every other offset.
"""
import random
from collections.abc import Mapping, Sequence

import torch
from torch.utils.data.dataloader import default_collate


def collate_fn(batch, max_point=-1):
    if not isinstance(batch, Sequence):
        raise TypeError(f"{type(batch)} is not supported.")
    if max_point > 0:  # drop samples that would push the batch over the point budget
        kept, total = [], 0
        for sample in batch:
            n = sample["coord"].shape[0]
            if total + n > max_point:
                print("SKIP: accum_num_points", total, "num_coords", n)
                continue
            total += n
            kept.append(sample)
        return collate_fn(kept)
    first = batch[0]
    if isinstance(first, torch.Tensor):
        return torch.cat(list(batch))
    if isinstance(first, str):
        return list(batch)
    if isinstance(first, Sequence):  # list-style samples: append the row count, cumulate it
        for sample in batch:
            sample.append(torch.tensor([sample[0].shape[0]]))
        out = [collate_fn(column) for column in zip(*batch)]
        out[-1] = torch.cumsum(out[-1], dim=0).int()
        return out
    if isinstance(first, Mapping):
        out = {key: collate_fn([sample[key] for sample in batch]) for key in first}
        for key in out:
            if "offset" in key:
                out[key] = torch.cumsum(out[key], dim=0)
        return out
    return default_collate(batch)


def point_collate_fn(batch, mix_prob=0, max_point=-1):
    assert isinstance(batch[0], Mapping), "only dict samples are supported"
    batch = collate_fn(batch, max_point=max_point)
    if "offset" in batch and random.random() < mix_prob:
        batch["offset"] = torch.cat([batch["offset"][1:-1:2], batch["offset"][-1].unsqueeze(0)],
                                    dim=0)
    return batch
