"""Batch assembly for point-cloud samples (behaviour of ponder/datasets/utils.py: collate_fn
:16-57, point_collate_fn :60-72).

Rules, applied recursively per key: tensors are concatenated along dim 0 (point clouds have no
common length to stack on); strings are collected into a list; a list-style sample gets its row
count appended and cumulated; every dict key containing "offset" holds one length per sample and
becomes the running total of rows; anything else goes through torch's default collation.
``max_point`` skips samples that would push the batch over a point budget; ``mix_prob`` merges
neighbouring pairs of samples into one scene (Mix3D) by keeping every second offset.
"""
import random
from collections.abc import Mapping, Sequence
from functools import partial

import torch
from torch.utils.data.dataloader import default_collate


def _within_budget(samples, max_point):
    kept, total = [], 0
    for sample in samples:
        n = sample["coord"].shape[0]
        if total + n > max_point:
            print("SKIP: accum_num_points", total, "num_coords", n)
        else:
            total += n
            kept.append(sample)
    return kept


def _merge_dicts(samples):
    merged = {}
    for key in samples[0]:
        column = collate_fn([sample[key] for sample in samples])
        merged[key] = torch.cumsum(column, dim=0) if "offset" in key else column
    return merged


def _merge_lists(samples):
    for sample in samples:
        sample.append(torch.tensor([sample[0].shape[0]]))
    columns = [collate_fn(list(column)) for column in zip(*samples)]
    columns[-1] = torch.cumsum(columns[-1], dim=0).int()
    return columns


def collate_fn(batch, max_point=-1):
    if not isinstance(batch, Sequence):
        raise TypeError(f"{type(batch)} is not supported.")
    if max_point > 0:
        return collate_fn(_within_budget(batch, max_point))
    head = batch[0]
    if isinstance(head, torch.Tensor):
        return torch.cat(list(batch))
    if isinstance(head, str):  # before Sequence: a str is one too
        return list(batch)
    if isinstance(head, Sequence):
        return _merge_lists(batch)
    if isinstance(head, Mapping):
        return _merge_dicts(batch)
    return default_collate(batch)


def point_collate_fn(batch, mix_prob=0, max_point=-1):
    assert isinstance(batch[0], Mapping), "only dict samples are supported"
    return _mix_offsets(collate_fn(batch, max_point=max_point), mix_prob)


def _mix_offsets(batch, mix_prob):
    """Mix3D: with probability ``mix_prob`` neighbouring pairs of samples become one scene."""
    if "offset" in batch and random.random() < mix_prob:
        offset = batch["offset"]
        batch["offset"] = torch.cat([offset[1:-1:2], offset[-1:]], dim=0)
        if "offset_host" in batch:
            batch["offset_host"] = [int(v) for v in batch["offset"]]
        batch.pop("extent_host", None)  # per-scene hints of the unmixed scenes
    return batch


def _own_collate(samples, inner, mix_prob, max_point):
    if max_point > 0:
        samples = _within_budget(samples, max_point)
    batch = inner(samples)
    if mix_prob > 0:
        # Mix3D halves ``offset``; batches that carry per-scene TENSOR stacks next to the points
        # (views, poses, ray offsets: every pre-training batch) would be left with B scenes of poses
        # against B/2 scenes of points.  ``condition`` is not such a stack: the reference's PPT
        # fine-tuning configs collect it with mix_prob = 0.8 and the model reads condition[0] only
        # (configs/scannet/semseg-ppt-v1m1-0-sc-s3-st-spunet-lovasz-ft.py) - it passes through.
        per_scene = [k for k in ("rgb", "depth", "extrinsic", "intrinsic", "ray_offset")
                     if k in batch]
        if per_scene:
            raise ValueError(
                f"mix_prob={mix_prob} > 0 cannot be applied to batches with per-scene entries "
                f"{per_scene} (Mix3D merges neighbouring scenes' points only); set mix_prob=0 for "
                "this dataset")
        batch = _mix_offsets(batch, mix_prob)
    return batch


def loader_collate(dataset, mix_prob=0, max_point=-1):
    """The collate callable of a training loader.  The reference wires
    ``partial(point_collate_fn, mix_prob=cfg.mix_prob, max_point=cfg.max_point)`` into every loader
    (ponder/engines/train.py:243-258, ponder/datasets/dataloader.py:67-80) - the point budget is
    the out-of-memory guard of its pre-training configs.  Datasets whose samples need their own
    batch assembly (the synthetic scenes and lidar sweeps of this repository) declare ``collate_fn``
    and get the same budget and Mix3D handling around it."""
    own = getattr(dataset, "collate_fn", None)
    if own is None or own is point_collate_fn:
        return partial(point_collate_fn, mix_prob=mix_prob, max_point=max_point)
    return partial(_own_collate, inner=own, mix_prob=mix_prob, max_point=max_point)
