"""Host-side sample transforms of the pre-training configs, under the reference's registry names so
its ``data.train.transform`` lists build unchanged.

Restates ponder/datasets/transform.py for the transforms the pre-training configs name
(configs/scannet/pretrain-ponder-*.py :206-296, configs/nuscenes/pretrain-ponder-*.py :136-200):
Collect :28-52, Copy :56-70, ToTensor :74-98, Add :102-111, NormalizeColor :115-121, CenterShift
:173-194, RandomShift :198-214, RandomDropout :382-412, RandomRotate :416-471, RandomScale :534-557,
RandomFlip :561-584, ShufflePoint :1318-1337, Compose :1433-1443; GridSample lives in voxelize.py
and the lidar ray transforms in lidar.py.  Every transform makes the same ``random`` / ``np.random``
draws in the same order as the reference, so identical seeds give identical samples (pinned by
tests/test_transforms.py against the reference's classes and tests/golden/transform_chain.npz).

Two patterns recur and are factored out here instead of being spelled per transform:
  * a rigid / similarity change ``S`` of the point cloud is mirrored onto every camera matrix
    named in ``keys`` (world->camera extrinsics, lidar->camera / lidar->image) as ``M @ inv(S)``;
  * a row selection is applied to every per-point array that is present.
"""
import copy
import random
from collections.abc import Mapping, Sequence

import numpy as np
import torch

from ..utils.registry import Registry
from .lidar import PointRangeFilter, ProjectOnImage, RaySample
from .voxelize import GridSample

TRANSFORMS = Registry("transforms")
for _cls in (GridSample, PointRangeFilter, ProjectOnImage, RaySample):
    TRANSFORMS.register_module(module=_cls, name=_cls.__name__)

POINT_ARRAYS = ("coord", "color", "normal", "strength", "segment", "instance")


def _follow_cameras(data_dict, keys, S):
    """Cameras keep seeing the transformed cloud: M <- M @ inv(S) for every matrix of every key."""
    S_inv = np.linalg.inv(S)
    for key in keys:
        assert key in data_dict
        mats = data_dict[key]
        for i in range(len(mats)):
            mats[i] = mats[i] @ S_inv


def _select_rows(data_dict, idx, names=POINT_ARRAYS):
    for name in names:
        if name in data_dict:
            data_dict[name] = data_dict[name][idx]


def _translation(t):
    S = np.eye(4)
    S[:3, 3] = t
    return S


# ------------------------------------------------------------------ packaging
@TRANSFORMS.register_module()
class Collect:
    """Pick the keys a model consumes: ``keys`` as they are, ``stack_keys`` with a leading batch
    axis, one length tensor per ``offset_keys_dict`` entry (cumulated by the collate function), and
    ``<name>_keys=(a, b, ...)`` -> ``name`` = the float concatenation of a, b, ... along dim 1."""

    def __init__(self, keys, offset_keys_dict=None, stack_keys=(), **kwargs):
        self.keys, self.stack_keys = keys, stack_keys
        self.offset_keys = dict(offset="coord") if offset_keys_dict is None else offset_keys_dict
        self.concat = kwargs

    def __call__(self, data_dict):
        out = {key: data_dict[key] for key in self.keys}
        for key in self.stack_keys:
            out[key] = data_dict[key][None, ...]
        for key, source in self.offset_keys.items():
            out[key] = torch.tensor([data_dict[source].shape[0]])
        for name, parts in self.concat.items():
            assert isinstance(parts, Sequence)
            out[name.replace("_keys", "")] = torch.cat([data_dict[p].float() for p in parts], dim=1)
        return out


@TRANSFORMS.register_module()
class Copy:
    def __init__(self, keys_dict=None):
        self.keys_dict = (dict(coord="origin_coord", segment="origin_segment")
                          if keys_dict is None else keys_dict)

    def __call__(self, data_dict):
        for src, dst in self.keys_dict.items():
            v = data_dict[src]
            if isinstance(v, np.ndarray):
                data_dict[dst] = v.copy()
            elif isinstance(v, torch.Tensor):
                data_dict[dst] = v.clone().detach()
            else:
                data_dict[dst] = copy.deepcopy(v)
        return data_dict


@TRANSFORMS.register_module()
class ToTensor:
    """numpy -> torch, recursively: integers become int64, floats float32, bools stay bool."""

    def __call__(self, data):
        if isinstance(data, (torch.Tensor, str)):
            return data
        if isinstance(data, int):
            return torch.LongTensor([data])
        if isinstance(data, float):
            return torch.FloatTensor([data])
        if isinstance(data, np.ndarray):
            if np.issubdtype(data.dtype, bool):
                return torch.from_numpy(data)
            if np.issubdtype(data.dtype, np.integer):
                return torch.from_numpy(data).long()
            if np.issubdtype(data.dtype, np.floating):
                return torch.from_numpy(data).float()
        if isinstance(data, Mapping):
            return {k: self(v) for k, v in data.items()}
        if isinstance(data, Sequence):
            return [self(v) for v in data]
        raise TypeError(f"type {type(data)} cannot be converted to tensor.")


@TRANSFORMS.register_module()
class Add:
    def __init__(self, keys_dict=None):
        self.keys_dict = {} if keys_dict is None else keys_dict

    def __call__(self, data_dict):
        data_dict.update(self.keys_dict)
        return data_dict


@TRANSFORMS.register_module()
class NormalizeColor:
    """Point colours to [-1, 1], image colours to [0, 1]."""

    def __call__(self, data_dict):
        if "color" in data_dict:
            data_dict["color"] = data_dict["color"] / 127.5 - 1
        if "rgb" in data_dict:
            data_dict["rgb"] = (data_dict["rgb"] / 255.0).clip(0, 1)
        return data_dict


# ------------------------------------------------------------------ geometry
@TRANSFORMS.register_module()
class CenterShift:
    """Move the xy centre of the bounding box to the origin (and the floor to z = 0)."""

    def __init__(self, apply_z=True, keys=()):
        self.apply_z, self.keys = apply_z, keys

    def __call__(self, data_dict):
        lo, hi = data_dict["coord"].min(axis=0), data_dict["coord"].max(axis=0)
        shift = [(lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, lo[2] if self.apply_z else 0]
        data_dict["coord"] -= shift
        _follow_cameras(data_dict, self.keys, _translation(-np.array(shift)))
        return data_dict


@TRANSFORMS.register_module()
class RandomShift:
    def __init__(self, shift=(0.2, 0.2, 0.2), keys=()):
        self.shift, self.keys = shift, keys

    def __call__(self, data_dict):
        t = np.random.normal(scale=self.shift, size=3)
        data_dict["coord"] += t
        _follow_cameras(data_dict, self.keys, _translation(t))
        return data_dict


@TRANSFORMS.register_module()
class RandomDropout:
    """With probability ``dropout_application_ratio`` keep a random (1 - dropout_ratio) subset."""

    def __init__(self, dropout_ratio=0.2, dropout_application_ratio=0.5):
        self.dropout_ratio = dropout_ratio
        self.dropout_application_ratio = dropout_application_ratio

    def __call__(self, data_dict):
        if random.random() < self.dropout_application_ratio:
            n = len(data_dict["coord"])
            idx = np.random.choice(n, int(n * (1 - self.dropout_ratio)), replace=False)
            if "sampled_index" in data_dict:  # data-efficient ScanNet: labelled points always stay
                idx = np.unique(np.append(idx, data_dict["sampled_index"]))
                labelled = np.zeros_like(data_dict["segment"]).astype(bool)
                labelled[data_dict["sampled_index"]] = True
                data_dict["sampled_index"] = np.where(labelled[idx])[0]
            _select_rows(data_dict, idx)
        return data_dict


def _axis_rotation(axis, angle):
    c, s = np.cos(angle), np.sin(angle)
    if axis == "x":
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    if axis == "y":
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    if axis == "z":
        return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    raise NotImplementedError(axis)


@TRANSFORMS.register_module()
class RandomRotate:
    """Rotate by a uniform angle (in units of pi) about ``axis`` through ``center`` (default: the
    bounding-box centre); normals rotate along."""

    def __init__(self, angle=None, center=None, axis="z", always_apply=False, p=0.5, keys=()):
        self.angle = [-1, 1] if angle is None else angle
        self.axis, self.center, self.keys = axis, center, keys
        self.p = 1 if always_apply else p

    def __call__(self, data_dict):
        if random.random() > self.p:
            return data_dict
        R = _axis_rotation(self.axis, np.random.uniform(self.angle[0], self.angle[1]) * np.pi)
        center = self.center
        if center is None:
            lo, hi = data_dict["coord"].min(axis=0), data_dict["coord"].max(axis=0)
            center = [(lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, (lo[2] + hi[2]) / 2]
        data_dict["coord"] -= center
        data_dict["coord"] = np.dot(data_dict["coord"], np.transpose(R))
        data_dict["coord"] += center
        S_rot = np.eye(4)
        S_rot[:3, :3] = R
        _follow_cameras(data_dict, self.keys,
                        _translation(np.array(center)) @ S_rot @ _translation(-np.array(center)))
        if "normal" in data_dict:
            data_dict["normal"] = np.dot(data_dict["normal"], np.transpose(R))
        return data_dict


@TRANSFORMS.register_module()
class RandomScale:
    def __init__(self, scale=None, anisotropic=False, keys=()):
        self.scale = [0.95, 1.05] if scale is None else scale
        self.anisotropic, self.keys = anisotropic, keys

    def __call__(self, data_dict):
        scale = np.random.uniform(self.scale[0], self.scale[1], 3 if self.anisotropic else 1)
        data_dict["coord"] *= scale
        S = np.eye(4)
        S[:3, :3] *= scale
        _follow_cameras(data_dict, self.keys, S)
        if "depth_scale" in data_dict:
            assert not self.anisotropic, "anisotropic not supported yet."
            data_dict["depth_scale"] *= scale
        return data_dict


@TRANSFORMS.register_module()
class RandomFlip:
    """Mirror x and, independently, y with probability ``p`` each."""

    def __init__(self, p=0.5, keys=()):
        self.p, self.keys = p, keys

    def __call__(self, data_dict):
        S = np.eye(4)
        for axis in (0, 1):
            if np.random.rand() < self.p:
                data_dict["coord"][:, axis] = -data_dict["coord"][:, axis]
                S[axis, axis] = -1
                if "normal" in data_dict:
                    data_dict["normal"][:, axis] = -data_dict["normal"][:, axis]
        _follow_cameras(data_dict, self.keys, S)
        return data_dict


@TRANSFORMS.register_module()
class ShufflePoint:
    def __call__(self, data_dict):
        assert "coord" in data_dict
        order = np.arange(data_dict["coord"].shape[0])
        np.random.shuffle(order)
        _select_rows(data_dict, order, ("coord", "grid_coord", "displacement", "color", "normal",
                                        "segment", "instance"))
        return data_dict


class Compose:
    def __init__(self, cfg=None):
        self.cfg = [] if cfg is None else cfg
        self.transforms = [TRANSFORMS.build(t) for t in self.cfg]

    def __call__(self, data_dict):
        for t in self.transforms:
            data_dict = t(data_dict)
        return data_dict
