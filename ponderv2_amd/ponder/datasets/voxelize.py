"""Voxelisation transform that defines the voxel set the backbone sees.

Restates ``GridSample`` of ponder/datasets/transform.py:1078-1213 (train mode :1103-1145; the two
hash functions :1180-1213).  Integer work: the 64-bit hashes wrap mod 2^64 exactly as numpy's
uint64 arithmetic does in the reference, and the same ``np.argsort`` / ``np.random.randint`` calls
are made in the same order so identical seeds pick identical representatives.
"""
import numpy as np

_FNV_OFFSET = np.uint64(14695981039346656037)
_FNV_PRIME = np.uint64(1099511628211)


def fnv_hash_vec(arr):
    """Per-coordinate FNV (multiply, then xor) over the columns of an int array, uint64."""
    assert arr.ndim == 2
    a = arr.astype(np.uint64)
    h = np.full(a.shape[0], _FNV_OFFSET, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for j in range(a.shape[1]):
            h = h * _FNV_PRIME
            h = np.bitwise_xor(h, a[:, j])
    return h


def ravel_hash_vec(arr):
    """Row-major ravel of (coords - min) with per-axis extent max+1, uint64."""
    assert arr.ndim == 2
    a = (arr - arr.min(0)).astype(np.uint64)
    ext = a.max(0).astype(np.uint64) + np.uint64(1)
    keys = np.zeros(a.shape[0], dtype=np.uint64)
    for j in range(a.shape[1] - 1):
        keys += a[:, j]
        keys *= ext[j + 1]
    keys += a[:, -1]
    return keys


class GridSample:
    """One random point per occupied voxel (train mode) + integer grid coordinates."""

    def __init__(self, grid_size=0.05, hash_type="fnv", mode="train",
                 keys=("coord", "color", "normal", "segment"), return_grid_coord=False,
                 return_min_coord=False):
        assert mode == "train", "only the training mode is on the pre-training path"
        self.grid_size = grid_size
        self.hash = fnv_hash_vec if hash_type == "fnv" else ravel_hash_vec
        self.keys = keys
        self.return_grid_coord, self.return_min_coord = return_grid_coord, return_min_coord

    def __call__(self, data_dict):
        scaled = data_dict["coord"] / np.array(self.grid_size)
        grid_coord = np.floor(scaled).astype(int)
        min_coord = grid_coord.min(0) * np.array(self.grid_size)
        grid_coord -= grid_coord.min(0)
        key = self.hash(grid_coord)
        idx_sort = np.argsort(key)
        _, count = np.unique(key[idx_sort], return_counts=True)
        first = np.cumsum(np.insert(count, 0, 0)[0:-1])
        pick = first + np.random.randint(0, count.max(), count.size) % count
        idx_unique = idx_sort[pick]
        if self.return_grid_coord:
            data_dict["grid_coord"] = grid_coord[idx_unique]
        if self.return_min_coord:
            data_dict["min_coord"] = min_coord.reshape([1, 3])
        for k in self.keys:
            data_dict[k] = data_dict[k][idx_unique]
        return data_dict
