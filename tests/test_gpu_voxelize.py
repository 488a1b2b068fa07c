"""Device-side voxelisation ON the MI355X: ``grid_sample_torch`` /
``device_grid_sample`` (ponder/datasets/voxelize.py, the device half of the reference's GridSample,
datasets/transform.py:1078-1213) against the host transform - same hash bit patterns, same voxel set
in the same (unsigned key) order, same integer coordinates; fnv and ravel.  The signed 64-bit sort of
``key ^ 2^63`` has to reproduce numpy's unsigned order on the GPU's sort too, which only a run on
the device shows."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hash_type", ["fnv", "ravel"])
def test_device_grid_sample_same_voxels_as_host_transform_on_the_gpu(device, hash_type):
    from ponderv2_amd.ponder.datasets import GridSample, fnv_hash_vec, ravel_hash_vec
    from ponderv2_amd.ponder.datasets.voxelize import (fnv_hash_torch, grid_sample_torch,
                                                       ravel_hash_torch)

    rng = np.random.default_rng(0)
    arr = rng.integers(0, 500, size=(4000, 3))
    host_hash = (fnv_hash_vec if hash_type == "fnv" else ravel_hash_vec)(arr)
    dev_hash = (fnv_hash_torch if hash_type == "fnv" else ravel_hash_torch)(torch.from_numpy(arr).to(device))
    assert np.array_equal(dev_hash.cpu().numpy().view(np.uint64), host_hash)

    pts = rng.uniform(-1.5, 2.0, size=(200000, 3)).astype(np.float32)
    np.random.seed(3)
    host = GridSample(grid_size=0.05, hash_type=hash_type, mode="train", keys=("coord",),
                      return_grid_coord=True)(dict(coord=pts.copy()))
    idx, grid = grid_sample_torch(torch.from_numpy(pts).to(device), 0.05, hash_type)
    assert idx.is_cuda and grid.is_cuda
    idx, grid = idx.cpu().numpy(), grid.cpu().numpy()
    assert np.array_equal(grid, host["grid_coord"])                  # same voxels, same order
    own = np.floor(pts[idx] / 0.05).astype(int) - np.floor(pts / 0.05).astype(int).min(0)
    assert np.array_equal(own, grid)                                 # representative lies in its voxel
    assert len(np.unique(idx)) == len(idx)
    idx0, _ = grid_sample_torch(torch.from_numpy(pts).to(device), 0.05, hash_type,
                                pick=torch.zeros(len(idx), dtype=torch.int64, device=device))
    cell = np.floor(pts / 0.05).astype(int)
    _, first_member = np.unique(cell, axis=0, return_index=True)
    assert sorted(idx0.cpu().tolist()) == sorted(first_member.tolist())


def test_device_voxelisation_of_a_raw_batch_on_the_gpu_feeds_the_backbone(device):
    """A collated batch of RAW points, copied to the GPU and voxelised there: per scene the host
    GridSample's voxels in its order; and the result is a valid input of the sparse backbone (the
    rulebook builder rejects duplicate or out-of-range coordinates by construction of its tests)."""
    from ponderv2_amd import kernels as K
    from ponderv2_amd.ponder.datasets import collate_fn, make_scene
    from ponderv2_amd.ponder.datasets.voxelize import device_grid_sample

    kw = dict(n_raw=60000, num_views=1, image_hw=(12, 16))
    host = [make_scene(s, **kw) for s in (5, 6)]
    raw = collate_fn([make_scene(s, voxelize=False, **kw) for s in (5, 6)])
    raw = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in raw.items()}
    out = device_grid_sample(raw, grid_size=0.02, hash_type="fnv")
    assert out["grid_coord"].is_cuda
    ends = [0] + out["offset"].tolist()
    for b, h in enumerate(host):
        g = out["grid_coord"][ends[b]:ends[b + 1]].cpu().numpy()
        assert np.array_equal(g, h["grid_coord"])
        c = out["coord"][ends[b]:ends[b + 1]].cpu().numpy()
        cell = np.floor(c.astype(np.float64) / 0.02).astype(int)
        assert np.array_equal(cell - cell.min(0), g)
    batch_idx = torch.repeat_interleave(torch.arange(2, device=device),
                                        torch.diff(out["offset"], prepend=out["offset"].new_zeros(1)))
    coords = torch.cat([batch_idx[:, None], out["grid_coord"]], 1).int()
    rb = K.build_subm_rulebook(coords, 3)
    assert rb.n_out == coords.shape[0] and rb.n_pairs >= rb.n_out   # every voxel pairs with itself
    assert int(rb.kstart_host[14] - rb.kstart_host[13]) == rb.n_out  # centre offset: all rows, no duplicates


@pytest.mark.parametrize("hash_type", ["fnv", "ravel"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_voxelize_kernels_equal_the_sort_based_transform(device, hash_type, dtype):
    """csrc/voxelize.hip (hash-table de-duplication on the reference's key, sort of the unique keys
    only, segmented pick) against ``grid_sample_torch`` (two full sorts) and the host GridSample: the
    same voxels in the same order, and for equal draws the SAME representative of every voxel."""
    from ponderv2_amd.ponder.datasets import GridSample
    from ponderv2_amd.ponder.datasets.voxelize import grid_sample_device, grid_sample_torch

    rng = np.random.default_rng(1)
    pts = rng.uniform(-3.0, 2.5, size=(300000, 3)).astype(np.float32)
    pts[:5000] = pts[5000:10000]                       # exact duplicates as well
    coord = torch.from_numpy(pts).to(device, dtype)
    np.random.seed(3)
    host = GridSample(grid_size=0.04, hash_type=hash_type, mode="train", keys=("coord",),
                      return_grid_coord=True)(dict(coord=pts.copy()))
    n_vox = len(host["grid_coord"])
    draws = torch.from_numpy(rng.integers(0, 1 << 30, size=n_vox)).to(device)
    idx_t, grid_t = grid_sample_torch(coord, 0.04, hash_type)       # (its own draws: count known here)
    assert idx_t.numel() == n_vox
    idx_k, grid_k = grid_sample_device(coord, 0.04, hash_type, pick=draws)
    assert idx_k.dtype == torch.int64 and grid_k.dtype == torch.int64 and grid_k.shape == (n_vox, 3)
    assert np.array_equal(grid_k.cpu().numpy(), host["grid_coord"])  # same voxels, same order
    assert torch.equal(grid_k, grid_t)
    # the representative is the ((draw mod max count) mod count)-th member in point-index order
    cell = np.floor(pts.astype(np.float64) / 0.04).astype(np.int64)
    cell -= cell.min(0)
    got = idx_k.cpu().numpy()
    assert np.array_equal(cell[got], host["grid_coord"])
    key = (cell[:, 0] * (cell[:, 1].max() + 1) + cell[:, 1]) * (cell[:, 2].max() + 1) + cell[:, 2]
    by_key = np.argsort(key, kind="stable")
    sk = key[by_key]
    starts = np.flatnonzero(np.r_[True, sk[1:] != sk[:-1]])
    counts = np.diff(np.r_[starts, len(sk)])
    assert len(starts) == n_vox
    t = (draws.cpu().numpy() % counts.max())
    # voxels of `host` are in hash order, those of `starts` in ravel order: match through the coordinates
    want = {}
    for s0, c in zip(starts, counts):
        want[tuple(cell[by_key[s0]])] = (by_key[s0:s0 + c], c)
    for v in range(0, n_vox, max(1, n_vox // 2000)):
        members, c = want[tuple(host["grid_coord"][v])]
        assert got[v] == members[t[v] % c], v
    idx_r, _ = grid_sample_device(coord, 0.04, hash_type)             # device-drawn representatives
    assert np.array_equal(cell[idx_r.cpu().numpy()], host["grid_coord"])
