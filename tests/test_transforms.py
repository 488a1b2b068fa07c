"""Host transforms of the pre-training configs: identical samples to the reference's classes under
identical seeds (where the reference checkout is present), and against a committed fixture of the
reference's output (everywhere)."""
import copy
import os
import random

import numpy as np
import pytest
import torch

from oracle import ref_shims

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "transform_chain.npz")

# the ScanNet pre-training chain (configs/scannet/pretrain-ponder-spunet-v1m1-0-base.py:206-296)
SCANNET_CHAIN = [
    dict(type="CenterShift", apply_z=True, keys=["extrinsic"]),
    dict(type="RandomDropout", dropout_ratio=0.8, dropout_application_ratio=1.0),
    dict(type="RandomRotate", angle=[-1, 1], axis="z", center=[0, 0, 0], p=0.5, keys=["extrinsic"]),
    dict(type="RandomRotate", angle=[-1 / 64, 1 / 64], axis="x", p=0.5, keys=["extrinsic"]),
    dict(type="RandomRotate", angle=[-1 / 64, 1 / 64], axis="y", p=0.5, keys=["extrinsic"]),
    dict(type="RandomScale", scale=[0.9, 1.1], keys=["extrinsic"]),
    dict(type="RandomFlip", p=0.5, keys=["extrinsic"]),
    dict(type="GridSample", grid_size=0.02, hash_type="fnv", mode="train", return_grid_coord=True),
    dict(type="CenterShift", apply_z=False, keys=["extrinsic"]),
    dict(type="NormalizeColor"),
    dict(type="ShufflePoint"),
    dict(type="Add", keys_dict={"condition": "ScanNet"}),
    dict(type="ToTensor"),
    dict(type="Collect", keys=("coord", "grid_coord", "segment", "condition", "rgb", "depth",
                               "depth_scale"),
         stack_keys=("intrinsic", "extrinsic", "rgb", "depth", "semantic"),
         feat_keys=("color", "normal")),
]
# extra geometry transforms of the nuScenes chain (configs/nuscenes/...:140-168)
NUSCENES_EXTRA = [
    dict(type="RandomRotate", angle=[-0.25, 0.25], axis="z", center=[0, 0, 0], p=0.5,
         keys=["lidar2img", "lidar2cam"]),
    dict(type="RandomScale", scale=[0.9, 1.1], anisotropic=False, keys=["lidar2img", "lidar2cam"]),
    dict(type="RandomShift", shift=[0.5, 0.5, 0.5], keys=["lidar2img", "lidar2cam"]),
    dict(type="RandomFlip", p=0.5, keys=["lidar2img", "lidar2cam"]),
]


def raw_scene(seed=3):
    """What ScanNetRGBDDataset.get_data hands to the transforms (scannet.py:418-434)."""
    from ponderv2_amd.ponder.datasets import make_scene

    s = make_scene(seed, n_raw=8000, keep=1.0, num_views=2, image_hw=(12, 16), grid_size=0.01)
    return dict(coord=s["coord"].astype(np.float32), color=s["color"].astype(np.float32),
                normal=s["normal"].astype(np.float32), segment=s["segment"],
                instance=np.ones(len(s["coord"])) * -1, intrinsic=s["intrinsic"].astype(np.float64),
                extrinsic=s["extrinsic"].astype(np.float64), rgb=(s["rgb"] * 255).astype(np.float32),
                depth=s["depth"], semantic=s["semantic"].astype(np.int16), depth_scale=1.0 / 1000.0)


def run_chain(compose_cls, chain, data, seed):
    random.seed(seed)
    np.random.seed(seed)
    return compose_cls(chain)(copy.deepcopy(data))


def flatten(out):
    flat = {}
    for k, v in out.items():
        if torch.is_tensor(v):
            flat[k] = v.numpy()
        elif isinstance(v, np.ndarray):
            flat[k] = v
        elif isinstance(v, list):
            flat[k] = np.stack([np.asarray(a) for a in v])
        else:
            flat[k] = np.array(v)
    return flat


def assert_same(a, b):
    assert sorted(a) == sorted(b), (sorted(a), sorted(b))
    for k in a:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        assert x.dtype == y.dtype and x.shape == y.shape, (k, x.dtype, y.dtype, x.shape, y.shape)
        assert np.array_equal(x, y), k


@pytest.mark.skipif(not ref_shims.reference_available(), reason="reference checkout not present")
@pytest.mark.parametrize("seed", [0, 1, 2, 5])
def test_scannet_chain_identical_to_reference(seed):
    from ponderv2_amd.ponder.datasets import Compose

    ref_shims.install()
    T = ref_shims.load_reference_file("ponder/datasets/transform.py")
    data = raw_scene()
    assert_same(flatten(run_chain(Compose, SCANNET_CHAIN, data, seed)),
                flatten(run_chain(T.Compose, SCANNET_CHAIN, data, seed)))


@pytest.mark.skipif(not ref_shims.reference_available(), reason="reference checkout not present")
@pytest.mark.parametrize("seed", [0, 3])
def test_lidar_geometry_transforms_identical_to_reference(seed):
    from ponderv2_amd.ponder.datasets import Compose, make_sweep

    ref_shims.install()
    T = ref_shims.load_reference_file("ponder/datasets/transform.py")
    sweep = make_sweep(11, n_azimuth=120)
    data = {k: sweep[k] for k in ("coord", "strength", "segment", "lidar2img", "lidar2cam")}
    chain = NUSCENES_EXTRA + [dict(type="Copy", keys_dict={"coord": "origin_coord"})]
    assert_same(flatten(run_chain(Compose, chain, data, seed)),
                flatten(run_chain(T.Compose, chain, data, seed)))


def test_scannet_chain_matches_committed_reference_output():
    """Same comparison against the fixture written from the reference's classes by
    oracle/make_golden.py (transform_chain_case) - runs where the reference is absent."""
    from ponderv2_amd.ponder.datasets import Compose

    g = np.load(GOLDEN)
    out = flatten(run_chain(Compose, SCANNET_CHAIN, raw_scene(), int(g["seed"])))
    for k in out:
        if out[k].dtype.kind in "US":
            assert str(out[k]) == str(g[k])
            continue
        assert np.array_equal(out[k], g[k]), k


def test_cameras_follow_the_cloud():
    """After any geometry transform, projecting the moved points with the moved extrinsics gives
    the same camera coordinates as before (the invariant behind ``M @ inv(S)``)."""
    from ponderv2_amd.ponder.datasets import Compose

    data = raw_scene()
    pts = np.concatenate([data["coord"], np.ones((len(data["coord"]), 1), np.float32)], 1)
    before = np.einsum("vij,nj->vni", data["extrinsic"], pts.astype(np.float64))
    chain = [t for t in SCANNET_CHAIN[:7] if t["type"] != "RandomDropout"]
    random.seed(4)
    np.random.seed(4)
    out = Compose(chain)(copy.deepcopy(data))
    pts2 = np.concatenate([out["coord"], np.ones((len(out["coord"]), 1), np.float32)], 1)
    after = np.einsum("vij,nj->vni", out["extrinsic"], pts2.astype(np.float64))
    assert np.allclose(before[..., :3], after[..., :3], atol=2e-4)


@pytest.mark.skipif(not ref_shims.reference_available(), reason="reference checkout not present")
def test_point_collate_fn_identical_to_reference():
    from ponderv2_amd.ponder.datasets import point_collate_fn

    ref_shims.install()
    U = ref_shims.load_reference_file("ponder/datasets/utils.py")
    g = torch.Generator().manual_seed(0)

    def sample(n):
        return dict(coord=torch.randn(n, 3, generator=g), offset=torch.tensor([n]),
                    ray_offset=torch.tensor([n // 2]), name="scene%d" % n, rgb=torch.randn(1, 2, 4, generator=g),
                    nested=dict(a=torch.randn(n, 1, generator=g)), scalar=3)

    batch = [sample(5), sample(9), sample(4), sample(7)]
    for kwargs in (dict(), dict(max_point=20), dict(mix_prob=1.0)):
        random.seed(1)
        ours = point_collate_fn([dict(s, nested=dict(s["nested"])) for s in batch], **kwargs)
        random.seed(1)
        theirs = U.point_collate_fn([dict(s, nested=dict(s["nested"])) for s in batch], **kwargs)
        assert sorted(ours) == sorted(theirs)
        for k in ours:
            if torch.is_tensor(ours[k]):
                assert torch.equal(ours[k], theirs[k]) and ours[k].dtype == theirs[k].dtype, (k, kwargs)
            elif isinstance(ours[k], dict):
                assert torch.equal(ours[k]["a"], theirs[k]["a"])
            else:
                assert ours[k] == theirs[k], (k, kwargs)
    # list-style samples
    ours = __import__("ponderv2_amd.ponder.datasets.collate", fromlist=["collate_fn"]).collate_fn(
        [[torch.ones(3, 2)], [torch.ones(5, 2)]])
    theirs = U.collate_fn([[torch.ones(3, 2)], [torch.ones(5, 2)]])
    assert all(torch.equal(a, b) and a.dtype == b.dtype for a, b in zip(ours, theirs))
