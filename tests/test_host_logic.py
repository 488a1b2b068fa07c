"""Host logic: config loader, registries, synthetic data contract, model construction."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_shims

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_inheritance_and_overrides(tmp_path):
    from ponderv2_amd.ponder.utils.config import Config, DictAction

    (tmp_path / "base.py").write_text("a = 1\nmodel = dict(type='X', depth=2, head=dict(c=3))\nhooks=[dict(t=1), dict(t=2)]\n")
    (tmp_path / "child.py").write_text(
        "_base_ = ['base.py']\nmodel = dict(depth=5, head=dict(_delete_=True, d=4))\nb = a if False else 7\n")
    cfg = Config.fromfile(str(tmp_path / "child.py"))
    assert cfg.a == 1 and cfg.b == 7 and cfg.model.type == "X" and cfg.model.depth == 5
    assert cfg.model.head == dict(d=4)
    cfg.merge_from_dict({"model.depth": 9, "hooks.1.t": 5, "new.key": "v"})
    assert cfg.model.depth == 9 and cfg.hooks[1].t == 5 and cfg.new.key == "v"
    ns = {}
    exec(cfg.pretty_text, ns)
    assert ns["model"]["depth"] == 9
    assert DictAction._parse("[1,2,(a,b)]") == [1, 2, ("a", "b")] and DictAction._parse("true") is True


@pytest.mark.skipif(not ref_shims.reference_available(), reason="reference checkout not present")
@pytest.mark.parametrize("rel", ["configs/scannet/pretrain-ponder-spunet-v1m1-0-base.py",
                                 "configs/nuscenes/pretrain-ponder-spunet-v1m1-0-base.py",
                                 "configs/scannet/pretrain-ponder-ppt-v1m1-0-sc-s3-st-spunet.py",
                                 "configs/structured3d/pretrain-ponder-spunet-v1m1-0-base.py"])
def test_reference_configs_load_unchanged(rel):
    from ponderv2_amd.ponder.utils.config import Config

    path = os.path.join(ref_shims.REFERENCE_ROOT, rel)
    if not os.path.exists(path):
        pytest.skip("config not in this checkout")
    cfg = Config.fromfile(path)
    assert cfg.model.type.startswith("Ponder") and "hooks" in cfg and cfg.train.type


@pytest.mark.skipif(not ref_shims.reference_available(), reason="reference checkout not present")
def test_reference_scannet_model_section_builds_with_reference_param_counts():
    """MODELS.build on the unchanged reference config: parameter counts of SURVEY.md 2.2."""
    from ponderv2_amd.ponder.models import build_model
    from ponderv2_amd.ponder.utils.config import Config

    cfg = Config.fromfile(os.path.join(
        ref_shims.REFERENCE_ROOT, "configs/scannet/pretrain-ponder-spunet-v1m1-0-base.py"))
    m = build_model(cfg.model)
    n = lambda mod: sum(p.numel() for p in mod.parameters())  # noqa: E731
    assert n(m.backbone) == 39155904 and n(m.proj_net) == 2991520
    assert n(m.renderer) == 143687 and n(m.proj_head) == 49664


def test_registry_errors():
    from ponderv2_amd.ponder.utils.registry import Registry

    R = Registry("things")

    @R.register_module("a-name")
    class A:
        def __init__(self, x=1):
            self.x = x

    assert R.build(dict(type="a-name", x=4)).x == 4 and "a-name" in R
    with pytest.raises(KeyError):
        R.build(dict(type="missing"))
    with pytest.raises(KeyError):
        R.register_module("a-name")(A)


def test_synthetic_batch_contract():
    from ponderv2_amd.ponder.datasets import collate_fn, make_scene

    kw = dict(n_raw=8000, num_views=3, image_hw=(24, 32))
    b = collate_fn([make_scene(0, **kw), make_scene(1, **kw)])
    n = int(b["offset"][-1])
    assert b["coord"].shape == (n, 3) and b["grid_coord"].dtype == torch.int64
    assert b["feat"].shape == (n, 6) and b["rgb"].shape == (2, 3, 24, 32, 3)
    assert b["depth"].shape == (2, 3, 24, 32) and b["extrinsic"].shape == (2, 3, 4, 4)
    assert b["semantic"].dtype == torch.int64 and b["condition"] == ["ScanNet", "ScanNet"]
    # one point per voxel, per scene
    for lo, hi in zip([0, int(b["offset"][0])], b["offset"].tolist()):
        g = b["grid_coord"][lo:hi].numpy()
        assert len(np.unique(g, axis=0)) == len(g)
    # same seed -> same scene
    again = make_scene(0, **kw)
    assert np.array_equal(again["grid_coord"], make_scene(0, **kw)["grid_coord"])
    # depth is consistent with the geometry: back-projected pixels land on the point cloud's box
    s = make_scene(0, **kw)
    E, Kc = s["extrinsic"][0], s["intrinsic"][0]
    z = s["depth"][0] / 1000.0
    ys, xs = np.nonzero(z > 0)
    cam = np.stack([(xs - Kc[0, 2]) / Kc[0, 0] * z[ys, xs], (ys - Kc[1, 2]) / Kc[1, 1] * z[ys, xs],
                    z[ys, xs]], 1)
    world = (cam - E[:3, 3]) @ E[:3, :3]
    assert world.min() > -0.01 and (world.max(0) < np.array([6.01, 5.01, 2.61])).all()


def test_sparse_tensor_rejects_grids_beyond_the_rulebook_key_range():
    """the rulebook kernels pack (b, x+16, y+16, z+16) into 16-bit fields; a larger
    grid or batch must fail loudly on the host instead of aliasing voxels in the hash."""
    from ponderv2_amd.spconv.pytorch import MAX_BATCH_SIZE, MAX_SPATIAL_DIM, SparseConvTensor

    feat, idx = torch.zeros(2, 4), torch.zeros(2, 4, dtype=torch.int32)
    SparseConvTensor(feat, idx, [MAX_SPATIAL_DIM, 10, 10], MAX_BATCH_SIZE)
    with pytest.raises(ValueError):
        SparseConvTensor(feat, idx, [10, MAX_SPATIAL_DIM + 1, 10], 1)
    with pytest.raises(ValueError):
        SparseConvTensor(feat, idx, [10, 10, 10], MAX_BATCH_SIZE + 1)


def test_offset2batch():
    from ponderv2_amd.ponder.models.utils import batch2offset, offset2batch

    off = torch.tensor([3, 3, 7])
    b = offset2batch(off)
    assert b.tolist() == [0, 0, 0, 2, 2, 2, 2]
    assert batch2offset(torch.tensor([0, 0, 1, 1, 1])).tolist() == [2, 5]


# ------------------------------------------------------------------ outdoor (nuScenes-shaped) side
def test_lidar_batch_contract():
    """Keys / dtypes of the nuScenes Collect (configs/nuscenes/...-0-base.py:186-199)."""
    from ponderv2_amd.ponder.datasets import lidar_collate_fn, make_lidar_scene

    kw = dict(n_azimuth=200, point_nsample=24)
    b = lidar_collate_fn([make_lidar_scene(3, **kw), make_lidar_scene(4, **kw)])
    n, r = int(b["offset"][-1]), int(b["ray_offset"][-1])
    assert b["coord"].shape == (n, 3) and b["coord"].dtype == torch.float32
    assert b["grid_coord"].shape == (n, 3) and b["grid_coord"].dtype == torch.int64
    assert b["feat"].shape == (n, 4)  # [coord, strength]
    assert b["ray_start"].shape == (r, 3) and b["ray_end"].shape == (r, 3)
    assert b["ray_start"].dtype == torch.float32  # ToTensor casts every float array to f32
    assert b["offset_host"] == b["offset"].tolist() and b["ray_offset_host"] == b["ray_offset"].tolist()
    assert r == 2 * 6 * 24 and b["condition"] == ["nuScenes", "nuScenes"]
    # every ray ends on a voxelised lidar return and starts at one of the 6 camera centres
    ends = {tuple(np.round(p, 4)) for p in b["ray_end"].numpy().tolist()}
    pts = {tuple(np.round(p, 4)) for p in b["coord"].numpy().tolist()}
    assert ends <= pts
    assert len({tuple(np.round(p, 4)) for p in b["ray_start"].numpy().tolist()}) == 6
    assert (b["grid_coord"].min(0).values >= 0).all()


def test_lidar_transforms_match_reference_golden():
    """PointRangeFilter -> GridSample(ravel) -> ProjectOnImage -> RaySample restated in
    datasets/lidar.py reproduce what the reference's classes (transform.py:232-378) produced from
    the same sweep and numpy seed (fixture written by oracle/make_golden.py lidar_transform_case)."""
    from ponderv2_amd.ponder.datasets import (GridSample, PointRangeFilter, ProjectOnImage,
                                              RaySample, make_sweep)

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "lidar_transforms.npz"))
    data = make_sweep(7, n_azimuth=200)
    np.random.seed(7)
    data = PointRangeFilter(point_cloud_range=(-27.0, -27.0, -5.0, 27.0, 27.0, 3.0), padding=0.1)(data)
    data = GridSample(grid_size=0.1, hash_type="ravel", mode="train",
                      keys=("coord", "strength", "segment"), return_grid_coord=True)(data)
    data = ProjectOnImage(filter_overlap=True, close_radius=3.0)(data)
    assert [int(m.sum()) for m in data["img_proj_mask"]] == g["n_proj"].tolist()
    data = RaySample(point_nsample=24, fetch_color=False, fetch_segment=True)(data)
    assert np.array_equal(data["grid_coord"], g["grid_coord"])
    for k in ("ray_start", "ray_end", "ray_segment"):
        assert np.array_equal(data[k], g[k]), k


def test_project_on_image_keeps_nearest_point_per_pixel():
    from ponderv2_amd.ponder.datasets import ProjectOnImage

    K = np.eye(4)
    K[0, 0] = K[1, 1] = 100.0
    K[0, 2] = K[1, 2] = 50.0
    img = [np.zeros((100, 100, 3))]
    # camera looks down +z; three points on one pixel ray at depths 9, 4, 6 and one behind
    coord = np.array([[0.0, 0.0, 9.0], [0.0, 0.0, 4.0], [0.001, 0.0, 6.0], [0.0, 0.0, -2.0],
                      [10.0, 0.0, 5.0]], dtype=np.float32)
    out = ProjectOnImage(filter_overlap=True, close_radius=0.0)(
        dict(coord=coord, img=img, lidar2img=np.stack([K])))
    # point 0 sits exactly on the sensor axis: radius 0 is not > close_radius -> dropped like 1
    assert out["img_proj_mask"][0].tolist() == [False, False, True, False, False]
    out = ProjectOnImage(filter_overlap=True, close_radius=-1.0)(
        dict(coord=coord, img=img, lidar2img=np.stack([K])))
    assert out["img_proj_mask"][0].tolist() == [False, True, False, False, False]


def test_multi_dataset_loader_schedule():
    """Ratio interleave + epoch length of MultiDatasetDataloader (datasets/dataloader.py:25-117)."""
    from ponderv2_amd.ponder.datasets import ConcatDataset, MultiDatasetDataloader

    class Toy(torch.utils.data.Dataset):
        def __init__(self, tag, n, loop):
            self.tag, self.n, self.loop = tag, n, loop

        def __len__(self):
            return self.n * self.loop

        def __getitem__(self, i):
            return self.tag

        collate_fn = staticmethod(lambda items: items)

    a, b = Toy("a", 5, 2), Toy("b", 3, 1)
    cat = ConcatDataset([a, b], loop=2)
    assert len(cat) == (10 + 3) * 2 and cat[0] == "a" and cat[10] == "b" and cat[13] == "a"
    a, b = Toy("a", 5, 2), Toy("b", 3, 1)
    loader = MultiDatasetDataloader(ConcatDataset([a, b], loop=2), 1, 0, seed=1)
    tags = [batch[0] for batch in loader]
    # main dataset: 5 samples x concat loop 2 = 10 batches; ratio 2:1 -> a a b repeated
    assert tags == list("aab" * 5) and len(loader) == len(tags) == 15
    assert set(len(batch) for batch in loader) == {1}


def test_block_masking_keeps_the_reference_set():
    """mask_blocks: per scene exactly round(n_blocks*(1-ratio)) blocks keep their features, and
    with injected draws the kept set equals argsort(draw)[:n_keep] (ponder_outdoor_base.py:94-105)."""
    from ponderv2_amd.ponder.models.ponder.masking import mask_blocks

    g = torch.Generator().manual_seed(0)
    grid = torch.randint(0, 64, (4000, 3), generator=g)
    offset = torch.tensor([1500, 4000])
    feat = torch.randn(4000, 4, generator=g)
    token = torch.full((1, 4), 7.0, requires_grad=True)
    batch = torch.repeat_interleave(torch.arange(2), torch.tensor([1500, 2500]))
    block = torch.cat([batch[:, None], torch.div(grid, 8).int()], 1)
    ublock, inv = block.unique(dim=0, return_inverse=True)
    draws = torch.rand(len(ublock), generator=g)
    out = mask_blocks(grid, feat, offset, 8, 0.8, token, rand=draws)
    masked = (out == 7.0).all(1)
    for s in range(2):
        ids = torch.nonzero(ublock[:, 0] == s)[:, 0]
        n_keep = round(len(ids) * (1 - 0.8))
        kept_ref = set(ids[draws[ids].argsort()[:n_keep]].tolist())
        kept = set(inv[(batch == s) & ~masked].unique().tolist())
        assert kept == kept_ref
    assert torch.equal(out[~masked], feat[~masked])
    out.sum().backward()
    assert torch.allclose(token.grad, torch.full((1, 4), float(masked.sum())))


def test_block_masking_with_its_own_draws():
    """mask_blocks drawing its own random keys (no compaction, no host read): a block is kept or
    masked as a whole, exactly round(n_blocks*(1-ratio)) blocks per scene are kept, scenes do not
    mix, different seeds give different sets; negative block coordinates and an empty-looking last
    scene (one voxel) are handled."""
    from ponderv2_amd.ponder.models.ponder.masking import mask_blocks

    g = torch.Generator().manual_seed(1)
    grid = torch.randint(-20, 90, (3001, 3), generator=g)
    offset = torch.tensor([1200, 3000, 3001])
    feat = torch.randn(3001, 5, generator=g)
    token = torch.full((1, 5), -9.0)
    batch = torch.repeat_interleave(torch.arange(3), torch.tensor([1200, 1800, 1]))
    block = torch.cat([batch[:, None], torch.div(grid, 8).int()], 1)   # (the reference's expression)
    ublock, inv = block.unique(dim=0, return_inverse=True)
    sets = []
    for seed in (0, 1):
        torch.manual_seed(seed)
        out = mask_blocks(grid, feat, offset, 8, 0.75, token)
        masked = (out == -9.0).all(1)
        per_block = torch.zeros(len(ublock)).index_add_(0, inv, masked.float())
        size = torch.bincount(inv, minlength=len(ublock)).float()
        assert bool(((per_block == 0) | (per_block == size)).all())      # whole blocks
        for s in range(3):
            ids = ublock[:, 0] == s
            kept = int(((per_block == 0) & ids).sum())
            assert kept == round(int(ids.sum()) * (1 - 0.75)), (s, kept)
        assert torch.equal(out[~masked], feat[~masked])
        sets.append(masked)
    assert not torch.equal(sets[0], sets[1])


@pytest.mark.parametrize("hash_type", ["fnv", "ravel"])
def test_device_grid_sample_same_voxels_as_host_transform(hash_type):
    """grid_sample_torch (runs on any device) against the host GridSample: identical hash bit
    patterns, identical voxel set in identical (unsigned key) order, every representative a member
    of its voxel."""
    from ponderv2_amd.ponder.datasets import GridSample, fnv_hash_vec, ravel_hash_vec
    from ponderv2_amd.ponder.datasets.voxelize import (fnv_hash_torch, grid_sample_torch,
                                                       ravel_hash_torch)

    rng = np.random.default_rng(0)
    arr = rng.integers(0, 500, size=(4000, 3))
    host_hash = (fnv_hash_vec if hash_type == "fnv" else ravel_hash_vec)(arr)
    dev_hash = (fnv_hash_torch if hash_type == "fnv" else ravel_hash_torch)(torch.from_numpy(arr))
    assert np.array_equal(dev_hash.numpy().view(np.uint64), host_hash)

    pts = rng.uniform(-1.5, 2.0, size=(20000, 3)).astype(np.float32)
    np.random.seed(3)
    host = GridSample(grid_size=0.05, hash_type=hash_type, mode="train", keys=("coord",),
                      return_grid_coord=True)(dict(coord=pts.copy()))
    idx, grid = grid_sample_torch(torch.from_numpy(pts), 0.05, hash_type)
    assert np.array_equal(grid.numpy(), host["grid_coord"])          # same voxels, same order
    own = np.floor(pts[idx.numpy()] / 0.05).astype(int) - np.floor(pts / 0.05).astype(int).min(0)
    assert np.array_equal(own, grid.numpy())                         # representative lies in its voxel
    assert len(np.unique(idx.numpy())) == len(idx)
    # with pick == 0 the representative is the lowest point index of the voxel
    idx0, _ = grid_sample_torch(torch.from_numpy(pts), 0.05, hash_type,
                                pick=torch.zeros(len(idx), dtype=torch.int64))
    cell = np.floor(pts / 0.05).astype(int)
    _, first_member = np.unique(cell, axis=0, return_index=True)
    assert sorted(idx0.tolist()) == sorted(first_member.tolist())


def test_whole_model_autocast_runs_through_the_host_code(monkeypatch):
    """The reference's ScanNet config trains with enable_amp=True, i.e. the WHOLE model under
    autocast.  The kernels are fp32 launches autocast never touches; everything around them must
    hand them fp32 tensors.  Forward + backward of both models under bf16 autocast on the host."""
    import golden_cases as gc
    from oracle import cpu_backend
    from ponderv2_amd.ponder.datasets import (collate_fn, lidar_collate_fn, make_lidar_scene,
                                              make_scene)
    from ponderv2_amd.ponder.models import build_model
    from ponderv2_amd.ponder.utils.config import ConfigDict

    cpu_backend.install(monkeypatch)
    torch.manual_seed(0)
    small = dict(gc.SMALL_BACKBONE, channels=(16, 32, 48, 64, 64, 48, 32, 96))
    indoor = build_model(ConfigDict(gc.indoor_model_cfg(small, grid_shape=(32, 32, 8), ray_nsample=16))).train()
    kw = dict(n_raw=8000, num_views=2, image_hw=(24, 32))
    outdoor = build_model(ConfigDict(gc.outdoor_model_cfg(dict(small, in_channels=4),
                                                          **gc.OUTDOOR_SMALL))).train()
    cases = [(indoor, collate_fn([make_scene(100, **kw), make_scene(101, **kw)])),
             (outdoor, lidar_collate_fn([make_lidar_scene(200, **gc.OUTDOOR_SCENE_KW),
                                         make_lidar_scene(201, **gc.OUTDOOR_SCENE_KW)]))]
    for model, batch in cases:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out = model(batch)
        out["loss"].backward()
        assert torch.isfinite(out["loss"])
        grads = [p.grad for p in model.backbone.parameters() if p.grad is not None]
        assert grads and all(g.dtype == torch.float32 and torch.isfinite(g).all() for g in grads)


def test_loaders_apply_the_point_budget_of_the_config():
    """mix_prob / max_point reach every loader's collate as in the reference (engines/train.py:
    243-258, datasets/dataloader.py:67-80): the reference's pre-training configs set
    max_point = 2000000, which must build and must drop samples over the budget."""
    from ponderv2_amd.ponder.datasets import ConcatDataset, MultiDatasetDataloader
    from ponderv2_amd.ponder.datasets.collate import loader_collate, point_collate_fn

    class Scenes(torch.utils.data.Dataset):
        loop = 1

        def __len__(self):
            return 4

        def __getitem__(self, i):
            n = 10 * (i + 1)
            return dict(coord=torch.zeros(n, 3), offset=torch.tensor([n]))

    # reader-style dataset (no collate of its own): the reference's partial(point_collate_fn, ...)
    col = loader_collate(Scenes(), mix_prob=0, max_point=35)
    assert col.func is point_collate_fn and col.keywords == dict(mix_prob=0, max_point=35)
    batch = col([Scenes()[0], Scenes()[1], Scenes()[2]])       # 10 + 20 fit, 30 does not
    assert batch["offset"].tolist() == [10, 30] and batch["coord"].shape[0] == 30
    # dataset with its own batch assembly: same budget around it
    from ponderv2_amd.ponder.datasets import SyntheticRGBDDataset, make_scene

    ds = SyntheticRGBDDataset(length=2, num_views=1, image_hw=(12, 16), n_raw=2000)
    samples = [ds[0], ds[1]]
    n0 = len(samples[0]["coord"])
    assert loader_collate(ds, max_point=n0 + 1)(samples)["offset"].tolist() == [n0]
    # Mix3D merges the points of neighbouring scenes only: a pre-training batch carries per-scene
    # views / poses next to them, which would go out of step - refused with a clear message
    with pytest.raises(ValueError, match="mix_prob"):
        loader_collate(ds, mix_prob=1.0)(samples)

    class OwnCollate(Scenes):   # own batch assembly, plain point batches: mixing applies
        @staticmethod
        def collate_fn(batch):
            return point_collate_fn(batch)

    mixed = loader_collate(OwnCollate(), mix_prob=1.0)([Scenes()[0], Scenes()[1]])
    assert mixed["offset"].tolist() == [30]

    class Conditioned(Scenes):   # the reference's PPT fine-tuning batches: points + "condition"
        @staticmethod             # with mix_prob = 0.8 (semseg-ppt-v1m1-0-sc-s3-st-spunet-lovasz-ft.py)
        def collate_fn(batch):
            out = point_collate_fn([{k: v for k, v in b.items() if k != "condition"} for b in batch])
            out["condition"] = [b["condition"] for b in batch]
            return out

    tagged = [dict(Scenes()[i], condition="ScanNet") for i in range(2)]
    mixed = loader_collate(Conditioned(), mix_prob=1.0)(tagged)
    assert mixed["offset"].tolist() == [30] and mixed["condition"][0] == "ScanNet"
    # the multi-dataset loader no longer refuses the reference's settings
    loader = MultiDatasetDataloader(ConcatDataset([Scenes(), Scenes()], loop=1), 2, 0, mix_prob=0,
                                    seed=3, max_point=2000000)
    first = next(iter(loader))
    assert first["offset"].numel() == 2


def test_build_optimizer_takes_the_reference_param_dicts():
    """Absolute per-group lr / momentum / weight_decay keyed by name substring
    (ponder/utils/optimizer.py:21-56), group 0 at cfg.lr."""
    from ponderv2_amd.ponder.utils.config import ConfigDict
    from ponderv2_amd.ponder.utils.optimizer import build_optimizer

    model = torch.nn.Sequential()
    model.add_module("backbone", torch.nn.Linear(3, 3))
    model.add_module("modulation", torch.nn.Linear(3, 3))
    model.add_module("head", torch.nn.Linear(3, 3))
    cfg = ConfigDict(dict(type="SGD", lr=0.05, momentum=0.9, weight_decay=1e-4))
    opt = build_optimizer(cfg, model, [ConfigDict(dict(keyword="modulation", lr=0.005)),
                                       ConfigDict(dict(keyword="head", momentum=0.5, weight_decay=0.0))])
    g0, g1, g2 = opt.param_groups
    assert (g0["lr"], g0["momentum"], g0["weight_decay"]) == (0.05, 0.9, 1e-4) and len(g0["params"]) == 2
    assert (g1["lr"], g1["momentum"]) == (0.005, 0.9) and len(g1["params"]) == 2
    assert (g2["lr"], g2["momentum"], g2["weight_decay"]) == (0.05, 0.5, 0.0)
    assert len(build_optimizer(cfg, model, None).param_groups) == 1


def test_worker_seeds_follow_the_reference_formula(monkeypatch):
    from ponderv2_amd.ponder.engines import defaults

    seen = []
    monkeypatch.setattr(defaults, "set_seed", seen.append)
    defaults.worker_init_fn(3, num_workers=8, rank=2, seed=100)
    assert seen == [8 * 2 + 3 + 100]


def test_device_voxelisation_of_a_raw_batch_equals_host_gridsample():
    """datasets.voxelize.device_grid_sample on a collated batch of RAW points: per scene the same
    voxel set, in the same order, with the same integer coordinates as the host GridSample
    transform of the reference (datasets/transform.py:1078-1145); representatives are members of
    their voxel.  This is what Trainer.run_step applies when cfg.device_voxelize is set."""
    from ponderv2_amd.ponder.datasets import collate_fn, make_scene
    from ponderv2_amd.ponder.datasets.voxelize import device_grid_sample

    kw = dict(n_raw=6000, num_views=1, image_hw=(12, 16))
    host = [make_scene(s, **kw) for s in (5, 6)]
    raw = collate_fn([make_scene(s, voxelize=False, **kw) for s in (5, 6)])
    assert "grid_coord" not in raw and raw["coord"].shape[0] == 2 * 1200
    out = device_grid_sample(raw, grid_size=0.02, hash_type="fnv")
    ends = [0] + out["offset"].tolist()
    assert out["offset_host"] == ends[1:]
    for b, h in enumerate(host):
        g = out["grid_coord"][ends[b]:ends[b + 1]].numpy()
        assert np.array_equal(g, h["grid_coord"])                      # same voxels, same order
        c = out["coord"][ends[b]:ends[b + 1]].numpy()
        cell = np.floor(c.astype(np.float64) / 0.02).astype(int)
        assert np.array_equal(cell - cell.min(0), g)                  # representative lies in its voxel
        assert out["feat"].shape[0] == out["coord"].shape[0] == out["segment"].shape[0]
    assert out["rgb"] is raw["rgb"]                                     # everything else passes through


def test_trainer_lookahead_stages_the_next_batch_before_yielding_the_current():
    """Trainer.staged_batches: every loader batch comes out once, in order, already staged - and
    batch i+1 is staged BEFORE batch i is handed to the step (so that its device-side geometry
    overlaps step i)."""
    from types import SimpleNamespace

    from ponderv2_amd.ponder.engines.train import Trainer

    events = []

    def stage(b):
        events.append(("stage", b["i"]))
        return dict(b, _staged=True)

    fake = SimpleNamespace(stage=stage)
    seen = []
    for b in Trainer.staged_batches(fake, [dict(i=i) for i in range(4)]):
        events.append(("step", b["i"]))
        seen.append(b)
    assert [b["i"] for b in seen] == [0, 1, 2, 3] and all(b["_staged"] for b in seen)
    assert events == [("stage", 0), ("stage", 1), ("step", 0), ("stage", 2), ("step", 1), ("stage", 3),
                      ("step", 2), ("step", 3)]
    assert list(Trainer.staged_batches(fake, [])) == []


def test_small_scene_hint_and_sync_free_helpers():
    """The host-side scene extents answer "is any scene smaller than the dense grid" without a
    device read when no scene is near the threshold, and fall back to the exact test otherwise;
    offset2batch with a known row count and the unchecked inverse equal the plain forms."""
    from ponderv2_amd.ponder.models.ponder import ponder_indoor_base as pib
    from ponderv2_amd.ponder.models.utils import offset2batch

    model = type("M", (), {})()
    model.grid_size, model.grid_shape = 0.02, (128, 128, 32)
    fn = pib.PonderIndoor._small_scenes
    assert fn(model, dict(extent_host=[5.3, 4.1])) == []            # no "resolution" needed at all
    near = dict(extent_host=[5.3, 0.66], resolution=torch.tensor([264, 30]))
    assert fn(model, near) == [1]                                     # exact test on the doubtful scene
    assert fn(model, dict(resolution=torch.tensor([264, 30]))) == [1]  # no hint: exact test

    offset = torch.tensor([3, 3, 7, 12])
    assert torch.equal(offset2batch(offset), offset2batch(offset, 12))
    a = torch.randn(3, 4, 4, dtype=torch.float64) + 4 * torch.eye(4, dtype=torch.float64)
    assert torch.equal(pib._inv(a), torch.linalg.inv(a))


def test_folded_volume_materialises_for_every_other_consumer():
    """fused_head.FoldedVolume (the projection network's output with its final 1x1x1 convolution left
    to the fused head) is only created for device volumes; ``unfold`` / ``materialize`` give the
    128-channel volume the reference builds, and UNet3Dv1m2(fold_final=True) on the host returns it."""
    import torch.nn as nn

    from ponderv2_amd import fused_head as fhd
    from ponderv2_amd.ponder.models.ponder.unet3d import UNet3Dv1m2

    torch.manual_seed(0)
    net = UNet3Dv1m2(96, 128)
    x = torch.randn(1, 96, 8, 16, 16)
    ref = net(x)
    assert torch.equal(net(x, fold_final=True), ref)          # host tensors: nothing to fold
    conv = nn.Conv3d(fhd.KX, fhd.FS + fhd.F2, 1)
    pre = torch.randn(2, fhd.KX, 4, 6, 5)
    assert not fhd.fold_supported(conv, pre)                    # not on a device
    vol = fhd.FoldedVolume(pre, conv)
    assert vol.shape == (2, 128, 4, 6, 5) and vol.dim() == 5 and not vol.is_cuda
    out = fhd.unfold([vol, pre])
    assert torch.allclose(out[0], conv(pre)) and out[1] is pre
    x5, wfp = vol.rows()
    assert x5.shape == (2, 4, 6, 5, fhd.KX) and wfp.shape == (128, fhd.KXP)
    # [xt, s, 0...] . wfp^T is the convolution of a cell (s = 1: all corners inside)
    cell = torch.cat([x5[1, 2, 3, 4], torch.ones(1), torch.zeros(fhd.KXP - fhd.KX - 1)])
    assert torch.allclose(wfp @ cell, conv(pre)[1, :, 2, 3, 4], atol=1e-6)


def test_optimizer_builder_prefers_the_fused_step_only_on_the_device(monkeypatch):
    from ponderv2_amd.ponder.utils import optimizer as opt

    cfg = dict(type="SGD", lr=0.1, momentum=0.9, nesterov=True)
    host = [torch.nn.Parameter(torch.zeros(3))]
    assert "fused" not in opt._prefer_fused(cfg, host)                      # host parameters
    assert opt._prefer_fused(dict(cfg, foreach=True), host) == dict(cfg, foreach=True)
    fake = [type("P", (), {"is_cuda": True, "dtype": torch.float32, "is_floating_point": lambda s: True})()]
    monkeypatch.setattr(torch, "is_floating_point", lambda p: True)
    assert opt._prefer_fused(cfg, fake).get("fused") is True
    monkeypatch.setenv("PV2_FUSED_OPTIMIZER", "0")
    assert "fused" not in opt._prefer_fused(cfg, fake)


def test_split_backward_conv_equals_the_library_convolution():
    """unet3d._SplitBackwardConv issues grad-input and grad-weight as two ``convolution_backward``
    calls (the second goes to the backward side stream on the device); values and all three
    gradients equal ``nn.Conv3d`` / ``nn.ConvTranspose3d`` (with ``output_size``) exactly."""
    import torch.nn as nn

    from ponderv2_amd.ponder.models.ponder import unet3d as U

    torch.manual_seed(0)
    cases = [(nn.Conv3d(4, 6, 3, padding=1, bias=False), (2, 4, 5, 6, 7), None),
             (nn.Conv3d(4, 6, 1), (2, 4, 5, 6, 7), None),
             (nn.ConvTranspose3d(4, 3, 3, stride=2, padding=1), (2, 4, 3, 4, 5), [6, 7, 9])]
    for mod, shape, out_size in cases:
        x = torch.randn(*shape, requires_grad=True)
        y = mod(x) if out_size is None else mod(x, out_size)
        g = torch.randn_like(y)
        y.backward(g)
        ref = [x.grad.clone(), mod.weight.grad.clone(),
               None if mod.bias is None else mod.bias.grad.clone()]
        x.grad = None
        mod.zero_grad()
        transposed = isinstance(mod, nn.ConvTranspose3d)
        pad = (tuple(mod._output_padding(x, out_size, mod.stride, mod.padding, mod.kernel_size, 3,
                                         mod.dilation)) if transposed else (0, 0, 0))
        y2 = U._SplitBackwardConv.apply(x, mod.weight, mod.bias, tuple(mod.stride), tuple(mod.padding),
                                        tuple(mod.dilation), transposed, pad, mod.groups)
        assert torch.equal(y, y2)
        y2.backward(g)
        assert torch.equal(x.grad, ref[0]) and torch.equal(mod.weight.grad, ref[1])
        assert mod.bias is None or torch.equal(mod.bias.grad, ref[2])
    # on host tensors the wrapper is the module itself (no side stream to feed)
    conv = nn.Conv3d(32, 128, 1)
    v = torch.randn(1, 32, 2, 3, 4)
    assert torch.equal(U.library_conv(conv, v), conv(v)) and not U.pointwise_conv_supported(conv, v)


def test_side_stream_switches_and_leaf_rule():
    from ponderv2_amd import sidestream

    w = torch.nn.Parameter(torch.zeros(4, 3))
    assert sidestream.safe_leaf(w) and sidestream.safe_leaf(w.reshape(2, 6))   # a leaf / a view of one
    assert not sidestream.safe_leaf(w.t().contiguous() * 1.0)                    # a copy: not a leaf
    w.grad = torch.zeros_like(w)
    assert not sidestream.safe_leaf(w.reshape(2, 6))                             # autograd would ADD
    assert not sidestream.active(w)                                              # host tensor, no backward
    was = sidestream.ENABLED
    try:
        sidestream.disable("test")
        assert not sidestream.ENABLED and "test" in sidestream.status()
        sidestream.enable()
        assert sidestream.status() == ("on" if sidestream.ENABLED else "off (PV2_WGRAD_STREAM=0)")
    finally:
        sidestream.ENABLED = was


def test_border_class_table_equals_the_full_size_constant_part():
    """sparse_input._constant_part: the (3,3,3,C) class table expanded by a row gather equals the
    response of a zero-padded 3x3x3 conv to a constant field, computed the long way, with its
    gradients - including axes of size 1 and 2, where "first" and "last" coincide or touch."""
    import torch.nn.functional as F

    from ponderv2_amd.ponder.models.ponder import sparse_input as si

    torch.manual_seed(0)
    for dims in ((5, 4, 3), (1, 2, 6), (2, 1, 1)):
        weight = torch.randn(5, 4, 3, 3, 3, dtype=torch.float64, requires_grad=True)
        y0 = torch.randn(4, dtype=torch.float64, requires_grad=True)
        bias = torch.randn(5, dtype=torch.float64, requires_grad=True)
        got = si._constant_part(weight, y0, bias, 2, dims)
        field = y0[None, :, None, None, None].expand(2, 4, *dims)
        ref = F.conv3d(field, weight, bias, padding=1).permute(0, 2, 3, 4, 1).reshape(-1, 5)
        assert torch.allclose(got, ref, atol=1e-12)
        probe = torch.randn_like(ref)
        g_got = torch.autograd.grad((got * probe).sum(), (weight, y0, bias))
        g_ref = torch.autograd.grad((ref * probe).sum(), (weight, y0, bias))
        for a, b in zip(g_got, g_ref):
            assert torch.allclose(a, b, atol=1e-10)


def test_channels_last_max_pool_equals_max_pool3d():
    """unet3d.channels_last_max_pool3d (2-D NHWC pooling + maximum over z pairs; opt-in) against
    ``F.max_pool3d(x, 2)``: values exactly, gradients exactly away from ties; odd extents are
    floored the same way; the result stays channels-last."""
    import torch.nn.functional as F

    from ponderv2_amd.ponder.models.ponder.unet3d import channels_last_max_pool3d

    torch.manual_seed(0)
    for shape in ((2, 8, 4, 6, 10), (1, 3, 5, 7, 9)):
        x = torch.randn(*shape, dtype=torch.float64).contiguous(memory_format=torch.channels_last_3d)
        a = x.clone().requires_grad_(True)
        b = x.clone().requires_grad_(True)
        got, ref = channels_last_max_pool3d(a), F.max_pool3d(b, 2)
        assert torch.equal(got, ref) and got.is_contiguous(memory_format=torch.channels_last_3d)
        probe = torch.randn_like(ref)
        (got * probe).sum().backward()
        (ref * probe).sum().backward()
        assert torch.equal(a.grad, b.grad)


def test_fused_sgd_buffers_are_completed_again_after_a_partial_load():
    """A checkpoint without momentum buffers for some parameters (written by the for-each step, by
    the reference, or with per-condition norms that were never stepped) must not leave the fused
    SGD step with a mixed None / tensor buffer list: build_optimizer re-completes them after
    load_state_dict.  (Host tensors never get ``fused=True``; the hook is exercised directly.)"""
    from ponderv2_amd.ponder.utils import optimizer as O

    model = torch.nn.Sequential(torch.nn.Linear(3, 3), torch.nn.Linear(3, 2))
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
    opt.defaults["fused"] = True            # what _prefer_fused sets on a GPU model
    for g in opt.param_groups:
        g["fused"] = True
    O._ready_for_fused(opt)
    assert all("momentum_buffer" in opt.state[p] for p in model.parameters())
    partial = opt.state_dict()
    for idx in (2, 3):                      # the second layer was never stepped in that checkpoint
        partial["state"].pop(idx)
    opt.load_state_dict(partial)
    bufs = [opt.state[p].get("momentum_buffer") for p in model.parameters()]
    assert all(b is not None for b in bufs)


def test_pending_batchnorm_counts_do_not_survive_a_checkpoint_load():
    from ponderv2_amd import rownorm

    bn = torch.nn.BatchNorm1d(4)
    for _ in range(3):
        rownorm._bump_batches_tracked(bn)
    assert int(bn.num_batches_tracked) == 0 and bn._pv2_pending_batches == 3
    saved = bn.state_dict()                 # flushes: the checkpoint says 3
    assert int(saved["num_batches_tracked"]) == 3
    rownorm._bump_batches_tracked(bn)       # a warm-up step before the resume
    other = torch.nn.BatchNorm1d(4)
    other.num_batches_tracked.fill_(10)
    bn.load_state_dict(other.state_dict())
    assert bn._pv2_pending_batches == 0 and int(bn.state_dict()["num_batches_tracked"]) == 10
    rownorm._bump_batches_tracked(bn)
    model = torch.nn.Sequential(bn)
    rownorm.flush_bn_counters(model)        # for direct readers of the buffers (EMA copy, broadcast)
    assert int(bn.num_batches_tracked) == 11 and bn._pv2_pending_batches == 0


def test_fused_bn_affine_overrides_fall_back_to_the_same_normalisation():
    """rownorm.fused_bn with an affine override where the kernels do not apply (host tensor, one
    row, eval mode): F.batch_norm with the caller's pair - not an assertion."""
    from ponderv2_amd.rownorm import fused_bn

    torch.manual_seed(0)
    bn = torch.nn.BatchNorm1d(5)
    x, w, b = torch.randn(9, 5), torch.randn(5), torch.randn(5)
    res = torch.randn(9, 5)
    got = fused_bn(bn, x, residual=res, relu=True, weight=w, bias=b)
    xh = (x - x.mean(0)) / torch.sqrt(x.var(0, unbiased=False) + bn.eps)
    assert torch.allclose(got, torch.relu(xh * w + b + res), atol=1e-6)
    assert int(bn.num_batches_tracked) == 1
    assert torch.allclose(bn.running_mean, 0.1 * x.mean(0), atol=1e-6)
    bn.eval()
    got = fused_bn(bn, x, weight=w, bias=b)
    want = (x - bn.running_mean) / torch.sqrt(bn.running_var + bn.eps) * w + b
    assert torch.allclose(got, want, atol=1e-6)


def test_arena_views_split_matches_per_piece_views():
    """spunet_native._Arena.views: the gradient views of the native backward from ONE split call equal
    the piece-by-piece views (offsets are 64-float aligned, gaps are skipped, the tail is ignored)."""
    from ponderv2_amd.spunet_native import _Arena

    a = _Arena()
    offs = [a.reserve(n) for n in (10, 100, 7, 64, 1)]
    a.allocate("cpu")
    a.tensor.copy_(torch.arange(a.size, dtype=torch.float32))
    specs = [(offs[0], 4), (offs[0] + 4, 6), (offs[1], 100), (offs[3], 64), (offs[4], 1)]
    got = a.views(specs)
    assert len(got) == len(specs)
    for (off, n), v in zip(specs, got):
        assert torch.equal(v, a.view(off, n)) and v.data_ptr() == a.view(off, n).data_ptr()


def test_cells_geometry_ahead_of_time_equals_inline():
    """``cells_from_voxels`` with the geometry computed beforehand (what PonderIndoor.prefetch does a step
    ahead) returns the same cells, rows and rulebook as computing everything in place."""
    import torch

    from ponderv2_amd.ponder.models.ponder import sparse_input as si
    from oracle import cpu_backend

    g = torch.Generator().manual_seed(3)
    B, dims = 2, (3, 5, 4)
    total = B * dims[0] * dims[1] * dims[2]
    lin = torch.randint(0, total, (40,), generator=g)
    feat = torch.randn(40, 8, generator=g)
    with cpu_backend.installed():
        a = si.cells_from_voxels(feat, lin, B, dims)
        geo = si.cells_geometry(lin, B, dims, build_rulebook=True)
        b = si.cells_from_voxels(feat, None, B, dims, geometry=geo)
        assert torch.equal(a.lin, b.lin) and torch.equal(a.feat, b.feat) and a.dims == b.dims
        ra, rb = a.rulebook(), b.rulebook()
    assert rb is geo["rulebook"]
    assert torch.equal(ra.pair_in[:ra.n_pairs], rb.pair_in[:rb.n_pairs])
    assert torch.equal(ra.pair_out[:ra.n_pairs], rb.pair_out[:rb.n_pairs])


def test_spconv_shaped_state_dict_loads():
    """A checkpoint as the reference would save it with spconv 2.x loads into the product model, strictly.

    No spconv wheel and no released checkpoint exist offline, so the two things a real checkpoint would
    pin - the weight layout ``[C_out, kD, kH, kW, C_in]`` of SubMConv3d / SparseConv3d / SparseInverseConv3d
    (spconv 2.x, KRSC; ponder/models/sparse_unet/spconv_unet_v1m1_base.py:41-66,111-181 builds them with
    bias=False) and their default initialisation - stay UNPINNED against spconv itself (DESIGN.md section 4,
    INTEGRATION.md).  What this test fixes is the contract the product states: a ``state_dict`` built
    INDEPENDENTLY of the product's parameters - key names from the reference's module tree, shapes from the
    documented layout and the config's channel counts - loads with ``strict=True``, and a conv reads it as
    ``[C_out, K, C_in]`` without a transpose."""
    import golden_cases as gc
    from ponderv2_amd.ponder.models import build_model
    from ponderv2_amd.ponder.utils.config import ConfigDict
    from ponderv2_amd.spconv import pytorch as spconv

    cfg = dict(gc.FULL_BACKBONE)
    model = build_model(ConfigDict(cfg))
    gen = torch.Generator().manual_seed(0)
    synthetic, convs = {}, 0
    for name, mod in model.named_modules():
        if isinstance(mod, (spconv.SubMConv3d, spconv.SparseConv3d, spconv.SparseInverseConv3d)):
            ks = mod.kernel_size if isinstance(mod.kernel_size, (tuple, list)) else (mod.kernel_size,) * 3
            shape = (mod.out_channels, *ks, mod.in_channels)        # spconv 2.x KRSC
            synthetic[name + ".weight"] = torch.randn(shape, generator=gen)
            assert mod.bias is None
            convs += 1
        elif isinstance(mod, torch.nn.BatchNorm1d):
            c = mod.num_features
            synthetic.update({name + ".weight": torch.rand(c, generator=gen), name + ".bias": torch.randn(c, generator=gen),
                              name + ".running_mean": torch.randn(c, generator=gen),
                              name + ".running_var": torch.rand(c, generator=gen) + 0.5,
                              name + ".num_batches_tracked": torch.tensor(7)})
    assert convs == 59          # 59 sparse-conv layers
    assert sum(v.numel() for k, v in synthetic.items() if k.endswith(".weight") and v.dim() == 5) > 38e6
    missing, unexpected = model.load_state_dict(synthetic, strict=True)
    assert not missing and not unexpected
    # the first residual block's conv as the kernels see it: [C_out, K, C_in], contiguous, no copy
    conv = model.enc[0][0].conv1
    w = conv.weight.reshape(conv.out_channels, -1, conv.in_channels)
    assert w.data_ptr() == conv.weight.data_ptr() and w.shape[1] == 27
    assert torch.equal(w[:, 13, :], synthetic["enc.0.block0.conv1.weight"][:, 1, 1, 1, :])


def test_record_stream_reaches_every_tensor_of_every_rulebook():
    """``kernels._record_stream`` (what makes the prefetched geometry safe to use on the training stream)
    must visit EVERY tensor of EVERY rulebook in the result.  Round 5 found that it walked a rulebook's
    fields through a temporary list whose ``id()`` went into the seen-set: the next rulebook's temporary could
    be handed the same address and the whole rulebook was skipped - its tensors stayed unrecorded and the
    caching allocator re-used them under the training stream's kernels (a memory fault of bench.py)."""
    from ponderv2_amd import kernels as K

    recorded = []

    class FakeDeviceTensor(torch.Tensor):
        is_cuda = property(lambda self: True)

        def record_stream(self, stream):
            recorded.append(id(self))

    def t():
        return torch.zeros(3).as_subclass(FakeDeviceTensor)

    everything = []

    def plan():
        p = K.OsmPlanData.__new__(K.OsmPlanData)
        p.perm, p.tblp, p.tmask, p.n_pad, p.kflip, p.struct = t(), t(), t(), 256, 0, None
        everything.extend([p.perm, p.tblp, p.tmask])
        return p

    result = {}
    for i in range(12):
        pin, pout, ks = t(), t(), t()
        rb = K.Rulebook(8, 1, 1, pin, pout, ks, np.zeros(9, np.int64), _tiles_dev=t())
        rb.nbr = t()
        rb._pos = (t(), 1, t(), 1)
        rb.osm, rb.osm_t = plan(), plan()
        everything.extend([pin, pout, ks, rb.nbr, rb._tiles_dev, rb._pos[0], rb._pos[2]])
        result[f"spconv{i}"] = dict(kind="down", rulebook=rb, in_indices=t(), out_indices=t())
        everything.extend([result[f"spconv{i}"]["in_indices"], result[f"spconv{i}"]["out_indices"]])
    K._record_stream(result, stream=object())
    assert sorted(recorded) == sorted(id(x) for x in everything)


def test_gradient_slabs_partition_the_executor_arena_in_completion_order():
    """``spunet_native._gradient_slabs`` (round 5: what the overlapped gradient reduction is told): the
    parameter-gradient arena is laid out first unit to last, the backward finishes units last to first - so
    the slabs must tile [first conv unit's offset, arena end) from the END, each at least ``slab_elems`` long
    (the last one takes what is left), each named by the LOWEST unit it contains, the stem outside."""
    from ponderv2_amd import spunet_native as sn
    from ponderv2_amd._lib import UNET_CONCAT, UNET_CONV_BN, UNET_STEM

    class Hook:
        slab_elems = 5000

        def __init__(self):
            self.asked = None

        def wants(self, tensors):
            self.asked = list(tensors)
            return True

    plan, tensors, arena = sn.Plan(), [], sn._Arena()
    kinds = [UNET_STEM] + [UNET_CONV_BN, UNET_CONV_BN, UNET_CONCAT] * 4 + [UNET_CONV_BN]
    for i, kind in enumerate(kinds):
        u = sn._Unit()
        u.kind = kind
        if kind != UNET_CONCAT:
            c_out, k, c_in = 8 + 4 * (i % 3), 27 if kind != UNET_STEM else 125, 8
            w = torch.nn.Parameter(torch.zeros(c_out, k, c_in))
            bw, bb = torch.nn.Parameter(torch.zeros(c_out)), torch.nn.Parameter(torch.zeros(c_out))
            u.c_out, u.w_index = c_out, len(tensors)
            tensors += [w, bw, bb]
            u.gsum_off = arena.reserve(2 * c_out)
            u.dw_off = arena.reserve(w.numel())
            u.affine_in_arena = True      # (leaf BatchNorm pairs: their gradients are arena members)
        plan.units.append(u)
    hook = Hook()
    members, spans, units = sn._gradient_slabs(plan, tensors, hook, arena)
    convs = [(i, u) for i, u in enumerate(plan.units) if u.kind == UNET_CONV_BN]
    assert len(hook.asked) == 3 * len(convs)                      # the stem's tensors are not offered
    assert spans[0][1] == arena.size and spans[-1][0] == convs[0][1].gsum_off
    assert all(spans[j + 1][1] == spans[j][0] for j in range(len(spans) - 1))     # contiguous, descending
    assert all(hi - lo >= hook.slab_elems for lo, hi in spans[:-1]) and len(spans) >= 3
    assert units == sorted(units, reverse=True) and units[-1] == convs[0][0]
    for (lo, hi), i in zip(spans, units):
        assert plan.units[i].gsum_off == lo                       # named by its lowest unit
    assert len(members) == 3 * len(convs)
    for t, off, n in members:
        assert t.numel() == n and spans[-1][0] <= off and off + n <= arena.size
        assert sum(lo <= off and off + n <= hi for lo, hi in spans) == 1          # inside exactly one slab
    # round 6: a COMPUTED BatchNorm pair (SpUNet-v1m3's modulated affine: not a leaf) keeps its gradient in a
    # second arena - such a unit contributes its conv weight only, and its slab starts at the weight
    pd = plan.units[4]
    assert pd.kind == UNET_CONV_BN
    pd.affine_in_arena = False
    members2, spans2, units2 = sn._gradient_slabs(plan, tensors, hook, arena)
    assert len(members2) == 3 * len(convs) - 2
    assert all(off != pd.gsum_off and off != pd.gsum_off + pd.c_out for _, off, _ in members2)
    for t, off, n in members2:
        assert sum(lo <= off and off + n <= hi for lo, hi in spans2) == 1
    pd.affine_in_arena = True
    # too few conv units, or a reducer that declines: no slabs
    assert sn._gradient_slabs(plan, tensors, type("No", (), {"slab_elems": 1, "wants": lambda s, t: False})(),
                              arena) is None


def test_bench_reads_a_pmc_file_measured_on_these_kernel_sources():
    """``roofline.traffic`` of the bench line is a read of the round's committed PMC file; bench.py refuses a
    file measured on other kernel sources (traffic null).  This keeps the committed file and the sparse-conv
    sources together: a kernel edit without fresh counter passes - or a file that never reached the repo - fails
    here instead of silently dropping the figure from the driver's line."""
    import bench

    traffic, source = bench.pmc_traffic("spconv_fwd_lds_kernel<4")
    assert traffic is not None, source
    assert bench.kernel_source_hash() in source
    assert 1e7 < traffic < 1e9, traffic      # bytes per launch of the 128-channel products kernel (~1e8)
