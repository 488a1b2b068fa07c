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


def test_offset2batch():
    from ponderv2_amd.ponder.models.utils import batch2offset, offset2batch

    off = torch.tensor([3, 3, 7])
    b = offset2batch(off)
    assert b.tolist() == [0, 0, 0, 2, 2, 2, 2]
    assert batch2offset(torch.tensor([0, 0, 1, 1, 1])).tolist() == [2, 5]
