"""csrc/ray_setup.hip (PonderIndoor.prepare_ray in four launches) against the torch statement of the same
arithmetic in ponder_indoor_base.py - which the end-to-end goldens pin to the reference's
to_unit_cube / ray_sample (ponder_indoor_base.py:344-497 there)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu


def _clone(d):
    return {k: (v.clone() if torch.is_tensor(v) else (list(v) if isinstance(v, list) else v)) for k, v in d.items()}


@pytest.mark.parametrize("semantic", [True, False])
def test_fused_ray_setup_equals_the_torch_route(device, semantic):
    import bench
    from ponderv2_amd import ray_setup
    from ponderv2_amd.ponder.models import build_model
    from ponderv2_amd.ponder.utils.config import ConfigDict

    torch.manual_seed(0)
    cfg = bench.model_cfg(64, "float32")
    model = build_model(ConfigDict(cfg)).to(device).train()
    model.render_semantic = semantic
    model.bounds = [[-0.3, -0.3, -0.3], [0.3, 0.3, 0.3]]   # a tight box: part of the rays miss it
    batch = bench.make_batch(0, 2, 2, device)
    B, V, H, W = batch["depth"].shape
    n = model.ray_nsample
    g = torch.Generator().manual_seed(1)
    pix = torch.stack([torch.randint(0, H, (B, V, n), generator=g), torch.randint(0, W, (B, V, n), generator=g)], -1)
    batch["depth"][0, 0, :40] = 0.0                       # pixels without a depth reading
    pix[0, 0, :8, 0] = torch.arange(8)                    # ... some of them chosen
    batch["ray_pixels"] = pix.to(device)
    res = {}
    for fused in (True, False):
        ray_setup.ENABLED = fused
        before = ray_setup.CALLS
        try:
            ray, d = model.prepare_ray(_clone(batch))
        finally:
            ray_setup.ENABLED = True
        assert (ray_setup.CALLS > before) == fused
        res[fused] = (ray, {k: d[k] for k in ("extrinsic", "depth_scale", "pc_scale", "bbox", "coord")})
    (ra, da), (rb, db) = res[True], res[False]
    assert ra.keys() == rb.keys() and ("semantic" in ra) == semantic
    for k in da:
        assert da[k].shape == db[k].shape, k
        err = float((da[k].double() - db[k].double()).abs().max() / (db[k].abs().max() + 1e-12))
        assert err < 2e-6, (k, err)
    miss_a, miss_b = ra["depth"] < 0, rb["depth"] < 0
    assert torch.equal(miss_a, miss_b) and 0 < int(miss_a.sum()) < miss_a.numel()
    for k in ra:
        assert ra[k].shape == rb[k].shape and ra[k].dtype == rb[k].dtype, k
        err = float((ra[k].double() - rb[k].double()).abs().max() / (rb[k].abs().max() + 1e-12))
        assert err < 5e-6, (k, err)
    if semantic:
        assert torch.equal(ra["semantic"], rb["semantic"])
    assert torch.equal(ra["rgb"], rb["rgb"])


def test_fused_ray_setup_chooses_valid_pixels(device):
    """Without caller-given pixels: n distinct pixels with a depth reading per view, rays of unit length."""
    import bench
    from ponderv2_amd import ray_setup
    from ponderv2_amd.ponder.models import build_model
    from ponderv2_amd.ponder.utils.config import ConfigDict

    model = build_model(ConfigDict(bench.model_cfg(64, "float32"))).to(device).train()
    batch = bench.make_batch(0, 2, 2, device)
    batch["depth"][:, :, ::2] = 0.0
    before = ray_setup.CALLS
    ray, d = model.prepare_ray(_clone(batch))
    assert ray_setup.CALLS == before + 1
    B, V = batch["depth"].shape[:2]
    n = model.ray_nsample
    assert ray["ray_d"].shape == (B, V * n, 3)
    assert float((ray["ray_d"].norm(dim=-1) - 1).abs().max()) < 1e-5
    hit = ray["depth"] > 0
    assert float(hit.float().mean()) > 0.5
