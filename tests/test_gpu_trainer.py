"""The hook-driven Trainer on the device: raw points -> device voxelisation on the input stream ->
PonderIndoor step (dense node, render head) -> GradScaler/optimizer/scheduler -> hooks.

The CPU suite drives the same loop over two gloo ranks (tests/test_ddp_gloo.py); this is its
single-GPU twin, the path ``tools/train.py`` takes on an MI355X (reference: pointcept/engines/
train.py:133-230 ``Trainer.train``/``run_step``)."""
import json
import os

import pytest
import torch

import ddp_worker

pytestmark = pytest.mark.gpu


def _cfg(tmp_path, **over):
    from ponderv2_amd.ponder.utils.config import Config

    base = dict(
        weight=None, resume=False, evaluate=False, seed=3, save_path=str(tmp_path), num_worker=0,
        batch_size=2, epoch=1, eval_epoch=1, sync_bn=False, enable_amp=False, empty_cache=False,
        find_unused_parameters=True, mix_prob=0, max_point=2000000, param_dicts=None,
        device_voxelize=dict(grid_size=0.02, hash_type="fnv"),
        hooks=[dict(type="CheckpointLoader"), dict(type="IterationTimer", warmup_iter=0),
               dict(type="InformationWriter"), dict(type="CheckpointSaver", save_freq=None)],
        train=dict(type="DefaultTrainer"), model=ddp_worker.tiny_model_cfg(),
        optimizer=dict(type="SGD", lr=1e-3, momentum=0.9, weight_decay=1e-4, nesterov=True),
        scheduler=dict(type="OneCycleLR", max_lr=1e-3, pct_start=0.05, anneal_strategy="cos",
                       div_factor=10.0, final_div_factor=10000.0),
        data=dict(train=dict(type="SyntheticRGBDDataset", length=6, base_seed=80, num_views=2,
                             image_hw=(24, 32), n_raw=5000, voxelize=False)))
    base.update(over)
    return Config(base)


def _train(cfg, tmp_path):
    from ponderv2_amd.ponder.engines import default_setup
    from ponderv2_amd.ponder.engines.train import TRAINERS

    os.makedirs(tmp_path / "model", exist_ok=True)
    cfg = default_setup(cfg)
    trainer = TRAINERS.build(dict(type=cfg.train.type, cfg=cfg))
    before = {n: p.detach().clone() for n, p in trainer.model.named_parameters()}
    trainer.train()
    torch.cuda.synchronize()
    return trainer, before


@pytest.mark.timeout(600)
@pytest.mark.parametrize("amp", [None, "bfloat16", "float16"])
def test_trainer_steps_on_device(tmp_path, amp):
    over = {} if amp is None else dict(enable_amp=True, amp_dtype=amp)
    trainer, before = _train(_cfg(tmp_path, **over), tmp_path)
    assert next(trainer.model.parameters()).is_cuda
    rows = [json.loads(l) for l in open(tmp_path / "scalars.jsonl")]
    assert len(rows) == 3, rows
    assert all(r["loss"] == r["loss"] and abs(r["loss"]) < 1e6 for r in rows), rows
    moved = [n for n, p in trainer.model.named_parameters()
             if not torch.equal(p.detach(), before[n])]
    # every family of the path saw a gradient and an optimizer step
    for family in ("backbone.conv_input", "backbone.enc", "backbone.dec", "backbone.up",
                   "proj_net.encoders", "proj_net.decoders", "renderer.field"):
        assert any(family in n for n in moved), (family, moved[:8])
    assert all(torch.isfinite(p).all() for p in trainer.model.parameters())
    ckpt = torch.load(tmp_path / "model" / "model_last.pth", weights_only=False)
    assert ckpt["epoch"] == 1 and "backbone.conv_input.0.weight" in ckpt["state_dict"]


@pytest.mark.timeout(600)
def test_trainer_repeatable_on_device(tmp_path):
    """Two runs from the same seed end at the same parameters up to the float atomics left on the path
    (scatter-mean, the first projection level's accumulation, the sampler's grid gradient): ~1e-7
    relative per op (tools/check_cells_node.py), amplified over three steps wherever it flips a ReLU
    or a max-pool winner.  Garbage (uninitialised memory, a race) would show orders above this bound."""
    a, _ = _train(_cfg(tmp_path / "a"), tmp_path / "a")
    pa = {n: p.detach().clone() for n, p in a.model.named_parameters()}
    b, _ = _train(_cfg(tmp_path / "b"), tmp_path / "b")
    diff = [n for n, p in b.model.named_parameters() if not torch.equal(p.detach(), pa[n])]
    for n, p in b.model.named_parameters():
        err = (p.detach() - pa[n]).abs().max().item()
        assert err <= 1e-3 * pa[n].abs().max().item() + 1e-6, (n, err)
    print(f"{len(diff)} parameter tensors differ in the last bits between two seeded runs")


def test_prefetched_ray_setup_equals_inline(device):
    """``PonderIndoor.prefetch`` does the ray set-up with the batch (one step ahead, on the input stream);
    the step must see the same rays and targets as when ``prepare_ray`` runs inside it."""
    from ponderv2_amd.ponder.datasets import collate_fn
    from ponderv2_amd.ponder.datasets.voxelize import input_stream
    from ponderv2_amd.ponder.models import build_model
    from ponderv2_amd.ponder.utils.config import ConfigDict

    torch.manual_seed(0)
    model = build_model(ConfigDict(ddp_worker.tiny_model_cfg())).to(device).train()
    batch = collate_fn([ddp_worker.tiny_scene(60), ddp_worker.tiny_scene(61)])
    batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
    n = model.ray_nsample
    B, V, H, W = batch["depth"].shape
    g = torch.Generator().manual_seed(5)
    batch["ray_pixels"] = torch.stack([torch.randint(0, H, (B, V, n), generator=g),
                                       torch.randint(0, W, (B, V, n), generator=g)], -1).to(device)

    def clone():
        return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}

    torch.manual_seed(1)
    inline = model(clone())
    with input_stream(device) as pipe:
        staged = pipe.adopt(model.prefetch(clone()))
    assert "_ray_dict" in staged and "_cells_geometry" in staged
    torch.manual_seed(1)
    ahead = model(staged)
    for k in inline:
        torch.testing.assert_close(ahead[k], inline[k], rtol=1e-5, atol=1e-6, msg=lambda m, k=k: f"{k}: {m}")
