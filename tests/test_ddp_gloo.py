"""N>1 path on CPU: world_size 2 over gloo (the GPU path uses the same code with RCCL)."""
import json
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

import ddp_worker


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(600)
def test_ddp_gradients_are_rank_means(tmp_path):
    world = 2
    mp.spawn(ddp_worker.grad_sync_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world,
             join=True)
    res = [torch.load(tmp_path / f"rank{r}.pt") for r in range(world)]
    assert all(r["worst"] < 1e-5 for r in res), res
    assert res[0]["loss"] != res[1]["loss"]  # different scenes per rank (sharded, not replicated)
    assert res[0]["n_grads"] == res[1]["n_grads"] > 100
    # the DDP-free flat reduction: same averages, unused parameters keep grad None, a parameter
    # only one rank trains gets that rank's gradient / world on both
    assert all(r["worst_flat"] < 1e-5 for r in res), res
    assert all(r["unused_stay_none"] and r["skip_ok"] for r in res), res
    assert all(r["inplace_ok"] for r in res), res   # gradients kept in place between steps (set_to_none=False)
    assert res[0]["n_params"] > res[0]["n_grads"]   # (the model does have unused parameters)


@pytest.mark.timeout(900)
def test_flat_grad_sync_moves_arena_gradients_as_blocks(tmp_path):
    """Gradients handed back as views of one arena (the native sparse executor's parameter-gradient
    arena) travel through FlatGradSync in one copy each way; the result is the rank mean in every
    mix of arena / ordinary gradients across steps and ranks."""
    world = 2
    mp.spawn(ddp_worker.arena_sync_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world,
             join=True)
    for r in range(world):
        res = torch.load(tmp_path / f"arena{r}.pt")
        assert res["ok"], res
        assert res["blocks"] == 1 and res["block_members"] == 60, res


@pytest.mark.timeout(900)
def test_overlapped_slab_reduction_equals_the_flat_reduction_bit_for_bit(tmp_path):
    """FlatGradSync(overlap=True) - slabs of the executor's gradient arena all-reduced in place from the
    backward hook - against the flat form on the same gradients: identical bits on both ranks in every
    mix of hook / staged steps; and the layout-agreement reduce is issued by a rank without an arena."""
    world = 2
    mp.spawn(ddp_worker.overlap_sync_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world,
             join=True)
    for r in range(world):
        res = torch.load(tmp_path / f"overlap{r}.pt")
        assert res["ok"], res
        assert res["slabs"] >= 3 and res["covered"] == 58 and res["blocks_third"] == 0, res


def test_trainer_two_ranks_via_launch(tmp_path):
    """engines.launch spawns one process per 'GPU', DistributedSampler shards the scenes, hooks
    log and checkpoint on rank 0."""
    from ponderv2_amd.ponder.engines import launch
    from ponderv2_amd.ponder.utils.config import Config

    cfg = Config(dict(
        weight=None, resume=False, evaluate=False, seed=3, save_path=str(tmp_path), num_worker=0,
        batch_size=2, epoch=1, eval_epoch=1, sync_bn=False, enable_amp=False, empty_cache=False,
        find_unused_parameters=True, mix_prob=0, max_point=2000000, param_dicts=None,
        # the loader streams RAW points, the trainer voxelises them after the host->device copy
        device_voxelize=dict(grid_size=0.02, hash_type="fnv"),
        hooks=[dict(type="CheckpointLoader"), dict(type="IterationTimer", warmup_iter=0),
               dict(type="InformationWriter"), dict(type="CheckpointSaver", save_freq=None)],
        train=dict(type="DefaultTrainer"), model=ddp_worker.tiny_model_cfg(),
        optimizer=dict(type="SGD", lr=1e-4, momentum=0.9, weight_decay=1e-4, nesterov=True),
        scheduler=dict(type="OneCycleLR", max_lr=1e-4, pct_start=0.05, anneal_strategy="cos",
                       div_factor=10.0, final_div_factor=10000.0),
        data=dict(train=dict(type="SyntheticRGBDDataset", length=4, base_seed=80, num_views=2,
                             image_hw=(24, 32), n_raw=5000, voxelize=False))))
    os.makedirs(tmp_path / "model", exist_ok=True)
    launch(ddp_worker.trainer_main, num_gpus_per_machine=2, cfg=(cfg,))
    rows = [json.loads(l) for l in open(tmp_path / "scalars.jsonl")]
    assert len(rows) == 2 and all("loss" in r and r["loss"] == r["loss"] for r in rows)
    ckpt = torch.load(tmp_path / "model" / "model_last.pth", weights_only=False)
    assert ckpt["epoch"] == 1 and "backbone.conv_input.0.weight" in ckpt["state_dict"]


@pytest.mark.timeout(900)
def test_multi_dataset_trainer_outdoor_two_ranks(tmp_path):
    """MultiDatasetTrainer + PonderOutdoor-v2 (block masking, lidar rays) over 2 gloo ranks."""
    import golden_cases as gc
    from ponderv2_amd.ponder.engines import launch
    from ponderv2_amd.ponder.utils.config import Config

    model = gc.outdoor_model_cfg(dict(gc.SMALL_BACKBONE, in_channels=4,
                                      channels=(16, 32, 48, 64, 64, 48, 32, 96)), **gc.OUTDOOR_SMALL)
    cfg = Config(dict(
        weight=None, resume=False, evaluate=False, seed=5, save_path=str(tmp_path), num_worker=0,
        batch_size=2, epoch=1, eval_epoch=1, sync_bn=False, enable_amp=False, empty_cache=False,
        find_unused_parameters=True, mix_prob=0, param_dicts=None,
        # gradients averaged by ONE flat all-reduce after backward (utils/grad_sync.py) instead of
        # the DDP wrapper; with find_unused_parameters it also exchanges the per-parameter usage
        grad_sync="flat",
        hooks=[dict(type="CheckpointLoader"), dict(type="IterationTimer", warmup_iter=0),
               dict(type="InformationWriter"), dict(type="CheckpointSaver", save_freq=None)],
        train=dict(type="MultiDatasetTrainer"), model=model,
        optimizer=dict(type="AdamW", lr=2e-4, weight_decay=0.01),
        scheduler=dict(type="OneCycleLR", max_lr=2e-4, pct_start=0.4, anneal_strategy="cos",
                       div_factor=10.0, final_div_factor=100.0),
        data=dict(train=dict(type="ConcatDataset", datasets=[
            dict(type="SyntheticLidarDataset", length=4, base_seed=300, loop=1,
                 **gc.OUTDOOR_SCENE_KW)]))))
    os.makedirs(tmp_path / "model", exist_ok=True)
    launch(ddp_worker.trainer_main, num_gpus_per_machine=2, cfg=(cfg,))
    rows = [json.loads(l) for l in open(tmp_path / "scalars.jsonl")]
    assert len(rows) == 2 and all("depth_loss" in r and r["loss"] == r["loss"] for r in rows)
    ckpt = torch.load(tmp_path / "model" / "model_last.pth", weights_only=False)
    assert "mtoken" in ckpt["state_dict"]


@pytest.mark.timeout(900)
def test_multi_dataset_trainer_ppt_two_ranks(tmp_path):
    """BASELINE config 4 in miniature: PonderIndoor-v2 over SpUNet-v1m3 (PDNorm), two synthetic
    "datasets" with sampling ratio 2:1; every batch carries one condition, both ranks walk the
    same schedule (the sets of unused parameters differ per condition)."""
    import golden_cases as gc
    from ponderv2_amd.ponder.engines import launch
    from ponderv2_amd.ponder.utils.config import Config

    model = gc.indoor_model_cfg(dict(gc.PDNORM_BACKBONE, context_channels=256,
                                     channels=(16, 32, 48, 64, 64, 48, 32, 96),
                                     conditions=("ScanNet", "S3DIS", "Structured3D")),
                                grid_shape=(32, 32, 8), ray_nsample=6)
    model.update(conditions=("Structured3D", "ScanNet"),
                 valid_index=(tuple(range(0, 13)), tuple(range(5, 20))))
    scene = dict(type="SyntheticRGBDDataset", num_views=2, image_hw=(24, 32), n_raw=5000)
    cfg = Config(dict(
        weight=None, resume=False, evaluate=False, seed=11, save_path=str(tmp_path), num_worker=0,
        batch_size=2, epoch=1, eval_epoch=1, sync_bn=False, enable_amp=False, empty_cache=False,
        find_unused_parameters=True, mix_prob=0, param_dicts=None,
        hooks=[dict(type="CheckpointLoader"), dict(type="IterationTimer", warmup_iter=0),
               dict(type="InformationWriter"), dict(type="CheckpointSaver", save_freq=None)],
        train=dict(type="MultiDatasetTrainer"), model=model,
        optimizer=dict(type="SGD", lr=1e-4, momentum=0.9, weight_decay=1e-4, nesterov=True),
        scheduler=dict(type="OneCycleLR", max_lr=1e-4, pct_start=0.05, anneal_strategy="cos",
                       div_factor=10.0, final_div_factor=10000.0),
        data=dict(train=dict(type="ConcatDataset", loop=1, datasets=[
            dict(scene, length=4, base_seed=500, condition="Structured3D", num_classes=13, loop=2),
            dict(scene, length=2, base_seed=600, condition="ScanNet", num_classes=15, loop=1)]))))
    os.makedirs(tmp_path / "model", exist_ok=True)
    launch(ddp_worker.trainer_main, num_gpus_per_machine=2, cfg=(cfg,))
    rows = [json.loads(l) for l in open(tmp_path / "scalars.jsonl")]
    assert len(rows) == 3 and all(r["loss"] == r["loss"] for r in rows)  # schedule: S3D S3D ScanNet
    ckpt = torch.load(tmp_path / "model" / "model_last.pth", weights_only=False)
    sd = ckpt["state_dict"]
    assert "backbone.conv_input.bn.bns.2.running_mean" in sd
    # per-condition statistics: Structured3D (index 2) and ScanNet (0) were updated, S3DIS (1) never
    assert sd["backbone.conv_input.bn.bns.1.num_batches_tracked"] == 0
    assert sd["backbone.conv_input.bn.bns.2.num_batches_tracked"] == 2
    assert sd["backbone.conv_input.bn.bns.0.num_batches_tracked"] == 1


@pytest.mark.timeout(600)
def test_resume_continues_from_checkpoint(tmp_path):
    """CheckpointSaver -> CheckpointLoader(resume=True): the second run starts at the saved epoch
    with the saved optimiser / scheduler state and trains only the remaining epoch."""
    import golden_cases as gc
    from ponderv2_amd.ponder.utils.config import Config

    model = gc.outdoor_model_cfg(dict(gc.SMALL_BACKBONE, in_channels=4,
                                      channels=(16, 32, 48, 64, 64, 48, 32, 96)), **gc.OUTDOOR_SMALL)

    def cfg(**over):
        base = dict(
            weight=None, resume=False, evaluate=False, seed=9, save_path=str(tmp_path), num_worker=0,
            batch_size=2, epoch=2, eval_epoch=2, sync_bn=False, enable_amp=False, empty_cache=False,
            find_unused_parameters=True, mix_prob=0, param_dicts=None,
            hooks=[dict(type="CheckpointLoader"), dict(type="IterationTimer", warmup_iter=0),
                   dict(type="InformationWriter"), dict(type="CheckpointSaver", save_freq=1)],
            train=dict(type="MultiDatasetTrainer"), model=model,
            optimizer=dict(type="AdamW", lr=2e-4, weight_decay=0.01),
            scheduler=dict(type="OneCycleLR", max_lr=2e-4, pct_start=0.4, anneal_strategy="cos",
                           div_factor=10.0, final_div_factor=100.0),
            data=dict(train=dict(type="ConcatDataset", datasets=[
                dict(type="SyntheticLidarDataset", length=2, base_seed=700, loop=1,
                     **gc.OUTDOOR_SCENE_KW)])))
        base.update(over)
        return Config(base)

    os.makedirs(tmp_path / "model", exist_ok=True)
    threads = torch.get_num_threads()   # trainer_main pins 2 threads (it is written for spawned ranks)
    try:
        _run_and_resume(tmp_path, cfg)
    finally:
        torch.set_num_threads(threads)


def _run_and_resume(tmp_path, cfg):
    ddp_worker.trainer_main(cfg())                       # two epochs of one iteration each
    first = torch.load(tmp_path / "model" / "epoch_1.pth", weights_only=False)
    last = torch.load(tmp_path / "model" / "model_last.pth", weights_only=False)
    assert first["epoch"] == 1 and last["epoch"] == 2
    rows_full = [json.loads(l) for l in open(tmp_path / "scalars.jsonl")]
    assert len(rows_full) == 2
    os.remove(tmp_path / "scalars.jsonl")
    ddp_worker.trainer_main(cfg(weight=str(tmp_path / "model" / "epoch_1.pth"), resume=True))
    rows_resumed = [json.loads(l) for l in open(tmp_path / "scalars.jsonl")]
    assert len(rows_resumed) == 1                        # only epoch 2 was run
    again = torch.load(tmp_path / "model" / "model_last.pth", weights_only=False)
    assert again["epoch"] == 2
    # same scheduler position as the uninterrupted run
    assert again["scheduler"]["last_epoch"] == last["scheduler"]["last_epoch"] == 2
    assert again["optimizer"]["state"][0]["step"] == last["optimizer"]["state"][0]["step"]


def test_bench_spawns_its_own_ranks():
    """``python bench.py --gpus 2`` with no launcher around it re-runs itself as two ranks under
    torch.distributed.run (as the reference's launch() spawns its workers, engines/launch.py:38-100)
    and rank 0 alone reports ``n_gpus: 2``.  ``--launch-check`` exercises exactly that plumbing
    (spawn, rendezvous on 127.0.0.1, barrier, max-over-ranks reduce, one JSON line) without the
    model - over gloo here, where there is no GPU."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout            # rank 0 only
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 2 and doc["gpus_requested"] == 2 and doc["launch_check"] is True
    assert doc["max_over_ranks"] == 2.0            # the reduction saw both ranks
