"""Argument / config plumbing and the DDP wrapper (ponder/engines/defaults.py: create_ddp_model
:22-43, default_argument_parser :62-108, default_config_parser :111-130, default_setup :133-154)."""
import argparse
import os
import random

import numpy as np
import torch
from torch.nn.parallel import DistributedDataParallel

from ..utils import comm
from ..utils.config import Config, DictAction


def create_ddp_model(model, *, fp16_compression=False, **kwargs):
    """Plain module on one process; DistributedDataParallel (bucketed gradient all-reduce over
    RCCL, overlapped with backward) otherwise."""
    if comm.get_world_size() == 1:
        return model
    if "device_ids" not in kwargs and next(model.parameters()).is_cuda:
        kwargs["device_ids"] = [torch.cuda.current_device()]
        kwargs.setdefault("output_device", torch.cuda.current_device())
    kwargs.setdefault("gradient_as_bucket_view", True)  # gradients live in the buckets: no copy
    # DDP's hooks read every gradient the moment autograd produces it: weight gradients must then
    # be complete on the main stream, not in flight on the backward side stream
    from ponderv2_amd import sidestream
    sidestream.disable("DistributedDataParallel reads gradients during the backward pass")
    ddp = DistributedDataParallel(model, **kwargs)
    if fp16_compression:
        from torch.distributed.algorithms.ddp_comm_hooks import default as comm_hooks

        ddp.register_comm_hook(state=None, hook=comm_hooks.fp16_compress_hook)
    return ddp


def default_argument_parser(epilog=None):
    p = argparse.ArgumentParser(epilog=epilog, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--config-file", default="", metavar="FILE", help="path to config file")
    p.add_argument("--num-gpus", type=int, default=1, help="number of gpus *per machine*")
    p.add_argument("--num-machines", type=int, default=1)
    p.add_argument("--machine-rank", type=int, default=0)
    p.add_argument("--dist-url", default="auto")
    p.add_argument("--options", nargs="+", action=DictAction, help="custom options")
    return p


def set_seed(seed=None):
    if seed is None:
        seed = os.getpid() + int.from_bytes(os.urandom(2), "big")
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    return seed


def worker_init_fn(worker_id, num_workers, rank, seed):
    """DataLoader workers get the seed num_workers * rank + worker_id + seed
    (ponder/engines/defaults.py:46-59), so a fixed ``cfg.seed`` reproduces the augmentation stream."""
    set_seed(num_workers * rank + worker_id + seed)


def default_config_parser(file_path, options):
    cfg = Config.fromfile(file_path)
    if options is not None:
        cfg.merge_from_dict(options)
    if cfg.get("seed") is None:
        cfg.seed = set_seed(None)
    cfg.data.train.loop = cfg.epoch // cfg.eval_epoch
    os.makedirs(os.path.join(cfg.save_path, "model"), exist_ok=True)
    if not cfg.get("resume", False):
        cfg.dump(os.path.join(cfg.save_path, "config.py"))
    return cfg


def default_setup(cfg):
    world = comm.get_world_size()
    cfg.num_worker = cfg.num_worker if cfg.num_worker is not None else os.cpu_count()
    cfg.num_worker_per_gpu = cfg.num_worker // world
    assert cfg.batch_size % world == 0
    cfg.batch_size_per_gpu = cfg.batch_size // world
    cfg.batch_size_val_per_gpu = (cfg.batch_size_val // world) if cfg.get("batch_size_val") else 1
    cfg.batch_size_test_per_gpu = (cfg.batch_size_test // world) if cfg.get("batch_size_test") else 1
    rank = comm.get_rank()
    seed = None if cfg.seed is None else cfg.seed * max(cfg.num_worker_per_gpu, 1) + rank
    set_seed(seed)
    return cfg
