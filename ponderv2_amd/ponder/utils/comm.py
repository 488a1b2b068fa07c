"""Rank / world helpers and a barrier (ponder/utils/comm.py:24-101, the subset the pre-training
loop uses).  ``backend="nccl"`` is RCCL on ROCm; CPU-only runs use gloo."""
import torch
import torch.distributed as dist


def _ready():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if _ready() else 1


def get_rank():
    return dist.get_rank() if _ready() else 0


def is_main_process():
    return get_rank() == 0


def synchronize():
    if get_world_size() == 1:
        return
    if dist.get_backend() == dist.Backend.NCCL:
        dist.barrier(device_ids=[torch.cuda.current_device()])
    else:
        dist.barrier()


def reduce_dict(input_dict, average=True):
    """All-reduce a dict of scalar tensors (sum or mean over ranks)."""
    world = get_world_size()
    if world < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict)
        vals = torch.stack([input_dict[k].detach().float() for k in names])
        dist.all_reduce(vals)
        if average:
            vals /= world
        return dict(zip(names, vals))
