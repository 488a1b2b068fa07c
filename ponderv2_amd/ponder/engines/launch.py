"""One process per GPU (ponder/engines/launch.py:38-140): ``launch(main, num_gpus_per_machine)``
spawns the workers and initialises ``torch.distributed`` - backend "nccl" (= RCCL over xGMI on
MI355X) when GPUs are visible, gloo otherwise.  A job started by ``torch.distributed.run`` (RANK /
WORLD_SIZE in the environment) is adopted instead of spawning."""
import os
import socket
from datetime import timedelta

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ..utils import comm

DEFAULT_TIMEOUT = timedelta(minutes=30)


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init(rank, world_size, dist_url, local_rank, timeout):
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(local_rank)
    dist.init_process_group(backend="nccl" if use_gpu else "gloo", init_method=dist_url,
                            world_size=world_size, rank=rank, timeout=timeout)
    comm.synchronize()


def _worker(local_rank, main_func, world_size, gpus_per_machine, machine_rank, dist_url, cfg,
            timeout):
    _init(machine_rank * gpus_per_machine + local_rank, world_size, dist_url, local_rank, timeout)
    try:
        main_func(*cfg)
    finally:
        comm.synchronize()
        dist.destroy_process_group()


def launch(main_func, num_gpus_per_machine, num_machines=1, machine_rank=0, dist_url=None, cfg=(),
           timeout=DEFAULT_TIMEOUT):
    world_size = num_machines * num_gpus_per_machine
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        _init(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), "env://",
              int(os.environ.get("LOCAL_RANK", 0)), timeout)
        try:
            return main_func(*cfg)
        finally:
            dist.destroy_process_group()
    if world_size <= 1:
        return main_func(*cfg)
    if dist_url in (None, "auto"):
        assert num_machines == 1, "dist_url=auto is not supported in multi-machine jobs"
        dist_url = f"tcp://127.0.0.1:{_free_port()}"
    # (the spawned ranks carry an RCCL communicator: two hardware queues - bench.py's header has the
    # measurements; set by the LAUNCHER, before the ranks' HIP runtimes initialise, not by the library)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
    mp.spawn(_worker, nprocs=num_gpus_per_machine, daemon=False,
             args=(main_func, world_size, num_gpus_per_machine, machine_rank, dist_url, cfg, timeout))
