"""Does a one-rank RCCL all_reduce (or waiting for it) block the HOST?  A long-running kernel is queued
first; host time of the enqueue and of the wait tell."""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
x = torch.randn(8192, 8192, device=dev)
flat = torch.zeros(40_000_000, device=dev)
def busy():
    for _ in range(6): (x @ x)
for mode in ("async+wait", "sync_op", "async_nowait+stream_wait"):
    for rep in range(3):
        torch.cuda.synchronize(); busy(); t0 = time.perf_counter()
        if mode == "async+wait":
            h = dist.all_reduce(flat, async_op=True); t1 = time.perf_counter(); h.wait()
        elif mode == "sync_op":
            dist.all_reduce(flat); t1 = time.perf_counter()
        else:
            h = dist.all_reduce(flat, async_op=True); t1 = time.perf_counter()
            fut = h.get_future() if hasattr(h, "get_future") else None
        t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
        print("%-28s enqueue %.3f ms  wait %.3f ms  (gpu drain %.1f ms)" % (mode, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
dist.destroy_process_group()
