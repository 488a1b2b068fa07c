#!/usr/bin/env python
"""tests/test_gpu_grad_overlap.py's sequence in one process, reporting EVERY tensor that leaves the step's noise."""
import os, sys, socket
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ddp_worker, golden_cases as gc
from ponderv2_amd import spunet_native
from ponderv2_amd.ponder.datasets import collate_fn
from ponderv2_amd.ponder.models import build_model
from ponderv2_amd.ponder.utils.config import ConfigDict
from ponderv2_amd.ponder.utils.grad_sync import FlatGradSync

dev = torch.device("cuda:0")
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=dev)
torch.manual_seed(5)
cfg = gc.indoor_model_cfg(dict(gc.SMALL_BACKBONE, base_channels=32, channels=(32, 32, 64, 64, 64, 64, 32, 96)),
                          grid_shape=(32, 32, 8), ray_nsample=6)
model = build_model(ConfigDict(cfg)).to(dev).train()
batch = collate_fn([ddp_worker.tiny_scene(60), ddp_worker.tiny_scene(61)])
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}

def step(sync):
    torch.manual_seed(9)
    model.zero_grad(set_to_none=True)
    out = model({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
    out["loss"].backward()
    if sync is not None:
        sync.sync()
    torch.cuda.synchronize()
    return float(out["loss"]), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

l0, local = step(None)
runs = [("plain2", step(None)), ("plain3", step(None))]
sync = FlatGradSync(model.parameters(), overlap=True, slab_mb=0.5).attach()
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    runs.append(("sync%d" % k, step(sync)))
sync.detach()
for tag, (l, g) in runs:
    bad = []
    for n in local:
        sc = local[n].abs().max().item()
        d = (g[n] - local[n]).abs().max().item() / (sc + 1e-30)
        if d > 1e-3 and sc > 1e-6 and "upsample.bias" not in n:
            bad.append((d, n))
    bad.sort(reverse=True)
    print("%-7s loss %.8f (first %.8f)  %d tensors off > 1e-3: %s" % (tag, l, l0, len(bad), ", ".join("%s %.1e" % (n, d) for d, n in bad[:6])))
dist.destroy_process_group()
