#!/usr/bin/env python
"""Run-to-run noise of the tiny indoor model's gradients (tests/test_gpu_grad_overlap.py's setup, no process
group, no gradient sync): max relative deviation per tensor over N steps from the first."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ddp_worker, golden_cases as gc
from ponderv2_amd.ponder.datasets import collate_fn
from ponderv2_amd.ponder.models import build_model
from ponderv2_amd.ponder.utils.config import ConfigDict

dev = torch.device("cuda:0")
torch.manual_seed(5)
cfg = gc.indoor_model_cfg(dict(gc.SMALL_BACKBONE, base_channels=32, channels=(32, 32, 64, 64, 64, 64, 32, 96)),
                          grid_shape=(32, 32, 8), ray_nsample=6)
model = build_model(ConfigDict(cfg)).to(dev).train()
batch = collate_fn([ddp_worker.tiny_scene(60), ddp_worker.tiny_scene(61)])
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}

SYNC = None
if "--sync" in sys.argv:      # the overlapped slab reduction under a one-rank RCCL group, as the test
    import torch.distributed as dist
    from ponderv2_amd.ponder.utils.grad_sync import FlatGradSync
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", world_size=1, rank=0, device_id=dev)


def step():
    torch.manual_seed(9)
    model.zero_grad(set_to_none=True)
    out = model({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
    out["loss"].backward()
    if SYNC is not None:
        SYNC.sync()
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

ref = step()
if "--sync" in sys.argv:
    SYNC = FlatGradSync(model.parameters(), overlap=True, slab_mb=0.5).attach()
worst = {}
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for i in range(N):
    g = step()
    for n in ref:
        diff = (g[n] - ref[n]).abs()
        d = diff.max().item() / (ref[n].abs().max().item() + 1e-30)
        if d > worst.get(n, (0, 0))[0]:
            worst[n] = (d, i)
        if n == "backbone.down.0.1.weight" and d > 2e-3:
            k = int(diff.flatten().argmax())
            print("run %d: %s rel %.2e; %d of %d elements off by > 1e-3 of the max; element %d: %.6f vs %.6f"
                  % (i, n, d, int((diff > 1e-3 * ref[n].abs().max()).sum()), diff.numel(), k,
                     g[n].flatten()[k].item(), ref[n].flatten()[k].item()))
for n, (d, i) in sorted(worst.items(), key=lambda kv: -kv[1][0])[:12]:
    print("%.2e (run %d)  %s" % (d, i, n))
