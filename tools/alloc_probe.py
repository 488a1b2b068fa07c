#!/usr/bin/env python
"""Does the caching allocator settle?  reserved / allocated bytes and segment count every 10 steps of the bench step."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from ponderv2_amd.ponder.models import build_model
from ponderv2_amd.ponder.utils.config import ConfigDict
from ponderv2_amd.ponder.datasets.voxelize import input_stream

dev = torch.device("cuda:0")
model = build_model(ConfigDict(bench.model_cfg(256, "float32"))).to(dev).train()
opt = torch.optim.SGD(model.parameters(), lr=1e-4, momentum=0.9, nesterov=True, weight_decay=1e-4, fused=True)
batches = [bench.make_batch(i, 2, 2, dev) for i in range(8)]
k = [0]
def stage():
    with input_stream(dev) as pipe:
        b = model.prefetch(bench.clone_batch(batches[k[0] % len(batches)])); k[0] += 1
        return pipe.adopt(b)
staged = [stage()]
def step():
    cur = staged.pop(); staged.append(stage())
    out = model(cur); opt.zero_grad(set_to_none=True); out["loss"].backward(); opt.step()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    step()
    if i % 10 == 9:
        torch.cuda.synchronize()
        st = torch.cuda.memory_stats(dev)
        print("step %3d  reserved %.2f GB  allocated %.2f GB  segments ever %d  live segments %d  large-pool segments ever %d"
              % (i + 1, st["reserved_bytes.all.current"] / 2**30, st["allocated_bytes.all.current"] / 2**30,
                 st["segment.all.allocated"], st["segment.all.current"], st["segment.large_pool.allocated"]))
