#!/usr/bin/env python
"""Which Python call sites launch the ATen glue kernels of a training step (torch.profiler with
stacks), grouped by kernel family: where do the copies / adds / fills come from?"""
import collections, os, sys
import torch
from torch.profiler import ProfilerActivity, profile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from ponderv2_amd.ponder.models import build_model
from ponderv2_amd.ponder.utils.config import ConfigDict

dev = torch.device("cuda:0")
model = build_model(ConfigDict(bench.model_cfg(256, "float32"))).to(dev).train()
opt = torch.optim.SGD(model.parameters(), lr=1e-4, momentum=0.9, nesterov=True, weight_decay=1e-4)
batch = bench.make_batch(0, 2, 2, dev)
def step():
    out = model(bench.clone_batch(batch)); opt.zero_grad(set_to_none=True); out["loss"].backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ops = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
for ev in prof.key_averages(group_by_stack_n=16):
    if not ev.key.startswith("aten::"):
        continue
    self_t = ev.self_device_time_total
    if self_t <= 0:
        continue
    frames = [f for f in (ev.stack or []) if "/repo/" in f and "tools/" not in f]
    site = frames[0] if frames else "(backward: autograd engine)"
    site = site.replace(ROOT + "/", "").replace("/tmp/code/OpenGVLab__PonderV2/repo/", "")
    o = ops[ev.key]
    o[0] += ev.count; o[1] += self_t; o[2][site] += self_t
tot = sum(v[1] for v in ops.values())
print("aten ops with device time: %.2f ms" % (tot / 1e3))
for name, (n, t, sites) in sorted(ops.items(), key=lambda kv: -kv[1][1])[:16]:
    if "convolution" in name or "batch_norm" in name:
        continue
    print("%-28s %4d calls %7.3f ms" % (name, n, t / 1e3))
    for s_, st in sites.most_common(7):
        print("      %7.3f ms  %s" % (st / 1e3, s_[:150]))
