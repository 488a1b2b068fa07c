#!/usr/bin/env python
"""Where do the kernel launches of a training step come from?  One step under torch.profiler;
launches (kernels + async copies / memsets) are counted per phase of the forward pass (host
``record_function`` ranges) and, for the backward pass, per autograd node type."""
import collections, os, sys
import torch
from torch.profiler import ProfilerActivity, profile, record_function
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from ponderv2_amd.ponder.models import build_model
from ponderv2_amd.ponder.utils.config import ConfigDict

dev = torch.device("cuda:0")
model = build_model(ConfigDict(bench.model_cfg(256, "float32"))).to(dev).train()
opt = torch.optim.SGD(model.parameters(), lr=1e-4, momentum=0.9, nesterov=True, weight_decay=1e-4, fused=True)
batch = bench.make_batch(0, 2, 2, dev)
staged = [model.prefetch(bench.clone_batch(batch))]


def step(tag=False):
    rf = record_function if tag else (lambda name: __import__("contextlib").nullcontext())
    cur = staged.pop()
    with rf("phase:stage_next_batch(prefetch geometry)"):
        staged.append(model.prefetch(bench.clone_batch(batch)))
    with rf("phase:extract_feature(backbone fwd)"):
        d = model.extract_feature(cur)
    with rf("phase:prepare_ray"):
        ray = d.pop("_ray_dict", None)      # (set up with the batch when prefetched)
        if ray is None:
            ray, d = model.prepare_ray(d)
    with rf("phase:prepare_volume(to_dense + UNet3D)"):
        vol = model.prepare_volume(d)
    with rf("phase:render"):
        out = model.render_func(ray, vol)
    with rf("phase:losses"):
        res = model.render_loss(out, ray)
    loss = res[0] if isinstance(res, (tuple, list)) else res["loss"]
    with rf("phase:zero_grad"):
        opt.zero_grad(set_to_none=True)
    with rf("phase:backward"):
        loss.backward()
    with rf("phase:optimizer"):
        opt.step()


for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(tag=True)
    torch.cuda.synchronize()


def launches(ev):
    return len(ev.kernels) + sum(launches(c) for c in ev.cpu_children)


events = prof.events()
phases = [e for e in events if e.name.startswith("phase:")]
total = 0
for ph in phases:
    n = launches(ph)
    total += n
    print("%-46s %5d launches  host %.2f ms" % (ph.name[6:], n, ph.cpu_time_total / 1e3))
    groups = collections.Counter()
    hosts = collections.Counter()
    for c in ph.cpu_children:
        name = c.name.replace("autograd::engine::evaluate_function: ", "bwd ")
        groups[name] += launches(c)
        hosts[name] += c.cpu_time_total
    for name, k in groups.most_common(24):
        if k:
            print("        %5d  %-60s host %.2f ms" % (k, name[:60], hosts[name] / 1e3))
# the backward pass runs on the autograd engine's device thread: its nodes are no children of the
# host range above; group them by node type wherever they ran
nodes = [e for e in events if e.name.startswith("autograd::engine::evaluate_function: ")]
groups, hosts, counts = collections.Counter(), collections.Counter(), collections.Counter()
for e in nodes:
    name = e.name.split(": ", 1)[1]
    groups[name] += launches(e)
    hosts[name] += e.cpu_time_total
    counts[name] += 1
n_b = sum(groups.values())
total += n_b
print("%-46s %5d launches  host %.2f ms  (%d nodes)" % ("backward (autograd nodes, engine thread)", n_b,
                                                        sum(hosts.values()) / 1e3, sum(counts.values())))
for name, k in sorted(hosts.items(), key=lambda kv: -kv[1])[:40]:
    print("        %5d launches %4d nodes  %-52s host %.2f ms" % (groups[name], counts[name], name[:52], k / 1e3))
print("total launches in the step:", total)


# the copies: which op asked for them, on what shapes
def chain(e):
    names = []
    while e is not None and len(names) < 6:
        if not e.name.startswith(("phase:", "autograd::engine")):
            names.append(e.name + (str(e.input_shapes)[:60] if e.input_shapes else ""))
        e = e.cpu_parent
    return " <- ".join(names)


copies = collections.Counter()
times = collections.Counter()
for e in events:
    for k in e.kernels:
        if "direct_copy" in k.name or "copyBuffer" in k.name or "gather" in k.name or "BinaryFunctor" in k.name:
            key = chain(e)
            copies[key] += 1
            times[key] += k.duration
# fills and device-to-device copies: every one is a launch on the training stream's chain
small = collections.Counter()
for e in events:
    for k in e.kernels:
        if "FillFunctor" in k.name or "Memcpy" in k.name or "Memset" in k.name or "copyBuffer" in k.name \
                or "fillBuffer" in k.name:
            small[("fill  " if ("Fill" in k.name or "Memset" in k.name or "fillBuffer" in k.name) else "copy  ")
                  + chain(e)] += 1
print("fills / copies by caller (count):")
for key, n in small.most_common(60):
    print("  %3d  %s" % (n, key[:250]))
print("copy / gather / big elementwise kernels by caller (us, count):")
for key, t in times.most_common(24):
    print("  %8.1f %3d  %s" % (t, copies[key], key[:230]))
