#!/usr/bin/env python
"""cProfile of the host side of training steps (where does the enqueue time go?)."""
import cProfile, io, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from ponderv2_amd.ponder.models import build_model
from ponderv2_amd.ponder.utils.config import ConfigDict
from ponderv2_amd.ponder.utils.optimizer import build_optimizer

dev = torch.device("cuda:0")
OUTDOOR = "--outdoor" in sys.argv
AMP = torch.bfloat16 if "--amp" in sys.argv else None  # the scoped reduced-precision mode (bench.py --amp bf16)
if OUTDOOR:
    model = build_model(ConfigDict(bench.outdoor_model_cfg())).to(dev).train()
    opt = build_optimizer(dict(type="AdamW", lr=2e-4, weight_decay=0.01), model)
    batch = bench.make_outdoor_batch(0, 4, 512, dev)
else:
    model = build_model(ConfigDict(bench.model_cfg(256, "float32"))).to(dev).train()
    opt = build_optimizer(dict(type="SGD", lr=1e-4, momentum=0.9, nesterov=True, weight_decay=1e-4), model)
    batch = bench.make_batch(0, 2, 2, dev)
PREFETCH = "--prefetch" in sys.argv
from ponderv2_amd.ponder.datasets.voxelize import input_stream
def stage():   # as bench.py / engines/train.py: on the input stream
    with input_stream(dev) as pipe:
        b = bench.clone_batch(batch)
        if PREFETCH:
            b = model.prefetch(b)
        return pipe.adopt(b)
staged = [stage()]
def step():
    cur = staged.pop()
    staged.append(stage())
    with torch.autocast("cuda", dtype=AMP or torch.bfloat16, enabled=AMP is not None):
        out = model(cur)
    opt.zero_grad(set_to_none=True); out["loss"].backward(); opt.step(); return out
for _ in range(4): step()
torch.cuda.synchronize()
# section timing on the host (enqueue only)
import contextlib
def sect():
    model._ambient_amp = AMP
    d = bench.clone_batch(batch); t = [time.perf_counter()]
    d = model.extract_feature(d); t.append(time.perf_counter())
    if OUTDOOR:
        ray = model.prepare_ray(d); t.append(time.perf_counter())
        vol = model.prepare_volume(d); t.append(time.perf_counter())
        B = vol[0].shape[0]
        ray = dict(ray)
    else:
        ray, d = model.prepare_ray(d); t.append(time.perf_counter())
        vol = model.prepare_volume(d); t.append(time.perf_counter())
    out = model.render_func(ray, vol); t.append(time.perf_counter())
    res = model.render_loss(out, ray); t.append(time.perf_counter())
    # backward phases: the moment the gradient reaches each section boundary (engine thread clock)
    stamps = []
    def mark(name, x):
        x = getattr(x, "pre", x)
        if torch.is_tensor(x) and x.requires_grad:
            x.register_hook(lambda g, name=name: stamps.append((name, time.perf_counter())))
    for k, v in out.items():
        mark("render_out", v)
    mark("volume", vol[0]); mark("backbone_feat", d["sparse_backbone_feat"])
    first = next(p for n, p in model.named_parameters() if "conv_input" in n)
    mark("first_layer_weight", first)
    opt.zero_grad(set_to_none=True); tb = time.perf_counter(); res[0].backward(); t.append(time.perf_counter())
    seen = {}
    for n, ts in stamps:
        seen.setdefault(n, []).append(ts)
    print("   backward: " + " | ".join("%s %.2f..%.2f" % (n, 1e3 * (min(v) - tb), 1e3 * (max(v) - tb))
                                       for n, v in seen.items()) + " | returns %.2f" % (1e3 * (t[-1] - tb)), flush=True)
    opt.step(); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    names = ["backbone_fwd", "prepare_ray", "prepare_volume", "render", "losses", "backward", "opt", "drain"]
    print(" | ".join("%s %.2f" % (n, 1e3 * (b - a)) for n, a, b in zip(names, t[:-1], t[1:])), flush=True)
for _ in range(3): sect()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("10 steps: host %.2f ms/step, wall %.2f ms/step" % ((t1 - t0) * 100, (t2 - t0) * 100), flush=True)
# the same loop with host stamps inside: which part takes longer in the steady state than from an idle
# device (= where the host waits for the device)?
acc = [0.0] * 5
def step_stamped():
    t = [time.perf_counter()]
    cur = staged.pop()
    staged.append(stage()); t.append(time.perf_counter())
    with torch.autocast("cuda", dtype=AMP or torch.bfloat16, enabled=AMP is not None):
        out = model(cur)
    t.append(time.perf_counter())
    opt.zero_grad(set_to_none=True); t.append(time.perf_counter())
    out["loss"].backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    for i in range(5):
        acc[i] += t[i + 1] - t[i]
for _ in range(3): step_stamped()
torch.cuda.synchronize(); acc = [0.0] * 5; t0 = time.perf_counter()
for _ in range(10): step_stamped()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("steady state, host ms per step: " + " | ".join("%s %.2f" % (n, a * 100) for n, a in zip(
    ["stage+prefetch", "forward", "zero_grad", "backward", "opt"], acc))
    + " | host %.2f wall %.2f" % ((t1 - t0) * 100, (t2 - t0) * 100), flush=True)
# ... and the forward's sections in the steady state (wrapped methods, host clock)
import functools
from ponderv2_amd import spunet_native
sec = {}
def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name
    @functools.wraps(fn)
    def inner(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            sec[label] = sec.get(label, 0.0) + time.perf_counter() - t
    setattr(obj, name, inner)
for nm in ("extract_feature", "prepare_ray", "prepare_volume", "render_func", "render_loss"):
    wrap(model, nm)
wrap(model.backbone, "_geometry", "  backbone._geometry")
wrap(spunet_native, "run", "  spunet_native.run")
if hasattr(model, "_mask_blocks"):
    wrap(model, "_mask_blocks", "  mask_blocks")
for _ in range(3): step()
torch.cuda.synchronize(); sec.clear()
for _ in range(10): step()
torch.cuda.synchronize()
print("steady state, forward sections (host ms per step): " + " | ".join("%s %.2f" % (k.strip(), v * 100) for k, v in sec.items()), flush=True)
# ... and the DEVICE clock of the training stream at the same boundaries (events; backward phases through
# tensor hooks): per section, the time the stream spent = its kernels + whatever it waited for
for nm in ("extract_feature", "prepare_ray", "prepare_volume", "render_func", "render_loss"):
    pass
evs = []
def ev(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append((name, e))
def step_events():
    evs.clear(); ev("start")
    cur = staged.pop(); staged.append(stage())
    model._ambient_amp = AMP
    d = model.extract_feature(cur); ev("backbone_fwd")
    ray = d.pop("_ray_dict", None)      # (set up with the batch when prefetched)
    if ray is None:
        ray, d = model.prepare_ray(d)
    ev("prepare_ray")
    vol = model.prepare_volume(d); ev("prepare_volume")
    out = model.render_func(ray, vol); ev("render")
    res = model.render_loss(out, ray); ev("losses")
    def mark(name, x):
        x = getattr(x, "pre", x)
        if torch.is_tensor(x) and x.requires_grad:
            x.register_hook(lambda g, name=name: ev("bwd:" + name))
    mark("render+losses", out["rgb"]); mark("field_render", vol[0]); mark("dense_unet+cells", d["sparse_backbone_feat"])
    opt.zero_grad(set_to_none=True); res[0].backward(); ev("bwd:backbone")
    opt.step(); ev("optimizer")
for _ in range(3): step_events()
tot = {}
for _ in range(8):
    step_events(); torch.cuda.synchronize()
    seen = set()
    for (n0, e0), (n1, e1) in zip(evs[:-1], evs[1:]):
        key = n1 if n1 not in seen else n1 + "'"
        seen.add(n1)
        tot[key] = tot.get(key, 0.0) + e0.elapsed_time(e1)
print("device time of the training stream per section (ms per step, 8 steps, each step drained): "
      + " | ".join("%s %.2f" % (k, v / 8) for k, v in tot.items()) + " | sum %.2f" % (sum(tot.values()) / 8), flush=True)
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45); print(s.getvalue()[:9000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats("repo/", 60); print(s.getvalue()[:12000])
