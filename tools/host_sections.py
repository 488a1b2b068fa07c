#!/usr/bin/env python
"""Host time (perf_counter, no profiler) of the Python sections around the native U-Net executor in steady state:
plan building, op-record filling, the C calls themselves."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from ponderv2_amd import spunet_native as sn, _lib
from ponderv2_amd.ponder.models import build_model
from ponderv2_amd.ponder.utils.config import ConfigDict

dev = torch.device("cuda:0")
model = build_model(ConfigDict(bench.model_cfg(256, "float32"))).to(dev).train()
opt = torch.optim.SGD(model.parameters(), lr=1e-4, momentum=0.9, nesterov=True, weight_decay=1e-4, fused=True)
batch = bench.make_batch(0, 2, 2, dev)
acc = {}
def wrap(obj, name, key):
    fn = getattr(obj, name)
    def inner(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc[key] = acc.get(key, 0.0) + time.perf_counter() - t; return r
    setattr(obj, name, inner)
wrap(sn, "build_plan", "build_plan")
wrap(sn, "_fill_forward", "_fill_forward")
wrap(sn, "run", "run (all of the forward side)")
L = _lib.lib()
for name in ("pv2_unet_forward", "pv2_unet_backward", "pv2_unet_backward_ev"):
    fn = getattr(L, name)
    def mk(fn, name):
        def inner(*a):
            t = time.perf_counter(); r = fn(*a); acc["C " + name] = acc.get("C " + name, 0.0) + time.perf_counter() - t; return r
        return inner
    setattr(L, name, mk(fn, name))
orig_bwd = sn.SpUNetFunction.backward
def timed_bwd(ctx, g):
    t = time.perf_counter(); r = orig_bwd(ctx, g); acc["SpUNetFunction.backward"] = acc.get("SpUNetFunction.backward", 0.0) + time.perf_counter() - t; return r
sn.SpUNetFunction.backward = staticmethod(timed_bwd)
staged = [model.prefetch(bench.clone_batch(batch))]
def step():
    cur = staged.pop(); staged.append(model.prefetch(bench.clone_batch(batch)))
    out = model(cur); opt.zero_grad(set_to_none=True); out["loss"].backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize(); acc.clear()
N = 20
t0 = time.perf_counter()
for _ in range(N): step()
host = time.perf_counter() - t0
torch.cuda.synchronize()
print("host enqueue per step: %.2f ms" % (host / N * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-36s %.3f ms per step" % (k, v / N * 1e3))
