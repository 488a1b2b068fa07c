#!/usr/bin/env python
"""Micro-benchmark of the sparse-conv kernels on the bench scene (2 scenes): scatter-add forward, the
product-row route and the weight gradient per layer shape (PV2_FP32_MFMA=1: the fp32-MFMA kernels)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ponderv2_amd import kernels as K, _lib
from ponderv2_amd.ponder.datasets import collate_fn, make_scene

dev = torch.device("cuda:0")
b = collate_fn([make_scene(1000 + i, num_views=1, image_hw=(24, 32)) for i in range(2)])
batch = torch.repeat_interleave(torch.arange(2), torch.diff(b["offset"], prepend=torch.zeros(1, dtype=torch.long)))
coords = torch.cat([batch[:, None], b["grid_coord"]], 1).int().to(dev)
levels = [coords]
shape = [int(v) + 96 for v in b["grid_coord"].max(0).values]
for l in range(4):
    shape = [(s - 2) // 2 + 1 for s in shape]
    rb, oc = K.build_downsample_rulebook(levels[-1], 2, shape)
    levels.append(oc)
rbs = [K.build_subm_rulebook(c, 3) for c in levels]
print("levels", [len(c) for c in levels], "pairs", [r.n_pairs for r in rbs])

def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

L = _lib.lib()
cases = [(1, 64, 64), (2, 128, 128), (3, 256, 256), (4, 256, 256), (0, 96, 96), (1, 32, 32)]
for lvl, cin, cout in cases:
    rb = rbs[lvl]
    x = torch.randn(rb.n_in, cin, device=dev); w = torch.randn(cout, 27, cin, device=dev) * 0.05
    out = torch.zeros(rb.n_out, cout, device=dev)
    g = torch.randn(rb.n_out, cout, device=dev)
    fl = 2.0 * rb.n_pairs * cin * cout
    t = timeit(lambda: K.spconv_forward(x, w, rb, out=out))
    row = ["%.1fus %.1fTF" % (t, fl / t / 1e6)]
    tp = timeit(lambda: K.spconv_forward(x, w, rb))          # product rows + ordered reduce (default route)
    row.append("pr %.1fus %.1fTF" % (tp, fl / tp / 1e6))
    tw = timeit(lambda: K.spconv_backward_weight(x, g, rb, cout))
    print("L%d %3d->%3d pairs %6d | fwd: %s | wgrad %.1fus %.1fTF" % (lvl, cin, cout, rb.n_pairs, " | ".join(row), tw, fl / tw / 1e6))
