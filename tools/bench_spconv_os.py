#!/usr/bin/env python
"""Per-layer-shape A/B of the sparse-conv forward kernels on the bench scene (2 scenes): the
pair-major scatter-add kernel (incl. its zero-fill) against the output-stationary kernel, forward and
grad-input orientation, plus strided / inverse convs."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ponderv2_amd import kernels as K
from ponderv2_amd.ponder.datasets import collate_fn, make_scene

dev = torch.device("cuda:0")
b = collate_fn([make_scene(1000 + i, num_views=1, image_hw=(24, 32)) for i in range(2)])
batch = torch.repeat_interleave(torch.arange(2), torch.diff(b["offset"], prepend=torch.zeros(1, dtype=torch.long)))
coords = torch.cat([batch[:, None], b["grid_coord"]], 1).int().to(dev)
levels, downs = [coords], []
shape = [int(v) + 96 for v in b["grid_coord"].max(0).values]
for l in range(4):
    shape = [(s - 2) // 2 + 1 for s in shape]
    rb, oc = K.build_downsample_rulebook(levels[-1], 2, shape)
    levels.append(oc); downs.append(rb)
rbs = [K.build_subm_rulebook(c, 3) for c in levels]
stem = K.build_subm_rulebook(coords, 5)
print("levels", [len(c) for c in levels], "pairs", [r.n_pairs for r in rbs], "stem", stem.n_pairs)

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

def ab(name, rb, cin, cout, kk):
    x = torch.randn(rb.n_in, cin, device=dev); w = torch.randn(cout, kk, cin, device=dev) * 0.05
    fl = 2.0 * rb.n_pairs * cin * cout
    res = []
    for use, osl in ((False, False), (True, False), (True, True)):
        K.USE_OS, K.USE_OSL = use, osl
        t = timeit(lambda: K.spconv_forward(x, w, rb))
        res.append(t)
    K.USE_OS, K.USE_OSL = "auto", True
    print("%-22s %4d->%4d K%3d pairs %7d rows %6d | scatter %7.1f us %6.1f TF | gather-table OS %7.1f us %6.1f TF | "
          "LDS-tile OS %7.1f us %6.1f TF | x%.2f"
          % (name, cin, cout, kk, rb.n_pairs, rb.n_out, res[0], fl / res[0] / 1e6, res[1], fl / res[1] / 1e6,
             res[2], fl / res[2] / 1e6, res[0] / res[2]), flush=True)

ab("stem k5", stem, 6, 32, 125)
for lvl, cin, cout in [(0, 96, 96), (0, 128, 96), (1, 32, 32), (1, 96, 96), (1, 128, 96), (2, 64, 64), (2, 128, 128),
                       (2, 192, 128), (3, 128, 128), (3, 256, 256), (3, 384, 256), (4, 256, 256)]:
    ab("subm L%d fwd" % lvl, rbs[lvl], cin, cout, 27)
    ab("subm L%d dgrad" % lvl, rbs[lvl].transposed(), cout, cin, 27)
for l, (cin, cout) in enumerate([(32, 32), (32, 64), (64, 128), (128, 256)]):
    ab("down L%d->%d" % (l, l + 1), downs[l], cin, cout, 8)
    ab("inverse L%d->%d" % (l + 1, l), downs[l].transposed(), cout, cin, 8)
