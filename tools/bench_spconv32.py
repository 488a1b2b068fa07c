#!/usr/bin/env python
"""fp32 sparse conv per layer shape on the bench geometry: forward, grad-input (forward weights read
in place) and weight gradient.  Compare settings by running it under different environments
(PV2_FWD_PERSIST=0|1|2)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from ponderv2_amd import kernels as K
from ponderv2_amd.ponder.models.utils import offset2batch

dev = torch.device("cuda:0")
batch = bench.make_batch(0, 2, 2, dev)
idx = torch.cat([offset2batch(batch["offset"]).unsqueeze(-1).int(), batch["grid_coord"].int()], 1).contiguous()
geo = K.prepare_unet_geometry(idx, batch["sparse_shape"])
LAYERS = [("subm1", 32, 32), ("subm2", 64, 64), ("subm3", 128, 128), ("subm4", 256, 256), ("subm3", 384, 256),
          ("subm3", 256, 256), ("subm2", 192, 128), ("subm2", 128, 128), ("subm1", 128, 96), ("subm1", 96, 96),
          ("subm0", 128, 96), ("subm0", 96, 96)]


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


print("PV2_FWD_PERSIST=%s" % os.environ.get("PV2_FWD_PERSIST", "(default)"))
print("%-7s %4s %4s %7s %8s | fwd us  TF/s | dgrad us  TF/s | wgrad us  TF/s" % ("layer", "cin", "cout", "rows", "pairs"))
tot = [0.0, 0.0, 0.0]
for key, c_in, c_out in LAYERS:
    rb = geo[key]["rulebook"]
    x = torch.randn(rb.n_in, c_in, device=dev)
    g = torch.randn(rb.n_out, c_out, device=dev)
    w = torch.randn(c_out, rb.K, c_in, device=dev) * 0.05
    fl = 2.0 * rb.n_pairs * c_in * c_out
    t_f = timeit(lambda: K.spconv_forward(x, w, rb))
    t_d = timeit(lambda: K.spconv_grad_input(g, w, rb))
    t_w = timeit(lambda: K.spconv_backward_weight(x, g, rb, c_out))
    tot = [tot[0] + t_f, tot[1] + t_d, tot[2] + t_w]
    print("%-7s %4d %4d %7d %8d | %6.1f %5.1f | %7.1f %5.1f | %7.1f %5.1f" % (
        key, c_in, c_out, rb.n_out, rb.n_pairs, t_f, fl / t_f / 1e6, t_d, fl / t_d / 1e6, t_w, fl / t_w / 1e6))
print("sum: fwd %.0f us, dgrad %.0f us, wgrad %.0f us" % tuple(tot))
