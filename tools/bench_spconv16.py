#!/usr/bin/env python
"""Per-layer timing of the 16-bit sparse-conv kernels on the bench geometry (configs[1]: 2 scenes),
over the tuning variants of pv2_spconv16_os_forward (pv2_debug_set_os16_variant)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from ponderv2_amd import _lib, kernels as K

dev = torch.device("cuda:0")
batch = bench.make_batch(0, 2, 2, dev)
grid = batch["grid_coord"].int()
from ponderv2_amd.ponder.models.utils import offset2batch
b = offset2batch(batch["offset"])
idx = torch.cat([b.unsqueeze(-1).int(), grid], 1).contiguous()
shape = batch.get("sparse_shape") or torch.add(grid.max(0).values, 96).tolist()
PERM = "--perm" in sys.argv
if PERM:
    K.USE_OS = True
geo = K.prepare_unet_geometry(idx, shape)
# the layer shapes of SpUNet-v1m1 (32,64,128,256,256,128,96,96) on the five levels of the bench batch
LAYERS = [("subm1", 32, 32), ("subm2", 64, 64), ("subm3", 128, 128), ("subm4", 256, 256), ("subm3", 384, 256),
          ("subm3", 256, 256), ("subm2", 192, 128), ("subm2", 128, 128), ("subm1", 128, 96), ("subm1", 96, 96),
          ("subm0", 128, 96), ("subm0", 96, 96), ("spconv1", 32, 32), ("spconv2", 32, 64), ("spconv3", 64, 128),
          ("spconv4", 128, 256)]
dt = torch.bfloat16
L = _lib.lib()


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


variants = [0]
print("%-9s %4s %4s %7s %8s | " % ("layer", "cin", "cout", "rows", "pairs") + " ".join("v%d fwd/dgrad us" % v for v in variants) + " | wgrad us scalar / tr | weights MB(32-row)")
for key, c_in, c_out in LAYERS:
    rb = geo[key]["rulebook"]
    x = torch.randn(rb.n_in, c_in, device=dev).to(dt)
    g = torch.randn(rb.n_out, c_out, device=dev).to(dt)
    w = torch.randn(c_out, rb.K, c_in, device=dev) * 0.05
    fwd, bwd = K.packed_weights(w, dt)
    tn, ts, tp, tk = rb._transposed_os
    cells = []
    for v in variants:
        L.pv2_debug_set_os16_variant(v)
        t_f = timeit(lambda: K.spconv16_forward(x, fwd, rb.K, c_out, rb.nbr, rb.nbr_stride, rb.perm, rb.kflip, rb.n_out))
        t_b = timeit(lambda: K.spconv16_forward(g, bwd, rb.K, c_in, tn, ts, tp, tk, rb.n_in))
        cells.append("%6.1f/%6.1f" % (t_f, t_b))
    L.pv2_debug_set_os16_variant(8)   # weight gradient with scalar LDS reads
    t_w0 = timeit(lambda: K.spconv16_backward_weight(x, g, rb, c_out))
    L.pv2_debug_set_os16_variant(0)   # ... with ds_read_b64_tr_b16
    t_w = timeit(lambda: K.spconv16_backward_weight(x, g, rb, c_out))
    mb = (rb.n_out + 31) // 32 * rb.K * c_in * c_out * 2 / 1e6
    print("%-9s %4d %4d %7d %8d | " % (key, c_in, c_out, rb.n_out, rb.n_pairs) + "   ".join(cells) + " | %7.1f / %7.1f | %7.1f" % (t_w0, t_w, mb))
