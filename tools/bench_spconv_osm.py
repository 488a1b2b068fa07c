#!/usr/bin/env python
"""A/B of the sparse-conv routes on the bench geometry (2 scenes, 46.8 k voxels): product rows + ordered
reduce (round 3/4 default) against the mask-grouped output-stationary kernel (round 5) in every tile
shape, forward and grad-input, per layer shape of SpUNet-v1m1.  us per call (launches of one conv)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("PV2_CONV_OSM", "1")
import bench
from ponderv2_amd import kernels as K, _lib
from ponderv2_amd.ponder.models.utils import offset2batch

dev = torch.device("cuda:0")
L = _lib.lib()
batch = bench.make_batch(0, 2, 2, dev)
idx = torch.cat([offset2batch(batch["offset"]).unsqueeze(-1).int(), batch["grid_coord"].int()], 1).contiguous()
geo = K.prepare_unet_geometry(idx, batch["sparse_shape"])


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


cases = [("subm1", 32, 32), ("subm2", 64, 64), ("subm3", 128, 128), ("subm4", 256, 256),
         ("subm3", 384, 256), ("subm3", 256, 256), ("subm2", 192, 128), ("subm2", 128, 128),
         ("subm1", 128, 96), ("subm1", 96, 96), ("subm0", 128, 96), ("subm0", 96, 96),
         ("spconv1", 32, 32), ("spconv2", 32, 64), ("spconv3", 64, 128), ("spconv4", 128, 256),
         ("spconv4^T", 256, 256), ("spconv3^T", 256, 128), ("spconv2^T", 128, 96), ("spconv1^T", 96, 96)]
configs = [(0, 0), (4, 4), (2, 4), (1, 4), (4, 2), (2, 2), (1, 2), (3, 4), (3, 2)]
only = os.environ.get("PV2_AB_ONLY")
print("%-10s %4s %4s %7s %8s | %-17s | %s" % ("layer", "cin", "cout", "rows", "pairs", "pr fwd  dgrad (us)",
      "osm fwd/dgrad us per (NB,WR): " + " ".join("(%d,%d)" % c for c in configs)))
for key, cin, cout in cases:
    if only and only not in key:
        continue
    t = key.endswith("^T")
    rb = geo[key[:-2] if t else key]["rulebook"]
    if t:
        rb = rb.transposed()
    x = torch.randn(rb.n_in, cin, device=dev)
    g = torch.randn(rb.n_out, cout, device=dev)
    w = torch.randn(cout, rb.K, cin, device=dev) * 0.05
    flops = 2.0 * rb.n_pairs * cin * cout
    t_pr_f = timeit(lambda: K.spconv_forward(x, w, rb))
    t_pr_b = timeit(lambda: K.spconv_grad_input(g, w, rb))
    ref_f, ref_b = K.spconv_forward(x, w, rb), K.spconv_grad_input(g, w, rb)
    cols = []
    for nb, wr in configs:
        if nb and ((cout // 32) % nb or (cin // 32) % nb):
            cols.append("    -/-    ")
            continue
        L.pv2_debug_set_osm(-1, nb, wr, 0)
        of, ob = K.spconv_osm(x, w, rb), K.spconv_osm(g, w, rb, transposed=True)
        ef = (of - ref_f).abs().max().item() / ref_f.abs().max().item()
        eb = (ob - ref_b).abs().max().item() / ref_b.abs().max().item()
        tf = timeit(lambda: K.spconv_osm(x, w, rb))
        tb = timeit(lambda: K.spconv_osm(g, w, rb, transposed=True))
        cols.append("%5.1f/%5.1f%s" % (tf, tb, "" if max(ef, eb) < 3e-5 else "!ERR %.1e" % max(ef, eb)))
    L.pv2_debug_set_osm(-1, 0, 0, 0)
    print("%-10s %4d %4d %7d %8d | %6.1f %6.1f (%5.1f TF) | %s" % (
        key, cin, cout, rb.n_out, rb.n_pairs, t_pr_f, t_pr_b, flops / t_pr_f / 1e6, " ".join(cols)), flush=True)
