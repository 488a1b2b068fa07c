#!/usr/bin/env python
"""fp32 sparse conv per layer shape on the bench geometry: forward, grad-input and weight gradient,
A/B between the kernel selections (product-row path + deterministic weight gradient, the default;
the scatter-add kernels with atomics; output-stationary), and the fused conv + BatchNorm unit
against conv followed by the fused BatchNorm.  Times are HIP-event averages over 20 calls."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from ponderv2_amd import convbn, kernels as K, rownorm
from ponderv2_amd.ponder.models.utils import offset2batch
from ponderv2_amd.spconv import pytorch as spconv

dev = torch.device("cuda:0")
batch = bench.make_batch(0, 2, 2, dev)
idx = torch.cat([offset2batch(batch["offset"]).unsqueeze(-1).int(), batch["grid_coord"].int()], 1).contiguous()
geo = K.prepare_unet_geometry(idx, batch["sparse_shape"])
LAYERS = [("subm1", 32, 32), ("subm2", 64, 64), ("subm3", 128, 128), ("subm4", 256, 256), ("subm3", 384, 256),
          ("subm3", 256, 256), ("subm2", 192, 128), ("subm2", 128, 128), ("subm1", 128, 96), ("subm1", 96, 96),
          ("subm0", 128, 96), ("subm0", 96, 96),
          ("spconv1", 32, 32), ("spconv2", 32, 64), ("spconv3", 64, 128), ("spconv4", 128, 256),
          ("spconv4^T", 256, 256), ("spconv3^T", 256, 128), ("spconv2^T", 128, 96), ("spconv1^T", 96, 96)]
MODES = {"pr": dict(USE_PR="all", USE_OS="auto", USE_WGRAD_DET=True),
         "atomics": dict(USE_PR=False, USE_OS=False, USE_WGRAD_DET=False),
         "os": dict(USE_PR=False, USE_OS=True, USE_WGRAD_DET=False)}


def set_mode(name):
    for k, v in MODES[name].items():
        setattr(K, k, v)


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def rulebook(key):
    if key.endswith("^T"):
        return geo[key[:-2]]["rulebook"].transposed()
    return geo[key]["rulebook"]


print("%-10s %4s %4s %7s %8s | " % ("layer", "cin", "cout", "rows", "pairs")
      + " | ".join("%-7s fwd dgrad wgrad (us)" % m for m in MODES) + " | unit fwd / bwd: fused, modular (us)")
tot = {m: [0.0, 0.0, 0.0] for m in MODES}
unit_tot = [0.0, 0.0, 0.0, 0.0]
for key, c_in, c_out in LAYERS:
    rb = rulebook(key)
    x = torch.randn(rb.n_in, c_in, device=dev)
    g = torch.randn(rb.n_out, c_out, device=dev)
    w = torch.randn(c_out, rb.K, c_in, device=dev) * 0.05
    fl = 2.0 * rb.n_pairs * c_in * c_out
    cols = []
    for m in MODES:
        set_mode(m)
        t = (timeit(lambda: K.spconv_forward(x, w, rb)), timeit(lambda: K.spconv_grad_input(g, w, rb)),
             timeit(lambda: K.spconv_backward_weight(x, g, rb, c_out)))
        tot[m] = [a + b for a, b in zip(tot[m], t)]
        cols.append("%6.1f %6.1f %6.1f (%4.1f TF)" % (t[0], t[1], t[2], fl / t[0] / 1e6))
    # the conv + BatchNorm + ReLU unit, one direction at a time
    set_mode("pr")
    bn = torch.nn.BatchNorm1d(c_out, eps=1e-3, momentum=0.01).to(dev).train()
    wp = torch.nn.Parameter(w.clone())
    xr = x.clone().requires_grad_(True)

    class _Conv:   # what convbn.conv_bn reads from a conv module
        weight, out_channels, in_channels = wp, c_out, c_in

    unit = []
    for fused in (True, False):
        def fwd():
            if fused:
                return convbn.conv_bn(_Conv, bn, xr, rb, relu=True)
            return rownorm.fused_bn(bn, K.SparseConvFunction.apply(xr, wp, rb), relu=True)
        t_f = timeit(fwd)
        y = fwd()

        def bwd():
            torch.autograd.grad(y, (xr, wp, bn.weight, bn.bias), g, retain_graph=True)
        unit += [t_f, timeit(bwd)]
    unit_tot = [a + b for a, b in zip(unit_tot, (unit[0], unit[1], unit[2], unit[3]))]
    print("%-10s %4d %4d %7d %8d | " % (key, c_in, c_out, rb.n_out, rb.n_pairs) + " | ".join(cols)
          + " | %6.1f %6.1f , %6.1f %6.1f" % tuple(unit))
for m in MODES:
    print("sum %-8s: fwd %.0f us, dgrad %.0f us, wgrad %.0f us" % ((m,) + tuple(tot[m])))
print("sum units: fused fwd %.0f bwd %.0f us, modular fwd %.0f bwd %.0f us" % tuple(unit_tot))
