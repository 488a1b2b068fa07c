#!/usr/bin/env python
"""Microbenchmark of the fused BatchNorm kernels per PV2_BN_MAXBLOCKS (read once per process)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(93000, 32), (93000, 96), (40000, 64), (14000, 128), (4000, 256), (1000, 256), (93000, 128)]


def child():
    import torch
    import torch.nn as nn

    from ponderv2_amd.rownorm import fused_bn

    dev = torch.device("cuda:0")
    row = []
    for n, c in SHAPES:
        bn = nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev).train()
        x = torch.randn(n, c, device=dev, requires_grad=True)
        g = torch.randn(n, c, device=dev)

        def fwd():
            return fused_bn(bn, x, relu=True)

        def timeit(fn, reps=30):
            for _ in range(5):
                fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                fn()
            e.record()
            torch.cuda.synchronize()
            return s.elapsed_time(e) / reps * 1e3

        t_f = timeit(fwd)
        y = fwd()
        t_b = timeit(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
        row.append("%dx%d f%5.1f b%5.1f" % (n, c, t_f, t_b))
    print("maxblocks %5s | " % os.environ.get("PV2_BN_MAXBLOCKS", "-") + " | ".join(row), flush=True)


if __name__ == "__main__":
    if os.environ.get("PV2_ROWNORM_CHILD"):
        child()
    else:
        for m in (sys.argv[1:] or ["128", "256", "512", "1024", "2048"]):
            env = dict(os.environ, PV2_BN_MAXBLOCKS=m, PV2_ROWNORM_CHILD="1")
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)
