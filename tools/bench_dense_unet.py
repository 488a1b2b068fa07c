#!/usr/bin/env python
"""Micro-benchmark of the dense UNet3D-v1m2 projection (MIOpen) under a few settings."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ponderv2_amd.ponder.models.ponder.unet3d import UNet3Dv1m2  # noqa: E402


def run(tag, bench_flag, channels_last, autocast_dtype, B=2):
    torch.backends.cudnn.benchmark = bench_flag
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = UNet3Dv1m2(96, 128).to(dev).train()
    x = torch.randn(B, 96, 32, 128, 128, device=dev)
    x = x * (torch.rand(B, 1, 32, 128, 128, device=dev) < 0.1)
    if channels_last:
        net = net.to(memory_format=torch.channels_last_3d)
        x = x.contiguous(memory_format=torch.channels_last_3d)
    x.requires_grad_(True)

    def step():
        with torch.autocast("cuda", dtype=autocast_dtype, enabled=autocast_dtype is not None):
            y = net(x)
        s = torch.cuda.Event(enable_timing=True)
        s.record()
        y.float().square().mean().backward()
        return s

    t0 = time.time()
    step()
    torch.cuda.synchronize()
    first = time.time() - t0
    fw, bw = [], []
    for _ in range(3):
        a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        b = step()
        c.record()
        torch.cuda.synchronize()
        fw.append(a.elapsed_time(b))
        bw.append(b.elapsed_time(c))
    print(f"{tag}: first {first:.1f}s  fwd {min(fw):.1f} ms  bwd {min(bw):.1f} ms", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["a", "b", "c", "d"]
    if "a" in which:
        run("fp32 NCDHW benchmark=False", False, False, None)
    if "b" in which:
        run("fp32 NCDHW benchmark=True", True, False, None)
    if "c" in which:
        run("fp32 NDHWC benchmark=True", True, True, None)
    if "d" in which:
        run("bf16-autocast NCDHW benchmark=True", True, False, torch.bfloat16)
    if "e" in which:
        run("bf16-autocast NDHWC benchmark=True", True, True, torch.bfloat16)
