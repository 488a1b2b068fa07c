#!/usr/bin/env python
"""Microbenchmark of the trilinear sampler kernels (forward / backward / backward-of-backward) on
the two render-head shapes, per PV2_TRI_MODE (the mode is read once per process, so each mode runs
in a child process).  usage: python tools/bench_sampler.py [modes, default 0123]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = {  # name: (B, C, D, H, W, points per volume)
    "outdoor C=32": (4, 32, 5, 180, 180, 3072 * 96),
    "indoor C=128": (2, 128, 32, 128, 128, 512 * 132),
}


def child():
    import torch

    from ponderv2_amd import kernels as K

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for name, (B, C, D, H, W, P) in SHAPES.items():
        vol = torch.randn(B, C, D, H, W, device=dev).contiguous(memory_format=torch.channels_last_3d)
        grid = torch.rand(B, 1, 1, P, 3, device=dev) * 2.1 - 1.05
        out = K.trilinear_forward(vol, grid)
        gout = torch.randn_like(out)
        hV = torch.randn_like(vol)
        hG = torch.randn_like(grid)

        def timeit(fn, n=10):
            for _ in range(3):
                fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(n):
                fn()
            e.record()
            torch.cuda.synchronize()
            return s.elapsed_time(e) / n * 1e3

        t_f = timeit(lambda: K.trilinear_forward(vol, grid))
        t_b = timeit(lambda: K.trilinear_backward(gout, vol, grid, "zeros", True, False, True))
        t_bg = timeit(lambda: K.trilinear_backward(gout, vol, grid, "zeros", True, False, False))
        t_bb = timeit(lambda: K.trilinear_backward_backward(hV, hG, vol, grid, gout, "zeros", True,
                                                             False, True))
        t_z = timeit(lambda: torch.zeros_like(vol))
        pts = B * P
        print("mode %s  %-13s pts %8d  fwd %7.1f us (%5.0f GB/s corner reads)  bwd %7.1f  "
              "bwd(no gV) %7.1f  bwdbwd %7.1f  [zeros_like(vol) %6.1f]" % (
                  os.environ.get("PV2_TRI_MODE", "-"), name, pts, t_f,
                  pts * 8 * C * 4 / t_f / 1e3, t_b, t_bg, t_bb, t_z), flush=True)


if __name__ == "__main__":
    if os.environ.get("PV2_SAMPLER_CHILD"):
        child()
    else:
        for m in (sys.argv[1] if len(sys.argv) > 1 else "0123"):
            env = dict(os.environ, PV2_TRI_MODE=m, PV2_SAMPLER_CHILD="1")
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)
