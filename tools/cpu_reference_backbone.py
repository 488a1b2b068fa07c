#!/usr/bin/env python
"""BASELINE.md section 3, item 1 - the REFERENCE'S OWN ``spconv_unet_v1m1_base.py`` (unmodified, imported from the
reference checkout through oracle/ref_shims.py: spconv.pytorch -> the oracle's CPU runtime) timed on the host cores:
SparseUNet forward on configs[0] (1 scene, 20 000 voxels) and on the BENCHED batch of configs[1] (2 scenes, 46 842
voxels), thread sweep, 2 warm-up + 5 timed, median.  Runs only where a reference checkout exists (the build
container: 8 cores); the GPU box has none, which is why bench.py's cpu_baseline is kind "port" (the product's model
code on the same oracle kernels).  Usage: python tools/cpu_reference_backbone.py > profiles/r05_cpu_reference_backbone.txt"""
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ref_shims  # noqa: E402

if not ref_shims.reference_available():
    raise SystemExit("no reference checkout (PONDERV2_REFERENCE / /root/reference)")
ref_shims.install()
from ponder.models.builder import MODELS  # noqa: E402  (the reference's registry)
import golden_cases as gc  # noqa: E402
from ponderv2_amd.ponder.datasets import collate_fn, make_scene  # noqa: E402
from ponderv2_amd.ponder.utils.config import ConfigDict  # noqa: E402


def timed(fn, warm=2, n=5):
    for _ in range(warm):
        fn()
    out = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        out.append(time.perf_counter() - t0)
    return statistics.median(out)


torch.manual_seed(0)
model = MODELS.build(ConfigDict(dict(gc.FULL_BACKBONE))).train()
print("reference class:", type(model).__module__, type(model).__name__,
      "parameters", sum(p.numel() for p in model.parameters()))
batches = {"configs[0] 1 scene 20000 voxels": collate_fn([make_scene(0, num_views=2, image_hw=(480, 640), n_voxels=20000)]),
           "configs[1] 2 scenes (benched batch)": collate_fn([make_scene(i, num_views=2, image_hw=(480, 640)) for i in range(2)])}
ncpu = os.cpu_count() or 1
for name, b in batches.items():
    data = {k: b[k] for k in ("grid_coord", "feat", "offset")}

    def fwd():
        with torch.no_grad():
            model({k: v.clone() for k, v in data.items()})

    sweep = {}
    for nt in sorted({t for t in (2, 4, 8, 16, 32, ncpu) if t <= ncpu}):
        torch.set_num_threads(nt)
        sweep[nt] = timed(fwd, 1, 2)
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    t = timed(fwd)
    print(f"{name}: {int(b['offset'][-1])} voxels, SparseUNet forward {t:.2f} s at {best} of {ncpu} threads "
          f"(sweep {({k: round(v, 2) for k, v in sweep.items()})}; 2 warm-up + median of 5; rulebooks rebuilt per "
          "forward, as spconv does per indice_key)")
