#!/usr/bin/env python
"""A few launches of the product-row route and of the mask-grouped output-stationary route on the bench
geometry, for PMC passes (tools/gpu_pmc_micro.sh): where do the waves of the two kernels wait?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["PV2_CONV_OSM"] = "1"
import bench
from ponderv2_amd import kernels as K
from ponderv2_amd.ponder.models.utils import offset2batch

dev = torch.device("cuda:0")
batch = bench.make_batch(0, 2, 2, dev)
idx = torch.cat([offset2batch(batch["offset"]).unsqueeze(-1).int(), batch["grid_coord"].int()], 1).contiguous()
geo = K.prepare_unet_geometry(idx, batch["sparse_shape"])
for key, c in (("subm3", 128), ("subm2", 64), ("subm4", 256)):
    rb = geo[key]["rulebook"]
    x = torch.randn(rb.n_in, c, device=dev)
    w = torch.randn(c, rb.K, c, device=dev) * 0.05
    for _ in range(6):
        K.spconv_forward(x, w, rb)
        K.spconv_osm(x, w, rb)
    torch.cuda.synchronize()
