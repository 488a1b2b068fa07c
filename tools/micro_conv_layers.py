#!/usr/bin/env python
"""A few launches of every conv kernel family on the bench geometry (for PMC passes): the 16-bit
output-stationary forward and weight gradient, and the fp32 scatter forward / weight gradient, on
the C=128 layer of level 2 (25 684 voxels, 195 k pairs) and the C=256 layer of level 3."""
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
for key, c in (("subm2", 128), ("subm3", 256)):
    rb = geo[key]["rulebook"]
    x32 = torch.randn(rb.n_in, c, device=dev); g32 = torch.randn(rb.n_out, c, device=dev)
    w = torch.randn(c, rb.K, c, device=dev) * 0.05
    x16, g16 = x32.bfloat16(), g32.bfloat16()
    fwd, bwd = K.packed_weights(w, torch.bfloat16)
    for _ in range(6):
        K.spconv16_forward(x16, fwd, rb.K, c, rb.nbr, rb.nbr_stride, rb.perm, rb.kflip, rb.n_out)
        K.spconv16_backward_weight(x16, g16, rb, c)
        K.spconv_forward(x32, w, rb)
        K.spconv_backward_weight(x32, g32, rb, c)
torch.cuda.synchronize()
