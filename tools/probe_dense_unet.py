#!/usr/bin/env python
"""Memory formats of every intermediate of the dense UNet3D-v1m2 (is it channels-last all the way?)
and a torch-profiler table of its slowest device ops, forward + backward, at the ScanNet grid."""
import os
import sys

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ponderv2_amd.ponder.models.ponder.unet3d import UNet3Dv1m2  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = UNet3Dv1m2(96, 128).to(dev).train().to(memory_format=torch.channels_last_3d)
first = torch.randn(2, 32, 32, 128, 128, device=dev).contiguous(memory_format=torch.channels_last_3d)
first.requires_grad_(True)


def fmt(t):
    if not torch.is_tensor(t) or t.dim() != 5:
        return str(type(t).__name__)
    cl, co = t.is_contiguous(memory_format=torch.channels_last_3d), t.is_contiguous()
    return f"{tuple(t.shape)} {'CL' if cl else ''}{'/contig' if co else ''}{'' if cl or co else 'STRIDED ' + str(t.stride())}"


hooks = []
for name, m in net.named_modules():
    if isinstance(m, (nn.Conv3d, nn.ConvTranspose3d, nn.BatchNorm3d, nn.MaxPool3d, nn.ReLU)):
        hooks.append(m.register_forward_hook(
            lambda mod, inp, out, name=name: print(f"{name:44s} in {fmt(inp[0]):46s} -> out {fmt(out)}")))


def step():
    y = net(None, first=torch.relu(first))
    y.float().square().mean().backward()


step()
for h in hooks:
    h.remove()
torch.cuda.synchronize()
for _ in range(2):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=28,
                                                          max_name_column_width=48, max_shapes_column_width=70))
