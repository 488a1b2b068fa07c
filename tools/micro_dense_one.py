"""One layer of the dense U-Net (default: dec2 32 -> 32 at 2 x 32 x 128 x 128) through the three
dense_conv.hip kernels, a few launches each - the target of the PMC passes (tools/gpu_pmc_micro.sh)."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from ponderv2_amd import dense_conv as dc  # noqa: E402

ci, co = int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 32
z, y, x = (int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (32, 128, 128)
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device("cuda:0")
cl = lambda t: t.contiguous(memory_format=torch.channels_last_3d)  # noqa: E731
torch.manual_seed(0)
xin = cl(torch.randn(2, ci, z, y, x, device=dev))
gy = cl(torch.randn(2, co, z, y, x, device=dev))
w = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.05
pf, pb = dc.pack_weights(w, 0, False), dc.pack_weights(w, 1, True)
for _ in range(reps):
    dc.conv3_forward(xin, pf, co, 0, relu=True)
    dc.conv3_forward(gy, pb, ci, 0, mask_src=gy)
    dc.conv3_backward_weight(xin, gy, w, 0, mask_src=gy)
torch.cuda.synchronize()
print("done")
