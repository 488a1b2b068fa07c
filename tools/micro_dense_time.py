"""Time one dense layer (fwd / dgrad / wgrad) - HIP events, median of 10."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ponderv2_amd import dense_conv as dc
ci, co = int(sys.argv[1]), int(sys.argv[2]); z, y, x = (int(v) for v in sys.argv[3:6])
dev = torch.device("cuda:0")
cl = lambda t: t.contiguous(memory_format=torch.channels_last_3d)
torch.manual_seed(0)
xin = cl(torch.randn(2, ci, z, y, x, device=dev)); gy = cl(torch.randn(2, co, z, y, x, device=dev))
w = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.05
pf, pb = dc.pack_weights(w, 0, False), dc.pack_weights(w, 1, True)
def t(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[5]
print("dbg=%s fwd %.1f  dgrad %.1f  wgrad %.1f us" % (os.environ.get("PV2_DCONV_DEBUG", "0"),
      t(lambda: dc.conv3_forward(xin, pf, co, 0, relu=True)), t(lambda: dc.conv3_forward(gy, pb, ci, 0, mask_src=gy)),
      t(lambda: dc.conv3_backward_weight(xin, gy, w, 0, mask_src=gy))))
