"""Per-layer timing of the dense 3x3x3 convolutions of UNet3D-v1m2 at the bench size (2 scenes,
128 x 128 x 32 grid): csrc/dense_conv.hip against the library convolution (MIOpen / CK through ATen),
forward / grad-input / grad-weight.  HIP events, median of `reps` launches after warm-up."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from ponderv2_amd import dense_conv as dc  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def cl(t):
    return t.contiguous(memory_format=torch.channels_last_3d)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


B = 2
LAYERS = [  # (name, kind, c_in, c_out, (z, y, x) of the conv input)
    ("enc1 32->64", "conv", 32, 64, (16, 64, 64)),
    ("enc2 64->128", "conv", 64, 128, (8, 32, 32)),
    ("enc3 128->256", "conv", 128, 256, (4, 16, 16)),
    ("dec0 up 256->128", "convT", 256, 128, (4, 16, 16)),
    ("dec0 128->128", "conv", 128, 128, (8, 32, 32)),
    ("dec1 up 128->64", "convT", 128, 64, (8, 32, 32)),
    ("dec1 64->64", "conv", 64, 64, (16, 64, 64)),
    ("dec2 up 64->32", "convT", 64, 32, (16, 64, 64)),
    ("dec2 32->32", "conv", 32, 32, (32, 128, 128)),
]
tot = {"ours": 0.0, "lib": 0.0}
print("%-18s %10s | %8s %8s %7s | %8s %8s %7s | %8s %8s %7s" % (
    "layer", "GFLOP", "fwd us", "lib us", "TF/s", "dgrad us", "lib us", "TF/s", "wgrad us", "lib us", "TF/s"))
for name, kind, ci, co, (z, y, x) in LAYERS:
    torch.manual_seed(0)
    xin = cl(torch.randn(B, ci, z, y, x, device=dev))
    if kind == "conv":
        w = cl(torch.randn(co, ci, 3, 3, 3, device=dev) * 0.05)
        geom = ((1, 1, 1), (1, 1, 1), (1, 1, 1), False, (0, 0, 0), 1)
        gy = cl(torch.randn(B, co, z, y, x, device=dev))
        flops = 2.0 * B * z * y * x * 27 * ci * co
        pf, pb = dc.pack_weights(w, 0, False), dc.pack_weights(w, 1, True)
        ours = [lambda: dc.conv3_forward(xin, pf, co, 0, relu=True),
                lambda: dc.conv3_forward(gy, pb, ci, 0, mask_src=gy),
                lambda: dc.conv3_backward_weight(xin, gy, w, 0, mask_src=gy)]
    else:
        w = cl(torch.randn(ci, co, 3, 3, 3, device=dev) * 0.05)
        geom = ((2, 2, 2), (1, 1, 1), (1, 1, 1), True, (1, 1, 1), 1)
        gy = cl(torch.randn(B, co, 2 * z, 2 * y, 2 * x, device=dev))
        flops = 2.0 * B * z * y * x * 27 * ci * co
        pf, pb = dc.pack_weights(w, 1, False, mode=1), dc.pack_weights(w, 0, False, mode=2)
        ours = [lambda: dc.conv3_forward(xin, pf, co, 1, addend=gy),
                lambda: dc.conv3_forward(gy, pb, ci, 2),
                lambda: dc.conv3_backward_weight(xin, gy, w, 1, n_dim=1)]
    lib = [lambda: torch.ops.aten.convolution(xin, w, None, *geom),
           lambda: torch.ops.aten.convolution_backward(gy, xin, w, None, *geom, [True, False, False]),
           lambda: torch.ops.aten.convolution_backward(gy, xin, w, None, *geom, [False, True, False])]
    row = []
    for a, b in zip(ours, lib):
        ta, tb = timeit(a), timeit(b)
        tot["ours"] += ta
        tot["lib"] += tb
        row += [ta, tb, flops / ta / 1e6]
    print("%-18s %10.2f | %8.1f %8.1f %7.1f | %8.1f %8.1f %7.1f | %8.1f %8.1f %7.1f" % (
        (name, flops / 1e9) + tuple(row)))
print("total us per step: ours %.0f, library %.0f" % (tot["ours"], tot["lib"]))
