import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import golden_cases as gc
from oracle import rulebook as orb
from oracle.sparse_ops import sparse_conv
from ponderv2_amd import kernels as K
dev = torch.device("cuda:0")
g = np.load(os.path.join(gc.GOLDEN, "spunet_small.npz"))
coords = g["coords"]
shape = [int(v) + 96 for v in coords[:, 1:].max(0)]
oshape = [(s - 2) // 2 + 1 for s in shape]
rb, oc = K.build_downsample_rulebook(torch.from_numpy(coords).to(dev), 2, oshape)
ooc, pin, pout, ks = orb.downsample_rulebook(coords, 2, oshape)
ok_rb = np.array_equal(rb.pair_in.cpu().numpy(), pin) and np.array_equal(rb.pair_out.cpu().numpy(), pout) and np.array_equal(oc.cpu().numpy(), ooc)
torch.manual_seed(0)
n0, n1 = len(coords), len(ooc)
res = []
for cin, cout in ((32, 32), (16, 32), (32, 16)):
    x = torch.randn(n0, cin); w = torch.randn(cout, 8, cin) * 0.1
    ref = sparse_conv(x.double(), w.double(), torch.from_numpy(pin.astype(np.int64)), torch.from_numpy(pout.astype(np.int64)), ks, n1)
    got = K.spconv_forward(x.to(dev), w.to(dev), rb).double().cpu()
    e1 = ((got - ref).abs().max() / ref.abs().max()).item()
    # inverse direction
    y = torch.randn(n1, cout); wi = torch.randn(cin, 8, cout) * 0.1
    refi = sparse_conv(y.double(), wi.double(), torch.from_numpy(pout.astype(np.int64)), torch.from_numpy(pin.astype(np.int64)), ks, n0)
    goti = K.spconv_forward(y.to(dev), wi.to(dev), rb.transposed()).double().cpu()
    e2 = ((goti - refi).abs().max() / refi.abs().max()).item()
    gw = K.spconv_backward_weight(x.to(dev), torch.randn(n1, cout).to(dev), rb, cout)
    res += [e1, e2]
print("rulebook_ok", ok_rb, "errs", ["%.1e" % e for e in res])
